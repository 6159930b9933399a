"""Flat run configuration: the three YAML files merged into one dict, later files overriding earlier ones,
angles converted to radians in place -- the reference's CLI behaviour (bin/run_training.py:22-29,59-67)."""
import os

import numpy as np
import torch
import yaml

FILES = ("config_datasets.yaml", "deployment_options.yaml", "hyperparameters.yaml")


def load_yaml_config(config_dir="config"):
    cfg = {}
    for name in FILES:
        with open(os.path.join(config_dir, name)) as f:
            cfg.update(yaml.load(f, Loader=yaml.FullLoader))
    return cfg


def degrees_to_radians(cfg, datasets=None):
    for ds in (datasets if datasets is not None else cfg["datasets"]):
        vf = cfg[ds]["vertical_field_of_view"]
        vf[0] *= (np.pi / 180.0)
        vf[1] *= (np.pi / 180.0)
    hf = cfg["horizontal_field_of_view"]
    hf[0] *= (np.pi / 180.0)
    hf[1] *= (np.pi / 180.0)
    return cfg


def resolve_device(name):
    """``"cuda"`` becomes this process' GPU (LOCAL_RANK under torchrun), anything else passes through."""
    if str(name) == "cuda":
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    return torch.device(name)


def training_config(training_run_name, experiment_name="", checkpoint="", config_dir="config"):
    """The dict ``bin/run_training.py`` hands to ``Trainer`` (reference bin/run_training.py:21-88)."""
    cfg = load_yaml_config(config_dir)
    loaded = None
    if checkpoint:
        ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False)
        if "parameters" in ckpt:
            print("Found parameters in checkpoint of previous run! Setting part of parameters to those ones.")
            loaded = ckpt["parameters"]
        else:
            print("Checkpoint does not contain any parameters. Using those ones specified in the YAML files.")
    if loaded is not None:
        # the stored run config wins, except for where/what to run on (run_training.py:47-55); its angles are radians already
        loaded["device"] = resolve_device(cfg["device"])
        loaded["datasets"] = cfg["datasets"]
        for ds in loaded["datasets"]:
            loaded[ds]["training_identifiers"] = cfg[ds]["training_identifiers"]
            loaded[ds]["data_identifiers"] = loaded[ds]["training_identifiers"]
        cfg = loaded
    else:
        cfg["device"] = resolve_device(cfg["device"])
        for ds in cfg["datasets"]:
            cfg[ds]["data_identifiers"] = cfg[ds]["training_identifiers"]
        degrees_to_radians(cfg)
    cfg["checkpoint"] = str(checkpoint) if checkpoint else None
    cfg["training_run_name"] = str(training_run_name)
    cfg["run_name"] = cfg["training_run_name"]
    if experiment_name:
        cfg["experiment"] = experiment_name
    cfg["mode"] = "training"
    return cfg


def testing_config(testing_run_name, experiment_name="testing", checkpoint="", config_dir="config"):
    """The dict ``bin/run_testing.py`` hands to ``Tester`` (reference bin/run_testing.py:20-88)."""
    cfg = load_yaml_config(config_dir)
    ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False)
    if "parameters" in ckpt:
        print("Found parameters in checkpoint! Setting part of parameters to those ones.")
        loaded = ckpt["parameters"]
        loaded["device"] = resolve_device(cfg["device"])
        loaded["datasets"] = cfg["datasets"]
        for ds in loaded["datasets"]:
            loaded[ds]["testing_identifiers"] = cfg[ds]["testing_identifiers"]
            loaded[ds]["data_identifiers"] = loaded[ds]["testing_identifiers"]
        loaded["inference_only"] = cfg["inference_only"]
        loaded["store_dataset_in_RAM"] = cfg["store_dataset_in_RAM"]
        cfg = loaded
    else:
        print("Checkpoint does not contain any parameters. Using those ones specified in the YAML files.")
        cfg["device"] = resolve_device(cfg["device"])
        for ds in cfg["datasets"]:
            cfg[ds]["data_identifiers"] = cfg[ds]["testing_identifiers"]
        degrees_to_radians(cfg)
    if cfg["use_dropout"]:
        cfg["use_dropout"] = False
        print("Deactivating dropout for this mode.")
    cfg["run_name"] = str(testing_run_name)
    cfg["checkpoint"] = str(checkpoint)
    if experiment_name:
        cfg["experiment"] = experiment_name
    cfg["mode"] = "testing"
    cfg["unsupervised_at_start"] = True
    return cfg


def rosnode_config(checkpoint, dataset, lidar_topic, lidar_frame, integrate_odometry=True, config_dir="config"):
    """The dict the reference's ``bin/run_rosnode.py`` (:27-71) builds; here it configures ``ros_utils.odometry.ScanToScanOdometry``, the
    ROS-free core of that node.  The ROS node itself (publishers, TF, message conversion) is out of scope and not built."""
    cfg = load_yaml_config(config_dir)
    cfg["mode"] = "training"
    if cfg["use_dropout"]:
        cfg["use_dropout"] = False
        print("Deactivating dropout for this mode.")
    cfg["checkpoint"] = str(checkpoint)
    cfg["datasets"] = [str(dataset)]
    cfg["lidar_topic"] = str(lidar_topic)
    cfg["lidar_frame"] = str(lidar_frame)
    cfg["integrate_odometry"] = integrate_odometry
    cfg["device"] = resolve_device(cfg["device"])
    degrees_to_radians(cfg)
    return cfg
