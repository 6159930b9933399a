"""``Tester``: run a trained model over the test sequences and integrate the predicted transforms into trajectories.
Mirror of src/deploy/tester.py:16-162 without MLflow/plot artefacts: per sequence it writes the KITTI-format pose file,
the raw transformations and the integrated poses (the same three files the reference logs, deployer.py:96-117), into
``config["output_dir"]`` (default /tmp, as the reference)."""
import os

import numpy as np
import torch

from ..data import feed
from ..utility import poses as poses_module
from . import deployer


class Tester(deployer.Deployer):

    def __init__(self, config, dataset=None, geometry_backend=None):
        super().__init__(config=config, dataset=dataset, geometry_backend=geometry_backend)
        self.training_bool = False
        if not config["checkpoint"]:
            raise Exception("No checkpoint specified.")
        checkpoint = torch.load(config["checkpoint"], map_location=self.device, weights_only=False)
        self.model.load_state_dict(checkpoint["model_state_dict"])
        print("Model weights loaded from " + config["checkpoint"])
        print("Batch size set to 1 for the testing.")
        self.batch_size = 1
        self.model.eval()
        self.computed_transformations_datasets = [[[] for _ in config[ds]["data_identifiers"]] for ds in config["datasets"]]
        self.written = []

    def log_map(self, index_of_dataset, index_of_sequence, dataset, data_identifier):
        """Integrate and store one finished sequence (reference deployer.py:88-117, files only)."""
        T = self.computed_transformations_datasets[index_of_dataset][index_of_sequence]
        if not T:
            return
        out_dir = self.config.get("output_dir", "/tmp")
        base = os.path.join(out_dir, self.config["run_name"])
        tag = dataset + "_" + format(data_identifier, "02d")
        computed_poses = poses_module.compute_poses(computed_transformations=T)
        files = {"poses_text": base + "_poses_text_file_" + tag + ".txt", "transformations": base + "_transformations_" + tag + ".npy",
                 "poses": base + "_poses_" + tag + ".npy"}
        poses_module.write_poses_to_text_file(file_name=files["poses_text"], poses=computed_poses)
        np.save(files["transformations"], np.asarray(T))
        np.save(files["poses"], computed_poses)
        self.written.append(files)

    def test_dataset(self, dataloader):
        epoch_losses = {k: 0.0 for k in ("loss_epoch", "loss_point_cloud_epoch", "loss_field_of_view_epoch", "loss_po2po_epoch",
                                         "loss_po2pl_epoch", "loss_pl2pl_epoch", "visible_pixels_epoch")}
        cur_ds, cur_seq, dataset = 0, 0, self.config["datasets"][0]
        for index, preprocessed_dicts in enumerate(feed.DevicePrefetcher(dataloader, self.device)):
            with torch.set_grad_enabled(False):
                if not self.config["inference_only"]:
                    epoch_losses, T = self.step(preprocessed_dicts=preprocessed_dicts, epoch_losses=epoch_losses)
                else:
                    T = self.step(preprocessed_dicts=preprocessed_dicts, epoch_losses=epoch_losses)
            for d in preprocessed_dicts:
                if d["index_sequence"] != cur_seq or d["index_dataset"] != cur_ds:      # a sequence is complete
                    self.log_map(cur_ds, cur_seq, dataset, self.config[dataset]["data_identifiers"][cur_seq])
                    cur_seq, cur_ds, dataset = d["index_sequence"], d["index_dataset"], d["dataset"]
                self.computed_transformations_datasets[d["index_dataset"]][d["index_sequence"]].append(T.detach().cpu().numpy())
            if not index % 10:
                print("Index: " + str(index) + " / " + str(len(dataloader)) + ", dataset: " + dataset + ", sequence: " + str(cur_seq))
        self.log_map(cur_ds, cur_seq, dataset, self.config[dataset]["data_identifiers"][cur_seq])
        return epoch_losses

    def test(self):
        dataloader = torch.utils.data.DataLoader(dataset=self.dataset, batch_size=self.batch_size, shuffle=False,
                                                 collate_fn=Tester.list_collate, num_workers=self.config["num_dataloader_workers"])
        epoch_losses = self.test_dataset(dataloader=dataloader)
        if not self.config["inference_only"]:
            n = max(len(dataloader), 1)
            summary = {k: float(v) / n for k, v in epoch_losses.items()}
            print("loss: " + str(summary["loss_epoch"]) + ", loss_point_cloud: " + str(summary["loss_point_cloud_epoch"]))
            return summary
        return None
