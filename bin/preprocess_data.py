#!/usr/bin/env python3
"""Offline preprocessing entry point (reference bin/preprocess_data.py): raw KITTI scans -> scans/normals npy lists."""
import os
import sys

import click

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import delora_amd.compat  # noqa: E402,F401
import delora_amd.config  # noqa: E402
import preprocessing.preprocesser  # noqa: E402


@click.command()
@click.option("--yes", is_flag=True, help="do not ask before writing into the preprocessed_path directories")
def main(yes):
    cfg = delora_amd.config.load_yaml_config()
    cfg["device"] = delora_amd.config.resolve_device(cfg["device"])
    delora_amd.config.degrees_to_radians(cfg)
    for ds in cfg["datasets"]:
        cfg[ds]["data_identifiers"] = cfg[ds]["training_identifiers"] + cfg[ds]["testing_identifiers"]
    if not yes and input("Write preprocessed files for " + str(cfg["datasets"]) + "? [y/n] ").strip().lower() != "y":
        return
    preprocessing.preprocesser.Preprocesser(config=cfg).preprocess_data()


if __name__ == "__main__":
    main()
