#!/usr/bin/env python3
"""Testing entry point with the reference's command line (bin/run_testing.py:14-20):

    python bin/run_testing.py --testing_run_name NAME --checkpoint FILE [--experiment_name EXP]

runs the model over the ``testing_identifiers`` sequences and writes KITTI-format pose files.
"""
import os
import sys

import click

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import delora_amd.compat  # noqa: E402,F401
import delora_amd.config  # noqa: E402
import deploy.tester  # noqa: E402


@click.command()
@click.option("--testing_run_name", prompt="MLFlow name of the run", help="The name under which the run can be found afterwards.")
@click.option("--experiment_name", help="High-level testing sequence name for clustering in MLFlow.", default="testing")
@click.option("--checkpoint", prompt="Path to the saved checkpoint of the model you want to test")
def config(testing_run_name, experiment_name, checkpoint):
    cfg = delora_amd.config.testing_config(testing_run_name, experiment_name, checkpoint)
    print("----------------------------------")
    print("Configuration for this run: ")
    print(cfg)
    print("----------------------------------")
    return cfg


if __name__ == "__main__":
    cfg = config(standalone_mode=False)
    tester = deploy.tester.Tester(config=cfg)
    tester.test()
    for files in tester.written:
        print(files["poses_text"])
