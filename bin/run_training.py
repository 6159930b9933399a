#!/usr/bin/env python3
"""Training entry point with the reference's command line (bin/run_training.py:14-21):

    python bin/run_training.py --training_run_name NAME [--experiment_name EXP] [--checkpoint FILE] [--max_epochs N]

reads config/*.yaml relative to the working directory.  Multi-GPU: launch the same command through
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bin/run_training.py ...``
(one process per GPU, RCCL); a single process behaves exactly as before.
"""
import os
import sys

import click
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import delora_amd.compat  # noqa: E402,F401  (exposes deploy/, utility/, ... under the reference's import names)
import delora_amd.config  # noqa: E402
import deploy.trainer  # noqa: E402


@click.command()
@click.option("--training_run_name", prompt="MLFlow name of the run",
              help="The name under which the run can be found afterwards.")
@click.option("--experiment_name", help="High-level training sequence name for clustering in MLFlow.", default="")
@click.option("--checkpoint", help="Path to the saved checkpoint. Leave empty if none.", default="")
@click.option("--max_epochs", type=int, default=10000, show_default=True,
              help="Stop after this many epochs (the reference trains its 10000 epochs until interrupted, src/deploy/trainer.py:93).")
def config(training_run_name, experiment_name, checkpoint, max_epochs):
    cfg = delora_amd.config.training_config(training_run_name, experiment_name, checkpoint)
    cfg["max_epochs"] = int(max_epochs)
    print("----------------------------------")
    print("Configuration for this run: ")
    print(cfg)
    print("----------------------------------")
    return cfg


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group(backend="nccl")
    cfg = config(standalone_mode=False)
    trainer = deploy.trainer.Trainer(config=cfg)
    trainer.train(max_epochs=cfg.pop("max_epochs"))
