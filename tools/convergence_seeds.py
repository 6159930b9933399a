#!/usr/bin/env python3
"""The convergence claim as a statistic (review of round 5: one trajectory per precision is not evidence).

S training seeds x {float32, bfloat16} x {trunk as one autograd Function ("mono"), cut per layer as a DDP rank runs it ("layer")} on ONE
dataset whose scenes make the forward translation observable (cross-walls and pillars, delora_amd/data/synthetic.Scene), through the
pipeline of tools/convergence.py: offline preprocessing -> Trainer.train (identity pre-training, the unsupervised loss at --lr, the last
third at --lr / 10 resumed from the checkpoint) -> Tester.test -> KITTI-style relative errors on a held-out sequence
(reference: src/deploy/trainer.py:93-186, src/deploy/tester.py:38-162, src/utility/poses.py:11-74).  Reports mean and standard deviation
per cell of the final loss and of the held-out translation / rotation errors; eager steps (bit-reproducible).

    python tools/convergence_seeds.py [--seeds 5] [--epochs 180] [--lr 1e-4] [--out gpurun_out/convergence_seeds.json]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import convergence as C          # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--epochs", type=int, default=180)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sequences", type=int, default=4)
    ap.add_argument("--scans", type=int, default=41)
    ap.add_argument("--precisions", default="float32,bfloat16")
    ap.add_argument("--cuts", default="mono,layer")
    ap.add_argument("--cross-walls", type=int, default=10)
    ap.add_argument("--pillars", type=int, default=12)
    ap.add_argument("--budget-s", type=float, default=1500.0, help="stop starting new runs after this many seconds (cells then have fewer seeds)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence_seeds.json"))
    args = ap.parse_args(argv)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from delora_amd.models import ring_conv
    t_start = time.perf_counter()
    scene = {"cross_walls": args.cross_walls, "pillars": args.pillars}
    tree, truth = C.build_dataset(device, args.sequences, args.scans, scene=scene)
    out_dir = tempfile.mkdtemp(prefix="delora_convseeds_")
    report = {"what": "final loss and held-out KITTI-style errors: mean +- sd over training seeds, per precision and trunk cut",
              "image": "64x720", "batch": args.batch, "learning_rate": args.lr, "schedule": f"{args.epochs} epochs: 2/3 at lr, 1/3 at lr/10 resumed from the checkpoint",
              "scene": scene, "dataset": f"{args.sequences} training sequences + 1 held out, {args.scans} scans each", "hip_graph": False, "runs": [], "cells": {}}
    try:
        cells = [(p, c) for c in args.cuts.split(",") for p in args.precisions.split(",")]
        for k in range(args.seeds):                               # seed-major: a budget cut leaves every cell with the same seeds
            for precision, cut in cells:
                if time.perf_counter() - t_start > args.budget_s:
                    break
                ring_conv.TRUNK_SEGMENTS = cut
                t0 = time.perf_counter()
                r = C.train_and_test(device, tree, truth, precision, args.epochs, args.lr, args.batch, 11 + 101 * k, out_dir,
                                     extra={"store_dataset_in_RAM": True, "num_dataloader_workers": 0, "hip_graph": False, "trunk_segments": cut})
                row = {"seed": 11 + 101 * k, "precision": precision, "cut": cut, "steps": r["steps"],
                       "final_loss": r.get("loss_plateau_last_epochs_mean"), "first_loss": r.get("loss_first_epochs_mean"),
                       "t_rel_percent": r["held_out_sequence"]["translation_error_percent"], "r_rel_deg_per_m": r["held_out_sequence"]["rotation_error_deg_per_m"],
                       "per_step_t_m": r["held_out_sequence"]["per_step_translation_error_m_mean"], "per_step_r_deg": r["held_out_sequence"]["per_step_rotation_error_deg_mean"],
                       "yardstick_no_motion_percent": r["held_out_sequence"]["yardstick_no_motion_percent"],
                       "yardstick_mean_motion_percent": r["held_out_sequence"]["yardstick_mean_motion_percent"], "wall_s": round(time.perf_counter() - t0, 1)}
                report["runs"].append(row)
                print(row, flush=True)
        for precision, cut in cells:
            rows = [r for r in report["runs"] if r["precision"] == precision and r["cut"] == cut]
            if not rows:
                continue
            cell = {"n": len(rows)}
            for key in ("final_loss", "t_rel_percent", "r_rel_deg_per_m", "per_step_t_m", "per_step_r_deg"):
                v = np.array([r[key] for r in rows], dtype=np.float64)
                cell[key] = {"mean": round(float(v.mean()), 5), "sd": round(float(v.std(ddof=1)) if len(v) > 1 else 0.0, 5), "values": [round(float(x), 5) for x in v]}
            report["cells"][f"{precision}/{cut}"] = cell
        report["wall_s"] = round(time.perf_counter() - t_start, 1)
    finally:
        ring_conv.TRUNK_SEGMENTS = "mono"
        shutil.rmtree(tree, ignore_errors=True)
        shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    for k, c in report["cells"].items():
        print(k, "n", c["n"], "loss", c["final_loss"]["mean"], "+-", c["final_loss"]["sd"], " t_rel", c["t_rel_percent"]["mean"], "+-", c["t_rel_percent"]["sd"],
              " r_rel", c["r_rel_deg_per_m"]["mean"], "+-", c["r_rel_deg_per_m"]["sd"])
    print("written", args.out)


if __name__ == "__main__":
    main()
