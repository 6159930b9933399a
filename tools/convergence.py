#!/usr/bin/env python3
"""Convergence evidence for every precision the bench advertises (fp32, bf16 autocast, fp16 autocast + loss scaling).

A synthetic DATASET with known ego-motion -- several sequences of one scene each (delora_amd/data/synthetic.make_sequence: ~0.45 m
forward and up to 1.5 degrees of yaw per scan) -- goes through the reference's whole pipeline on the HIP path:

    offline preprocessing (64x2250 point + normal lists, src/preprocessing/preprocesser.py:52-68)
 -> Trainer.train() at the shipped 64x720: identity pre-training until its loss < 1e-2, then the unsupervised geometric loss
    (src/deploy/trainer.py:93-186; the same random seed, weights and sample order for every precision)
 -> Tester.test() on the training sequences and on a held-out one: transforms integrated by utility/poses.compute_poses
    (src/deploy/tester.py:38-162, src/utility/poses.py:11-74)
 -> KITTI-style relative translation / rotation error of the integrated trajectory against the ground-truth poses
    (utility/poses.relative_pose_errors; segment lengths in metres scaled to the sequences' ~18 m).

    python tools/convergence.py [--epochs 300] [--lr 1e-4] [--batch 8] [--out gpurun_out/convergence.json]

The learning rate is a parameter of the run and is recorded: the reference's 1e-5 (config/hyperparameters.yaml:4) is meant for days
of KITTI; a run of a few thousand steps needs a larger one to get anywhere (Adam moves a weight by ~lr per step)."""
import argparse
import copy
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


def build_dataset(device, train_sequences, scans_per_sequence, seed=7000, rings=64, azimuth_steps=2250, workdir=None, scene=None):
    """Sequences 0..S-1 = training, sequence S = held out.  Returns (tree path, {sequence: ground-truth relative transforms [K,4,4]})."""
    from delora_amd import config as cfgmod
    from delora_amd.data import synthetic
    from delora_amd.preprocessing.preprocesser import Preprocesser
    tmp = workdir or tempfile.mkdtemp(prefix="delora_conv_")
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfgmod.degrees_to_radians(cfg)
    cfg["device"] = device
    cfg["kitti"]["preprocessed_path"] = tmp
    if azimuth_steps != 2250:
        cfg["kitti"]["horizontal_cells_preprocessing"] = azimuth_steps
    cfg["kitti"]["vertical_cells"] = rings
    truth = {}
    pre = Preprocesser(cfg)
    for s in range(train_sequences + 1):
        scans, poses = synthetic.make_sequence(seed + 17 * s, scans_per_sequence, rings=rings, azimuth_steps=azimuth_steps, scene=scene)
        pre.preprocess_scans(scans, "kitti", s)
        truth[s] = np.stack([np.linalg.inv(a) @ b for a, b in zip(poses[:-1], poses[1:])])
    return tmp, truth


def run_config(device, tree, identifiers, precision, batch, lr, run_name, out_dir, H=64, W=720, extra=None):
    from delora_amd import config as cfgmod
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfgmod.degrees_to_radians(cfg)
    cfg["kitti"].update(preprocessed_path=tree, data_identifiers=list(identifiers), training_identifiers=list(identifiers),
                        vertical_cells=H, horizontal_cells=W)
    cfg.update(device=device, batch_size=batch, learning_rate=lr, checkpoint=None, mode="training", training_run_name=run_name,
               run_name=run_name, store_dataset_in_RAM=True, checkpoint_dir=out_dir, output_dir=out_dir, inference_only=False,
               unsupervised_at_start=False, checkpoint_every=0, checkpoint_keep_every=0)          # the last epoch's checkpoint only
    if precision != "float32":
        cfg["amp_dtype"] = precision
    cfg.update(extra or {})
    return cfg


def evaluate(device, tree, truth, sequences, checkpoint, precision, out_dir, run_name, H=64, W=720, lengths=(2.0, 5.0, 10.0, 15.0)):
    """Tester.test() over `sequences` with the trained checkpoint; errors of the integrated trajectories against the ground truth."""
    from delora_amd.deploy.tester import Tester
    from delora_amd.utility import poses as P
    cfg = run_config(device, tree, sequences, precision, 1, 0.0, run_name, out_dir, H, W)
    cfg.update(checkpoint=checkpoint, mode="testing", inference_only=True, unsupervised_at_start=True, store_dataset_in_RAM=False)
    tester = Tester(cfg)
    tester.test()
    out = {}
    for k, s in enumerate(sequences):
        T = np.asarray(tester.computed_transformations_datasets[0][k], dtype=np.float64).reshape(-1, 4, 4)
        est, gt = P.compute_poses(list(T)), P.compute_poses(list(truth[s]))
        err = P.relative_pose_errors(est, gt, lengths_m=lengths, step=1)
        step_t = np.linalg.norm(T[:, :3, 3] - truth[s][:, :3, 3], axis=1)
        dR = np.einsum("kij,kil->kjl", T[:, :3, :3], truth[s][:, :3, :3])
        step_r = np.degrees(np.arccos(np.clip(0.5 * (np.trace(dR, axis1=1, axis2=2) - 1.0), -1.0, 1.0)))
        # yardsticks on the same sequence: a predictor that always says "no motion", and one that always says "the mean motion"
        still = P.relative_pose_errors(P.compute_poses([np.eye(4)] * len(T)), gt, lengths_m=lengths, step=1)
        mean_T = np.eye(4)
        mean_T[:3, 3] = truth[s][:, :3, 3].mean(axis=0)
        mean = P.relative_pose_errors(P.compute_poses([mean_T] * len(T)), gt, lengths_m=lengths, step=1)
        out[int(s)] = {"translation_error_percent": round(100 * err["translation"], 3), "rotation_error_deg_per_m": round(np.degrees(err["rotation_rad_per_m"]), 4),
                       "segments": err["segments"], "per_step_translation_error_m_mean": round(float(step_t.mean()), 4),
                       "per_step_rotation_error_deg_mean": round(float(step_r.mean()), 4),
                       "end_point_error_m": round(float(np.linalg.norm(est[-1, :3, 3] - gt[-1, :3, 3])), 3),
                       "path_length_m": round(float(np.linalg.norm(truth[s][:, :3, 3], axis=1).sum()), 2),
                       "yardstick_no_motion_percent": round(100 * still["translation"], 2),
                       "yardstick_mean_motion_percent": round(100 * mean["translation"], 2),
                       "yardstick_mean_motion_rotation_deg_per_m": round(np.degrees(mean["rotation_rad_per_m"]), 4)}
    return out


def train_and_test(device, tree, truth, precision, epochs, lr, batch, seed, out_dir, H=64, W=720, extra=None, anneal_fraction=1.0 / 3.0):
    """Two stages through the product's own entry points: ``epochs * (1 - anneal_fraction)`` epochs at ``lr`` (identity pre-training,
    then the unsupervised loss), then -- resumed from the checkpoint the first stage wrote, the way a user resumes a run
    (src/deploy/trainer.py:27-36: weights + optimiser state, unsupervised from the start) -- the remaining epochs at ``lr / 10``: the
    reference has no schedule, a constant 1e-4 keeps spiking (three precisions then end wherever their last spike left them), and
    1e-5 is the reference's own rate."""
    from delora_amd.deploy.trainer import Trainer
    train_ids = sorted(truth)[:-1]
    held_out = sorted(truth)[-1]
    name = "conv_" + precision
    ckpt = os.path.join(out_dir, name + "_latest_checkpoint.pth")
    epochs_lo = int(round(epochs * anneal_fraction))
    stages = [(epochs - epochs_lo, lr, None)] + ([(epochs_lo, 0.1 * lr, ckpt)] if epochs_lo > 0 else [])
    history, wall, graph_steps, probes, feed_report = [], 0.0, 0, {}, None
    for stage, (n_epochs, rate, resume) in enumerate(stages):
        cfg = run_config(device, tree, train_ids, precision, batch, rate, name, out_dir, H, W, extra)
        cfg["checkpoint"] = resume
        torch.manual_seed(seed + stage)
        np.random.seed(seed + stage)
        trainer = Trainer(cfg)
        for group in trainer.optimizer.param_groups:          # (a resumed optimiser comes back with the rate it was saved with)
            group["lr"] = rate
        t0 = time.perf_counter()
        h = trainer.train(max_epochs=n_epochs)
        torch.cuda.synchronize()
        wall += time.perf_counter() - t0
        history += [dict(e, stage=stage, learning_rate=rate) for e in h]
        graph_steps += getattr(trainer, "graph_steps", 0)
        probes[f"stage{stage}"] = {str(k): v for k, v in getattr(trainer, "graph_probe_result", {}).items()}
        feed_report = getattr(trainer, "feed_report", None)
        steps_per_epoch = len(trainer.dataset) // batch
        del trainer
    identity_epochs = sum(1 for h in history if not h["unsupervised"])
    unsup = [h for h in history if h["unsupervised"]]
    curve = [round(h["loss_epoch"], 6) for h in unsup]
    result = {"precision": precision, "epochs": len(history), "steps": len(history) * steps_per_epoch, "steps_per_epoch": steps_per_epoch,
              "stages": [{"epochs": n, "learning_rate": r, "resumed_from_checkpoint": c is not None} for n, r, c in stages],
              "identity_epochs": identity_epochs, "unsupervised_epochs": len(unsup), "train_wall_s": round(wall, 2),
              "ms_per_step_incl_feed_and_checkpoints": round(1e3 * wall / max(1, len(history) * steps_per_epoch), 3),
              "graph_replayed_steps": graph_steps, "feed": feed_report, "hip_graph_auto": probes,
              "identity_loss_per_epoch": [round(h["loss_epoch"], 6) for h in history if not h["unsupervised"]],
              "unsupervised_loss_per_epoch": curve,
              "loss_po2pl_per_epoch": [round(h["loss_po2pl_epoch"], 6) for h in unsup], "loss_pl2pl_per_epoch": [round(h["loss_pl2pl_epoch"], 6) for h in unsup]}
    if curve:
        k = max(1, len(curve) // 10)
        result["loss_first_epochs_mean"] = round(float(np.mean(curve[:k])), 6)
        result["loss_plateau_last_epochs_mean"] = round(float(np.mean(curve[-k:])), 6)
        result["loss_30_epoch_means"] = [round(float(np.mean(curve[i:i + 30])), 4) for i in range(0, len(curve), 30)]
    result["train_sequences"] = evaluate(device, tree, truth, train_ids[:2], ckpt, precision, out_dir, name + "_train", H, W)
    result["held_out_sequence"] = evaluate(device, tree, truth, [held_out], ckpt, precision, out_dir, name + "_heldout", H, W)[held_out]
    return result


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=300, help="in all: the last third runs at lr / 10, resumed from the first stage's checkpoint")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sequences", type=int, default=4, help="training sequences (one more is generated and held out)")
    ap.add_argument("--scans", type=int, default=41, help="scans per sequence")
    ap.add_argument("--precisions", default="float32,bfloat16,float16")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--hip-graph", default="auto", help="auto | true | false (config key hip_graph)")
    ap.add_argument("--trunk-segments", default="", help="mono | layer | block: how the trunk is cut into autograd Functions (ring_conv.TRUNK_SEGMENTS; "
                    "layer is what a DDP rank uses -- same kernels, other slab boundaries in the merged weight-gradient launches)")
    ap.add_argument("--anneal-fraction", type=float, default=1.0 / 3.0)
    ap.add_argument("--on-disk", action="store_true", help="store_dataset_in_RAM: false (the YAML's default) instead of the RAM-resident dataset")
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence.json"))
    args = ap.parse_args(argv)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.trunk_segments:
        from delora_amd.models import ring_conv
        ring_conv.TRUNK_SEGMENTS = args.trunk_segments
    t0 = time.perf_counter()
    tree, truth = build_dataset(device, args.sequences, args.scans)
    out_dir = tempfile.mkdtemp(prefix="delora_conv_out_")
    report = {"what": "identity pre-training -> unsupervised training -> Tester -> KITTI-style relative errors, per precision, same seed",
              "image": "64x720", "batch": args.batch, "learning_rate": args.lr, "reference_learning_rate": 1e-5,
              "dataset": f"{args.sequences} training sequences + 1 held out, {args.scans} scans each (synthetic scenes, ~0.45 m and <=1.5 deg yaw per scan), "
                         f"preprocessed offline at 64x2250; generation {time.perf_counter() - t0:.0f} s",
              "segment_lengths_m": [2.0, 5.0, 10.0, 15.0], "hip_graph": args.hip_graph, "trunk_segments": args.trunk_segments or "mono",
              "epochs": args.epochs, "anneal_fraction": args.anneal_fraction, "runs": {}}
    try:
        for precision in args.precisions.split(","):
            report["runs"][precision] = train_and_test(device, tree, truth, precision, args.epochs, args.lr, args.batch, args.seed, out_dir,
                                                       extra={"store_dataset_in_RAM": not args.on_disk, "num_dataloader_workers": args.workers,
                                                              "hip_graph": {"true": True, "false": False}.get(args.hip_graph, "auto")},
                                                       anneal_fraction=args.anneal_fraction)
            r = report["runs"][precision]
            print(precision, "steps", r["steps"], "loss", r.get("loss_first_epochs_mean"), "->", r.get("loss_plateau_last_epochs_mean"),
                  "held-out", r["held_out_sequence"]["translation_error_percent"], "%", r["held_out_sequence"]["rotation_error_deg_per_m"], "deg/m", flush=True)
        base = report["runs"].get("float32")
        if base:
            for p, r in report["runs"].items():
                if p != "float32":
                    r["vs_float32"] = {"held_out_translation_error_ratio": round(r["held_out_sequence"]["translation_error_percent"] / max(base["held_out_sequence"]["translation_error_percent"], 1e-9), 3),
                                       "held_out_rotation_error_ratio": round(r["held_out_sequence"]["rotation_error_deg_per_m"] / max(base["held_out_sequence"]["rotation_error_deg_per_m"], 1e-9), 3),
                                       "plateau_loss_ratio": round(r["loss_plateau_last_epochs_mean"] / base["loss_plateau_last_epochs_mean"], 4) if "loss_plateau_last_epochs_mean" in r and "loss_plateau_last_epochs_mean" in base else None}
    finally:
        shutil.rmtree(tree, ignore_errors=True)
        shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print("written", args.out)
    return report


if __name__ == "__main__":
    main()
