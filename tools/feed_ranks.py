#!/usr/bin/env python3
"""Eight data-parallel ranks' worth of input feed on ONE host: R processes (one per rank, as torchrun starts them), each with its own
`PackedFeed` over the rank's `DistributedSampler` shard of one on-disk tree in the reference's format and W worker processes --
R x (1 + W) processes on this host's CPU quota.  No GPU work: a consumer takes every batch off its feed as fast as it comes (CPU
device: the slots are not page-locked, the "upload" is a host copy), so the figure is what the HOST side of the feed sustains per rank
when all ranks run at once -- to be held against the pairs/s one GPU consumes (bench.py: value / n_gpus).

    python tools/feed_ranks.py [--ranks 8] [--workers 2] [--batch 8] [--scans 33] [--epochs 6] [--out FILE]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rank(rank, ranks, workers, batch, tree, epochs, rings, cells, ret):
    from delora_amd import config as cfgmod
    from delora_amd.data import feed
    from delora_amd.data.dataset import PreprocessedPointCloudDataset
    torch.set_num_threads(1)
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfgmod.degrees_to_radians(cfg)
    cfg["kitti"].update(preprocessed_path=tree, data_identifiers=[0], vertical_cells=rings, horizontal_cells_preprocessing=cells)
    cfg.update(store_dataset_in_RAM=False, num_dataloader_workers=workers, load_normal_lists=False)
    ds = PreprocessedPointCloudDataset(cfg)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=ranks, rank=rank, shuffle=True, drop_last=True)
    batches = torch.utils.data.BatchSampler(sampler, batch_size=batch, drop_last=True)
    pf = feed.PackedFeed(ds, batches, batch, torch.device("cpu"), workers=workers)
    try:
        seen, points = 0, 0
        sampler.set_epoch(0)
        for b in pf:                                         # workers up, page cache warm
            pass
        t0 = time.perf_counter()
        for e in range(1, epochs + 1):
            sampler.set_epoch(e)
            for b in pf:
                seen += 1
                points += int(b.offs[-1])
        el = time.perf_counter() - t0
        ret[rank] = {"batches": seen, "pairs_per_s": batch * seen / el, "points": points, "seconds": el}
    finally:
        pf.close()


def run(ranks=8, workers=2, batch=8, scans=33, epochs=6, rings=64, cells=2250, tree=None):
    from delora_amd.data import synthetic
    own = tree is None
    if own:
        tree = tempfile.mkdtemp(prefix="delora_feed_ranks_")
        seq, _ = synthetic.make_sequence(4200, scans, rings=rings, azimuth_steps=cells, point_order="raster")
        synthetic.write_tree(tree, seq, sequence=0)
    try:
        ret = mp.Manager().dict()
        t0 = time.perf_counter()
        mp.spawn(_rank, args=(ranks, workers, batch, tree, epochs, rings, cells, ret), nprocs=ranks, join=True)
        wall = time.perf_counter() - t0
        per = [ret[r]["pairs_per_s"] for r in range(ranks)]
        quota = None
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else round(int(q) / int(p), 2)
        except (OSError, ValueError):
            pass
        return {"ranks": ranks, "workers_per_rank": workers, "processes": ranks * (1 + workers), "batch": batch, "host_cpus": os.cpu_count(),
                "cgroup_cpu_quota": quota, "pairs_per_s_per_rank_min": round(min(per), 1), "pairs_per_s_per_rank_mean": round(float(np.mean(per)), 1),
                "pairs_per_s_all_ranks": round(sum(per), 1), "batches_per_rank": int(ret[0]["batches"]), "wall_s": round(wall, 2),
                "dataset": f"{scans - 1} consecutive pairs, {rings}x{cells} scans, xyz only, the reference's on-disk layout, page cache warm",
                "note": "host side only (CPU device): what the feed processes of all ranks sustain together on this host"}
    finally:
        if own:
            import shutil
            shutil.rmtree(tree, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--scans", type=int, default=65)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    out = run(a.ranks, a.workers, a.batch, a.scans, a.epochs)
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
