#!/usr/bin/env python3
"""Per-query cost of pass B of the correspondence search (debug build tools/bin/libdelora_prof.so, -DNN_PROFILE)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from delora_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libdelora_prof.so")
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry
dev = torch.device("cuda:0")
X = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False, cnn=""))()
cfg = bench.build_config(X, dev)
batch = bench.make_batch(X, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
T = torch.eye(4, device=dev).repeat(8, 1, 1)
lib = _lib.load()
prof = torch.zeros((2 * 8 * 64 * 2048,), dtype=torch.int32, device=dev)
lib.dl_nn_debug_set.argtypes = [ctypes.c_void_p]
print("set", lib.dl_nn_debug_set(ctypes.c_void_p(prof.data_ptr())))
nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(-1, 2)
n = int((p[:, 0] > 0).sum())
cyc, win = p[:n, 0].astype(np.float64), p[:n, 1]
print(f"hard queries {n} of {int((nn >= 0).sum())}; cycles per query: mean {cyc.mean():.0f} median {np.median(cyc):.0f} p90 {np.percentile(cyc, 90):.0f} "
      f"p99 {np.percentile(cyc, 99):.0f} max {cyc.max():.0f}; sum/8192 waves {cyc.sum() / 8192:.0f} cycles")
print(f"window pixels: mean {win.mean():.0f} median {np.median(win):.0f} p99 {np.percentile(win, 99):.0f} max {win.max()}")
for lo, hi in ((0, 256), (256, 1024), (1024, 4096), (4096, 16384), (16384, 1 << 30)):
    m = (win >= lo) & (win < hi)
    if m.any():
        print(f"  window {lo:6d}..{hi:<10d} {m.sum():7d} queries, mean {cyc[m].mean():8.0f} cycles, share of total time {cyc[m].sum() / cyc.sum():.2f}")
