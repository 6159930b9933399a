#!/usr/bin/env python3
"""k_icp_loss with its operands resident in the caches (back-to-back launches) vs right after the correspondence search
and vs cold, from the kernel's own begin/end timestamps (dl_icp_loss_partial_timed).  usage: python tools/loss_warm.py [reps]"""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
X = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))()
cfg = bench.build_config(X, dev)
batch = bench.make_batch(X, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
geo = HipStepGeometry()
prep = geo.prepare(batch, sensor, (3, 5, 0.5, 10))
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
T = torch.eye(4, device=dev).repeat(8, 1, 1)
flags = G.LOSS_POINT_TO_PLANE | G.LOSS_PLANE_TO_PLANE
nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)


def run(pre):
    timers = G.LossTimers()
    G.LOSS_TIMER_FACTORY = timers.new
    for _ in range(reps):
        pre()
        G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, flags)
    torch.cuda.synchronize()
    G.LOSS_TIMER_FACTORY = None
    ms = timers.elapsed_ms()
    timers.close()
    return statistics.median(ms) * 1e3


flush = torch.empty(128_000_000, device=dev)
def search():
    global nn, match
    nn, _, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
run(lambda: None)
print(f"k_icp_loss back to back (operands cached): {run(lambda: None):.2f} us")
print(f"k_icp_loss right after the search:          {run(search):.2f} us")
print(f"k_icp_loss after 0.5 GB of unrelated traffic: {run(lambda: flush.add_(1.0)):.2f} us")
