#!/usr/bin/env python3
"""Time dl_nn_correspond on the bench batch (B=8, 64x2048, T = I as in the bench) and compare its result with a saved
one: python tools/nn_time.py [reps] [save|check path]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
if os.environ.get("DL_LIB"):
    from delora_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["DL_LIB"])
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode, path = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else (None, None)
dev = torch.device("cuda:0")
X = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False, cnn=""))()
cfg = bench.build_config(X, dev)
batch = bench.make_batch(X, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
T = torch.eye(4, device=dev).repeat(8, 1, 1)
for _ in range(3):
    nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
b.record(); torch.cuda.synchronize()
msg = ""
if mode == "save":
    torch.save(nn.cpu(), path)
elif mode == "check":
    msg = " identical" if torch.equal(nn.cpu(), torch.load(path)) else " DIFFERENT"
print(f"nn_correspond: {a.elapsed_time(b) / reps:.3f} ms per call{msg}")
