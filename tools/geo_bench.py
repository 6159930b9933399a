#!/usr/bin/env python3
"""Micro-benchmark of the geometry kernels on the bench batch (B=8, 64x2048); meant to run under rocprofv3.
usage: python tools/geo_bench.py [reps] [shift_m] [point_order]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shift = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
class A: batch=8; height=64; width=2048; point_order = sys.argv[3] if len(sys.argv) > 3 else "raster"   # raster (bench default) | shuffled | firing
dev = torch.device("cuda:0")
args = A()
cfg = bench.build_config(argparse_like := type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))(), dev)
batch = bench.make_batch(args, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
geo = HipStepGeometry()
prep = geo.prepare(batch, sensor, (3, 5, 0.5, 10))
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
B = 8
T = torch.eye(4, device=dev).repeat(B, 1, 1); T[:, 0, 3] = shift
flags = G.LOSS_POINT_TO_PLANE | G.LOSS_PLANE_TO_PLANE
for _ in range(reps):
    prep = geo.prepare(batch, sensor, (3, 5, 0.5, 10))
    nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
    terms, counts = G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, flags)
# calibration of the loss kernel: its read-only twin (same 13 streams, no arithmetic) and the kernel itself,
#   cold behind DIRTY foreign lines (1 GB of unrelated read-modify-write just before: the infinity cache is full of dirty lines whose
#        write-back shares the HBM with the kernel),
#   back to back ("warm"),
#   cold behind CLEAN foreign lines (the same, followed by 1 GiB of unrelated reads, which push the dirty lines out)
flush = torch.empty(128_000_000, device=dev)
flush2 = torch.zeros(256 * 1024 * 1024, device=dev)
sink = torch.zeros(1, device=dev)
for _ in range(reps):
    flush.add_(1.0); G.probe_stream_read(img[:, 1], nrm[:, 1], match, nn)
for _ in range(reps):
    G.probe_stream_read(img[:, 1], nrm[:, 1], match, nn)
for _ in range(reps):
    flush.add_(1.0); G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, flags)
for _ in range(reps):
    flush.add_(1.0); sink.add_(flush2.sum()); G.probe_stream_read(img[:, 1], nrm[:, 1], match, nn)
for _ in range(reps):
    flush.add_(1.0); sink.add_(flush2.sum()); G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, flags)
del flush2
del flush
# calibration: plain torch streaming kernels over the same number of bytes as the loss kernel moves (53 MB)
xa = torch.randn(53_000_000 // 4, device=dev); xb = torch.empty(53_000_000 // 8, device=dev); xc = torch.randn(53_000_000 // 8, device=dev)
for _ in range(reps):
    xa.sum(); xb.copy_(xc)
torch.cuda.synchronize()
ws = torch.empty(64, dtype=torch.int32, device=dev)
print("pairs", counts[:, 0].tolist(), "terms", terms[0].tolist())
