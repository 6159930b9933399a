#!/usr/bin/env python3
"""Robustness probe: correspondence search + loss with NaN / inf / huge poses must not fault."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry
args = bench.parse(["--steps", "1"])
dev = torch.device("cuda:0")
cfg = bench.build_config(args, dev)
batch = bench.to_device(bench.make_batch(args, 0), dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
B = img.shape[0]
for name, val in (("nan", float("nan")), ("inf", float("inf")), ("huge", 1e30), ("neg-huge", -1e30)):
    T = torch.eye(4, device=dev).repeat(B, 1, 1)
    T[0, 0, 3] = val; T[1, 1, 1] = val; T[2, :3, :3] = val
    nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
    terms, counts = G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, G.LOSS_POINT_TO_PLANE | G.LOSS_PLANE_TO_PLANE)
    torch.cuda.synchronize()
    print(name, "pairs", counts[:, 0].tolist(), "terms[3]", terms[3].tolist(), flush=True)
print("ok")
