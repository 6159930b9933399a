#!/usr/bin/env python3
"""Which part of the full-size step does not survive a second replay of its captured HIP graph?
   python tools/exp/graph_bisect.py prepare|nn|loss|cnn|cnnbwd"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from delora_amd import geometry as G
from delora_amd.deploy.trainer import Trainer
from delora_amd.data.dataset import ListDataset
what = sys.argv[1]
args = bench.parse(["--steps", "1"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = bench.build_config(args, dev)
host = bench.make_batch(args, 0)
batch = bench.to_device(host, dev)
trainer = Trainer(cfg, dataset=ListDataset(list(host)))
sensor = trainer.img_projection.sensor("kitti")
params = trainer._normal_params("kitti")
prep = trainer.geo.prepare(batch, sensor, params)
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
B = img.shape[0]
T = torch.eye(4, device=dev).repeat(B, 1, 1); T[:, 0, 3] = 0.4
nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
flags = G.loss_flags(cfg)
model = trainer.raw_model


def fn():
    if what == "prepare":
        return trainer.geo.prepare(batch, sensor, params)["images"].sum()
    if what == "nn":
        return G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)[0].sum()
    if what == "loss":
        return G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, flags)[0].sum()
    if what == "cnn":
        with torch.no_grad():
            t, q = model(prep["stacked"])
        return t.sum() + q.sum()
    if what == "cnnbwd":
        for p in model.parameters():
            p.grad = None
        t, q = model(prep["stacked"])
        (t.square().sum() + q.sum()).backward()
        return t.sum()


side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fn()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    print(what, "replay", i, float(out), flush=True)
print(what, "OK", flush=True)
