#!/usr/bin/env python3
"""Which parameter does a captured forward pass of the full-size network NOT re-read?  Capture model(stacked) once, then for
every parameter: scale it in place, replay the graph, compare with an eager forward, restore."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from delora_amd.deploy.trainer import Trainer
from delora_amd.data.dataset import ListDataset
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
args = bench.parse(["--batch", "8", "--rotate", "1"])
cfg = bench.build_config(args, dev)
if len(sys.argv) > 1:
    cfg["cnn_impl"] = sys.argv[1]
torch.manual_seed(1234)
host = bench.make_batch(args, 0)
tr = Trainer(cfg, dataset=ListDataset(list(host)))
batch = bench.to_device(host, dev)
model = tr.raw_model
sensor = tr.img_projection.sensor("kitti")
prep = tr.geo.prepare(batch, sensor, tr._normal_params("kitti"))
x = prep["stacked"].clone()


def fwd():
    with torch.no_grad():
        t, q = model(x)
    return torch.cat((t, q), dim=1)


side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
g.replay(); torch.cuda.synchronize()
print("baseline graph vs eager:", float((out - fwd()).abs().max()), flush=True)
for k, p in model.named_parameters():
    with torch.no_grad():
        p.mul_(1.25)
    g.replay(); torch.cuda.synchronize()
    e = fwd()
    d, change = float((out - e).abs().max()), float((e - out).abs().max())
    print(f"{k:45s} graph-vs-eager {d:.3e}  graph {[round(v, 4) for v in out[0, :4].tolist()]} eager {[round(v, 4) for v in e[0, :4].tolist()]}", "  <-- STALE" if d > 1e-5 else "", flush=True)
    if k.endswith("layer1.0.conv2.weight"):
        break
    with torch.no_grad():
        p.div_(1.25)
