#!/usr/bin/env python3
"""Capture single entry points of the convolution path into a HIP graph; replay with changed inputs; compare with eager."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from delora_amd.models import ring_conv as rc
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
g0 = torch.Generator().manual_seed(0)
x = torch.randn((8, 64, 512, 64), generator=g0).to(dev)
w = (torch.randn((64, 64, 3, 3), generator=g0) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
w2 = (torch.randn((128, 64, 3, 3), generator=g0) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
x8 = torch.randn((8, 8, 64, 2048), generator=g0).to(dev)
w1 = (torch.randn((64, 8, 3, 3), generator=g0) * 0.05).to(dev)


def wino():
    uf, _ = rc.wino_weights(w, want_bwd=False)
    return rc.wino_conv(x, uf, 64, act=1, epilogue=rc.EPI_ACT)


CASES = {
    "conv_nhwc stride1 direct": lambda: rc.conv_nhwc(x, rc.weight_storage(w), act=1, epilogue=rc.EPI_ACT),
    "conv_nhwc stride(1,2)": lambda: rc.conv_nhwc(x, rc.weight_storage(w2), stride=(1, 2), act=1, epilogue=rc.EPI_ACT),
    "wino_weights + wino_conv": wino,
    "pool_fwd": lambda: rc.pool_fwd(x)[0],
    "stem": lambda: rc.RingStem.apply(x8, w1, 1),
    "torch tanh(x) (control)": lambda: torch.tanh(x),
}
for name, fn in CASES.items():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    res = []
    for i in range(4):
        with torch.no_grad():
            x.mul_(1.1); w.mul_(1.05); w2.mul_(1.05); x8.mul_(1.1); w1.mul_(1.05)
        g.replay(); torch.cuda.synchronize()
        res.append(float((out - fn()).abs().max()))
    print(f"{name:28s} graph-vs-eager over 4 replays with changing inputs: {res}", flush=True)

# ---- composition: the trunk Function and the whole network
import bench
from delora_amd.models.model import OdometryModel
args = bench.parse(["--batch", "8"])
cfg = bench.build_config(args, dev)
torch.manual_seed(3)
model = OdometryModel(cfg).to(dev)
model.resnet.trunk_weights_channels_last()
res_net = model.resnet
blocks, weights = res_net._trunk_blocks()
x0 = torch.tanh(torch.randn((8, 64, 512, 64), generator=g0)).to(dev)
xin = torch.randn((8, 8, 64, 2048), generator=g0).to(dev)


def trunk(nblocks):
    nw = sum(3 if b[3] else 2 for b in blocks[:nblocks])
    with torch.no_grad():
        return rc.RingTrunk.apply(x0, 1, blocks[:nblocks], *weights[:nw])


def net():
    with torch.no_grad():
        t, q = model(xin)
    return torch.cat((t, q), dim=1)


CASES2 = {"trunk 1 block": lambda: trunk(1), "trunk 2 blocks": lambda: trunk(2), "trunk 3 blocks (first strided)": lambda: trunk(3),
          "trunk all 8 blocks": lambda: trunk(8), "whole network": net}
params = list(model.parameters())
for name, fn in CASES2.items():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    res = []
    for i in range(4):
        with torch.no_grad():
            for p in params:
                p.mul_(1.01)
        g.replay(); torch.cuda.synchronize()
        res.append(float((out - fn()).abs().max()))
    print(f"{name:28s} graph-vs-eager over 4 replays with changing weights: {res}", flush=True)
