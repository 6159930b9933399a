#!/usr/bin/env python3
"""Every convolution launch of one training step of the pose CNN (batch 8, 64x2048), ONE launch per (kernel, pass, layer shape), in a
fixed order, each bracketed by the library's launch profiler so that its profile-row name and kernel-launch count are known.
Meant to run under rocprofv3 (tools/prof_round.sh): `--kernel-trace --stats` gives per-kernel times, `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (separate passes) the HBM traffic; tools/conv_layers_pmc.py maps the dispatches back to the rows by order.

usage: python tools/conv_layers.py [float32|bfloat16|float16] [order.json]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from delora_amd import _lib
from delora_amd.models import ring_conv as rc

mode = sys.argv[1] if len(sys.argv) > 1 else "float32"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
B, H, W = 8, 64, 512                     # the pooled stem output of a 64x2048 input
half = getattr(torch, mode) if mode != "float32" else None
order = []


def run(what, fn):
    fn()                                  # warm (allocator, code objects)
    torch.cuda.synchronize()
    _lib.profile_begin(64)
    fn()
    torch.cuda.synchronize()
    rows, _ = _lib.profile_end()
    for r in rows:
        order.append({"op": what, "row": r["name"], "launches": r["launches"], "ms": r["ms"], "flop": r["flop"], "compulsory_bytes": r["bytes"]})


def t(shape, dtype=torch.float32):
    return torch.randn(shape, device=dev).to(dtype)


layers = [("layer1", H, W, 64, 64, (1, 1)), ("layer2.0", H, W, 64, 128, (1, 2)), ("layer2", H, W // 2, 128, 128, (1, 1)),
          ("layer3.0", H, W // 2, 128, 256, (1, 2)), ("layer3", H, W // 4, 256, 256, (1, 1)), ("layer4.0", H, W // 4, 256, 512, (2, 2)),
          ("layer4", H // 2, W // 8, 512, 512, (1, 1))]
for name, h, w, c, k, st in layers:
    ho, wo = h // st[0], w // st[1]
    wt = (torch.randn((k, c, 3, 3), device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    if half is None:
        x, g, gi = t((B, h, w, c)), t((B, ho, wo, k)), t((B, h, w, c))
        if st == (1, 1):
            uf, ub = rc.wino_weights(wt)
            run(name + " forward", lambda: rc.wino_conv(x, uf, k, act=1, epilogue=rc.EPI_ACT))
            run(name + " input gradient", lambda: rc.wino_conv(g, ub, c, act=1, epilogue=rc.EPI_DACT, dsrc=x))
            run(name + " weight gradient", lambda: rc.wgrad_nhwc(x, g, 3))
        else:
            ws, wd = rc.weight_storage(wt), (torch.randn((k, 1, 1, c), device=dev) * 0.05)
            run(name + " forward 3x3", lambda: rc.conv_nhwc(x, ws, stride=st, act=1, epilogue=rc.EPI_ACT))
            run(name + " forward 1x1", lambda: rc.conv_nhwc(x, wd, stride=st))
            run(name + " input gradient 1x1", lambda: rc.dgrad_strided(g, wd, st, (h, w), dense=True))
            run(name + " input gradient 3x3", lambda: rc.dgrad_strided(g, ws, st, (h, w), act=1, epilogue=rc.EPI_DACT, dsrc=gi))
            run(name + " weight gradient 3x3", lambda: rc.wgrad_nhwc(x, g, 3, stride=st))
            run(name + " weight gradient 1x1", lambda: rc.wgrad_nhwc(x, g, 1, stride=st))
    else:
        x, g, gi = t((B, h, w, c), half), t((B, ho, wo, k), half), t((B, h, w, c), half)
        wf, wb = rc.weights_h(wt, half)
        if st == (1, 1):
            run(name + " forward", lambda: rc.conv_nhwc_h(x, wf, 3, act=1, epilogue=rc.EPI_ACT))
            run(name + " input gradient", lambda: rc.conv_nhwc_h(g, wb, 3, act=1, epilogue=rc.EPI_DACT, dsrc=x, transposed=True))
            run(name + " weight gradient", lambda: rc.wgrad_nhwc_h(x, g, 3))
        else:
            wdp = (torch.randn((k, c, 1, 1), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
            wdf, wdb = rc.weights_h(wdp, half)
            run(name + " forward 3x3", lambda: rc.conv_nhwc_h(x, wf, 3, stride=st, act=1, epilogue=rc.EPI_ACT))
            run(name + " forward 1x1", lambda: rc.conv_nhwc_h(x, wdf, 1, stride=st))
            run(name + " input gradient 1x1", lambda: rc.dgrad_strided_h(g, wdb, 1, st, (h, w), dense=True))
            run(name + " input gradient 3x3", lambda: rc.dgrad_strided_h(g, wb, 3, st, (h, w), act=1, epilogue=rc.EPI_DACT, dsrc=gi))
            run(name + " weight gradient 3x3", lambda: rc.wgrad_nhwc_h(x, g, 3, stride=st))
            run(name + " weight gradient 1x1", lambda: rc.wgrad_nhwc_h(x, g, 1, stride=st))
doc = {"workload": f"one launch per convolution kernel / pass / layer shape of the pose CNN's trunk, batch {B}, 64x2048 input, {mode}", "order": order}
if out_path:
    json.dump(doc, open(out_path, "w"), indent=1)
for o in order:
    print(f"{o['op']:34s} {o['row']:52s} x{o['launches']}  {1e3 * o['ms']:8.1f} us  {o['flop'] / max(o['ms'], 1e-9) * 1e-9:7.1f} TFLOP/s")
