#!/usr/bin/env python3
"""How the loss kernel's cold time depends on what flushed the caches: 1 GiB of WRITES (bench.py's flush: the infinity cache is left
full of dirty lines whose write-back competes with the kernel's reads) against 1 GiB of READS (clean lines), and the batch size
(the ramp of a 54 MB launch).  usage: python tools/exp/loss_cold.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import delora_amd._lib as _L
if len(sys.argv) > 1: _L.LIB_PATH = sys.argv[1]
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry

dev = torch.device("cuda:0")
flush = torch.zeros((256 * 1024 * 1024,), dtype=torch.float32, device=dev)
sink = torch.zeros((1,), device=dev)
flush2 = torch.zeros((256 * 1024 * 1024,), dtype=torch.float32, device=dev)
for B in ((8,) if len(sys.argv) > 2 else (8, 16)):
    A = type("A", (), dict(batch=B, height=64, width=2048))
    cfg = bench.build_config(type("X", (), dict(height=64, width=2048, batch=B, amp="", channels_last=False))(), dev)
    batch = bench.make_batch(A(), 0, dev)
    sensor = G.Sensor.from_config(cfg, "kitti")
    geo = HipStepGeometry()
    prep = geo.prepare(batch, sensor, (3, 5, 0.5, 10))
    img, nrm = prep["images"], prep["normals"]
    T = torch.eye(4, device=dev).repeat(B, 1, 1); T[:, 0, 3] = 0.4
    nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], prep["packed"][:, 0], prep["normals_packed"][:, 0], T, sensor)
    flags = G.LOSS_POINT_TO_PLANE | G.LOSS_PLANE_TO_PLANE
    M = int((nn >= 0).sum())
    for name, fl in (("write flush (fill_)", lambda i: flush.fill_(float(i))), ("read flush (sum)", lambda i: sink.add_(flush.sum())),
                     ("read flush + 200 us idle", lambda i: (sink.add_(flush.sum()), torch.cuda._sleep(400000))), ("write flush, then 1 GiB of reads (other buffer)", lambda i: (flush.fill_(float(i)), sink.add_(flush2.sum()))),
                     ("write flush, then 2 x 1 GiB of reads", lambda i: (flush.fill_(float(i)), sink.add_(flush2.sum()), sink.add_(flush2.sum()))),
                     ("warm", lambda i: None)):
        timers = G.LossTimers(reserve=24)
        G.LOSS_TIMER_FACTORY = timers.new
        try:
            for i in range(20):
                fl(i)
                G.icp_loss(T, img[:, 1], nrm[:, 1], match, nn, flags)
            torch.cuda.synchronize()
        finally:
            G.LOSS_TIMER_FACTORY = None
        ms = np.array(timers.elapsed_ms()[4:]); timers.close()
        print(f"B={B:2d} {52 * M / 1e6:6.1f} MB  {name:50s} median {1e3 * np.median(ms):6.2f} us  min {1e3 * ms.min():6.2f} us  -> {52 * M / np.median(ms) / 1e9:5.2f} TB/s ({52 * M / np.median(ms) / 1e9 / 8:.3f} of 8)")
    del prep, img, nrm, nn, vis, match, batch
