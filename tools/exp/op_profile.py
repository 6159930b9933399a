#!/usr/bin/env python3
"""Which torch ops of one training step launch the small kernels?  torch.profiler over one steady-state step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from delora_amd.deploy.trainer import Trainer
from delora_amd.data.dataset import ListDataset
from torch.profiler import profile, ProfilerActivity
args = bench.parse(["--batch", "8"]); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = bench.build_config(args, dev); torch.manual_seed(1234)
host = bench.make_batch(args, 0); batch = bench.to_device(host, dev)
tr = Trainer(cfg, dataset=ListDataset(list(host))); bench.identity_pretrained_state(tr.raw_model)


def step():
    tr.optimizer.zero_grad(set_to_none=True)
    tr.step(preprocessed_dicts=[dict(s) for s in batch], epoch_losses=tr.new_epoch_losses())


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.count:4d} x {e.key:32s} dev {e.device_time_total:8.1f} us  shapes {str(e.input_shapes)[:110]}")
