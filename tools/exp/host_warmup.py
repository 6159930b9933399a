"""Host enqueue time of each of the first steps of a process (no synchronisation in between): how long the host needs to warm up.
usage: python tools/exp/host_warmup.py [--amp bfloat16]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
args = bench.parse(["--rotate", "8"] + sys.argv[1:])
cfg = bench.build_config(args, dev)
torch.manual_seed(1234)
host = bench.derived_batches(bench.make_batch(args, 0), 8, 0)
batches = [bench.to_device(b, dev) for b in host]
from delora_amd.deploy.trainer import Trainer
from delora_amd.data.dataset import ListDataset
tr = Trainer(cfg, dataset=ListDataset([d for b in host for d in b]))
bench.identity_pretrained_state(tr.raw_model)
import os
if os.environ.get('EV'):
    from delora_amd import geometry as G
    keep = G.LossTimers(reserve=int(os.environ['EV']))
if os.environ.get('IMPL'):
    print(bench.cnn_impl_in_use(tr, args))
ts = []
for i in range(64):
    t0 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True)
    tr.step(preprocessed_dicts=[dict(s) for s in batches[i % 8]], epoch_losses=tr.new_epoch_losses())
    ts.append(1e3 * (time.perf_counter() - t0))
    if i == 12:
        torch.cuda.synchronize()
print(" ".join(f"{t:.1f}" for t in ts))
print("memory stats: num_alloc_retries", torch.cuda.memory_stats()["num_alloc_retries"], "segments", torch.cuda.memory_stats()["segment.all.current"])
