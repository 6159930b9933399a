import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from delora_amd.models import ring_conv
mono = len(sys.argv) > 1 and sys.argv[1] == 'mono'
ring_conv.TRUNK_SEGMENTS = 'mono' if mono else 'block'
args = bench.parse(["--rotate", "8"])
dev = torch.device('cuda:0')
cfg = bench.build_config(args, dev)
torch.manual_seed(1234)
host = bench.derived_batches(bench.make_batch(args, 0), 8, 0)
batches = [bench.to_device(b, dev) for b in host]
from delora_amd.deploy.trainer import Trainer
from delora_amd.data.dataset import ListDataset
tr = Trainer(cfg, dataset=ListDataset([d for b in host for d in b]))
bench.identity_pretrained_state(tr.raw_model)
def step(i):
    tr.optimizer.zero_grad(set_to_none=True)
    tr.step(preprocessed_dicts=[dict(s) for s in batches[i % 8]], epoch_losses=tr.new_epoch_losses())
for i in range(8): step(i)
torch.cuda.synchronize()
s0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
for i in range(20): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
s1 = torch.cuda.memory_stats()
print('mono' if mono else 'per-block', 'enqueue ms/step', 1e3*(t1-t0)/20, 'wall ms/step', 1e3*(t2-t0)/20,
      'device allocs', s1['num_device_alloc']-s0['num_device_alloc'], 'device frees', s1['num_device_free']-s0['num_device_free'],
      'reserved GB', s1['reserved_bytes.all.current']/1e9, 'alloc retries', s1['num_alloc_retries']-s0['num_alloc_retries'])
# phases: time forward+geometry vs backward on the host
import torch.autograd.profiler as prof
