"""Where the host spends its time enqueuing one eager training step (cProfile over 20 steps of the bench workload)."""
import cProfile, pstats, sys, time, io, torch
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
def step(i):
    tr.optimizer.zero_grad(set_to_none=True)
    tr.step(preprocessed_dicts=[dict(s) for s in batches[i % 8]], epoch_losses=tr.new_epoch_losses())
for i in range(8): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(20): step(i)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.3f} ms/step (under cProfile), wall {1e3*(t2-t0)/20:.3f}")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
