"""A/B of the trunk as one autograd Function vs one per block: GPU-timeline duration of forward and backward of the CNN alone
(HIP events on the stream, no profiler attached)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from delora_amd.models import ring_conv
from delora_amd.models.model import OdometryModel
dev = torch.device('cuda:0')
args = bench.parse([])
cfg = bench.build_config(args, dev)
torch.manual_seed(1)
m = OdometryModel(cfg).to(dev)
m.resnet.trunk_weights_channels_last()
x = torch.randn((8, 8, 64, 2048), device=dev)
for mode in ("mono", "per-block", "mono", "per-block"):
    ring_conv.TRUNK_SEGMENTS = "block" if mode == "per-block" else "mono"
    for it in range(3):
        m.zero_grad(set_to_none=True)
        t, q = m(x); (t.sum() + q.sum()).backward()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * 10)]
    h0 = time.perf_counter()
    for it in range(10):
        m.zero_grad(set_to_none=True)
        ev[3 * it].record()
        t, q = m(x)
        loss = t.sum() + q.sum()
        ev[3 * it + 1].record()
        loss.backward()
        ev[3 * it + 2].record()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    f = sum(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(2, 10)) / 8
    b = sum(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(2, 10)) / 8
    print(f"{mode:10s} forward {f:.3f} ms  backward {b:.3f} ms  host enqueue {1e3 * (h1 - h0) / 10:.3f} ms/iter  wall {1e3 * (h2 - h0) / 10:.3f} ms/iter")
