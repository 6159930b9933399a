"""Micro-benchmark of the fused elementwise kernels of the pose CNN at the tensor sizes of the KITTI step (B=8):
   python tools/ring_bench.py [reps]    -> us per launch and effective GB/s (bytes the op must move / time)."""
import sys

import torch

from delora_amd.models.ring_ops import ring_act_pad, ring_act_pool_pad


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    B = 8
    cases = [("layer1 act+pad", (B, 64, 64, 512), None), ("layer1 add+act+pad", (B, 64, 64, 512), "padded"),
             ("layer2 add+act+pad", (B, 128, 64, 256), "padded"), ("layer3 add+act+pad", (B, 256, 64, 128), "dense"),
             ("layer4 add+act+pad", (B, 512, 32, 64), "padded")]
    for name, shape, res_kind in cases:
        x = torch.randn(shape, device=dev, requires_grad=True)
        N, C, H, W = shape
        res = None
        if res_kind == "padded":
            res = torch.randn((N, C, H, W + 2), device=dev, requires_grad=True)
        elif res_kind == "dense":
            res = torch.randn(shape, device=dev, requires_grad=True)
        n = x.numel()
        out = ring_act_pad(x, "tanh", pad=True, residual=res)
        g = torch.randn_like(out)
        t_f = timed(lambda: ring_act_pad(x, "tanh", pad=True, residual=res), reps)
        t_b = timed(lambda: torch.autograd.grad(out, [x] + ([res] if res is not None else []), g, retain_graph=True), reps)
        by_f = 4 * n * (2 + (res is not None))
        by_b = 4 * n * (3 + (res_kind == "padded"))
        print(f"{name:22s} {shape}: fwd {t_f:7.1f} us {by_f / t_f / 1e3:7.0f} GB/s   bwd {t_b:7.1f} us {by_b / t_b / 1e3:7.0f} GB/s")
    x = torch.randn((B, 64, 64, 1024), device=dev, requires_grad=True)
    out = ring_act_pool_pad(x, "tanh")
    g = torch.randn_like(out)
    t_f = timed(lambda: ring_act_pool_pad(x, "tanh"), reps)
    t_b = timed(lambda: torch.autograd.grad(out, [x], g, retain_graph=True), reps)
    n = x.numel()
    print(f"stem act+pool+pad      {tuple(x.shape)}: fwd {t_f:7.1f} us {4 * n * 1.5 / t_f / 1e3:7.0f} GB/s   "
          f"bwd {t_b:7.1f} us {4 * n * 2.0 / t_b / 1e3:7.0f} GB/s")
    # the same tensors through torch's own elementwise kernels, for scale
    y = torch.empty_like(x)
    t_c = timed(lambda: torch.tanh(x.detach(), out=y), reps)
    print(f"torch.tanh (read+write) {tuple(x.shape)}: {t_c:7.1f} us {8 * n / t_c / 1e3:7.0f} GB/s")
    t_c = timed(lambda: y.copy_(x.detach()), reps)
    print(f"copy (read+write)       {tuple(x.shape)}: {t_c:7.1f} us {8 * n / t_c / 1e3:7.0f} GB/s")


if __name__ == "__main__":
    main()
