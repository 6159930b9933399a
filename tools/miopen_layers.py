#!/usr/bin/env python3
"""Per-layer time of the library (MIOpen) convolutions torch picks for the pose CNN's layer shapes, fp32: forward,
backward-data and backward-weight separately, NCHW (the round-1 path, circular pad materialised by the caller) and
channels_last.  The numbers tools/bin/conv_harness has to beat."""
import sys
import torch

B = 8
LAYERS = [("layer1 3x3 64->64", 64, 512, 64, 64, 3, (1, 1)), ("layer2.0.conv1 s(1,2)", 64, 512, 64, 128, 3, (1, 2)),
          ("layer2.0.ds 1x1 s(1,2)", 64, 512, 64, 128, 1, (1, 2)), ("layer2 3x3 128->128", 64, 256, 128, 128, 3, (1, 1)),
          ("layer3.0.conv1 s(1,2)", 64, 256, 128, 256, 3, (1, 2)), ("layer3.0.ds 1x1 s(1,2)", 64, 256, 128, 256, 1, (1, 2)),
          ("layer3 3x3 256->256", 64, 128, 256, 256, 3, (1, 1)), ("layer4.0.conv1 s(2,2)", 64, 128, 256, 512, 3, (2, 2)),
          ("layer4.0.ds 1x1 s(2,2)", 64, 128, 256, 512, 1, (2, 2)), ("layer4 3x3 512->512", 32, 64, 512, 512, 3, (1, 1))]


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        tot = 0.0
        for name, H, W, C, K, ks, st in LAYERS:
            wp = W + 2 if ks == 3 else W
            x = torch.randn(B, C, H, wp, device=dev).contiguous(memory_format=fmt)
            w = (torch.randn(K, C, ks, ks, device=dev) * 0.05).contiguous(memory_format=fmt)
            pad = (1, 0) if ks == 3 else (0, 0)
            y = torch.nn.functional.conv2d(x, w, stride=st, padding=pad)
            g = torch.randn_like(y).contiguous(memory_format=fmt)
            flop = 2.0 * y.numel() * C * ks * ks
            f = timed(lambda: torch.nn.functional.conv2d(x, w, stride=st, padding=pad), reps)
            d = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, st, pad, (1, 1), False, (0, 0), 1, (True, False, False)), reps)
            wg = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, st, pad, (1, 1), False, (0, 0), 1, (False, True, False)), reps)
            tot += f + d + wg
            print(f"miopen {fmt_name} {name:26s} fwd {f:8.1f} us {flop / f * 1e-6:6.1f} TF | dgrad {d:8.1f} us {flop / d * 1e-6:6.1f} TF | "
                  f"wgrad {wg:8.1f} us {flop / wg * 1e-6:6.1f} TF")
        print(f"miopen {fmt_name} sum {tot:.1f} us")


if __name__ == "__main__":
    main()
