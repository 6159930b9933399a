#!/usr/bin/env python3
"""The reference's default operating point (unmodified YAML: 64x720, batch B, stored point + normal lists) as a bare loop of resident
steps, for kernel traces:   rocprofv3 --kernel-trace -- python tools/shipped_step.py B eager|graph STEPS [amp]
(bench.py: live_kernel_sum cuts the trace into steps and sums the kernel durations; tools/step_breakdown.py prints one step)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B, mode, steps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    amp = sys.argv[4] if len(sys.argv) > 4 else ""
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tree = bench.shipped_tree(device, max(8, 2 * B))
    try:
        trainer = bench.shipped_trainer(device, tree["path"], B, workers=0, hip_graph="off", amp=amp)
        batches = bench.shipped_resident_batches(trainer, 4)
        i = 0

        def eager():
            nonlocal i
            i += 1
            trainer.optimizer.zero_grad(set_to_none=True)
            trainer.step(preprocessed_dicts=[dict(d) for d in batches[i % len(batches)]], epoch_losses=trainer.new_epoch_losses())
        for _ in range(6):
            eager()
        run = eager
        if mode == "graph":
            from delora_amd.deploy.graph_step import GraphedStep
            g = GraphedStep(trainer, batches[0])
            assert g.captured

            def run():
                nonlocal i
                i += 1
                g(batches[i % len(batches)])
        import time
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"shipped_step B={B} {mode} {amp or 'float32'}: {1e3 * el / steps:.3f} ms/step ({B * steps / el:.1f} pairs/s), host enqueue {1e3 * host / steps:.3f} ms/step, "
              f"DL_WINO_SPLIT={os.environ.get('DL_WINO_SPLIT', '1')}")
    finally:
        import shutil
        shutil.rmtree(tree["path"], ignore_errors=True)


if __name__ == "__main__":
    main()
