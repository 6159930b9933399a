"""Whole-step HIP graph: capture projection -> normals -> CNN -> correspondences -> loss -> backward -> Adam once, replay it.

A training step at 64x2048, B=8 is ~450 kernel launches in ~28 ms; ~2.4 ms of that is launch gaps (profiles/).  All
launches of the step -- torch's and the C-ABI kernels, which are enqueued on torch's current stream and never
synchronise -- are capturable, so the static-shape step can be replayed as one graph.  Requirements: fixed tensor
shapes per batch (scan lengths: pad with far-away points, the projection drops them), an optimiser created with
``capturable=True``, and world_size == 1 (DDP's bucketed all-reduce is left eager).  ``GraphedStep`` falls back to the
eager step if capture fails.
"""
import os

import torch


class GraphedStep:
    def __init__(self, trainer, example_batch, warmup=3):
        self.trainer = trainer
        self.static_batch = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()} for d in example_batch]
        self.graph = None
        self.outputs = None
        if trainer.world_size != 1:
            return
        for group in trainer.optimizer.param_groups:
            group["capturable"] = True
        for p_, st in trainer.optimizer.state.items():           # steps taken so far were counted on the host
            if torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                st["step"] = st["step"].to(p_.device)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager(self.static_batch)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if os.environ.get("DL_GRAPH_DEBUG"):
                print("[graph_step] warm-up on the side stream finished; capturing", flush=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.outputs = self._eager(self.static_batch)
            self.graph = g
        except Exception as e:                      # noqa: BLE001 -- any capture problem: stay eager, say so
            print(f"[delora_amd] HIP graph capture of the step failed ({type(e).__name__}: {e}); running eagerly")
            self.graph = None
            torch.cuda.synchronize()

    def _eager(self, batch):
        tr = self.trainer
        tr.optimizer.zero_grad(set_to_none=True)
        ep = tr.new_epoch_losses()
        ep, T = tr.step(preprocessed_dicts=[dict(d) for d in batch], epoch_losses=ep)
        return ep, T

    @property
    def captured(self):
        return self.graph is not None

    def __call__(self, batch=None):
        """One training step.  With ``batch`` the tensors are copied into the static input buffers first (shapes must match)."""
        if batch is not None:
            for dst, src in zip(self.static_batch, batch):
                for k, v in src.items():
                    if torch.is_tensor(v):
                        dst[k].copy_(v, non_blocking=True)
        if self.graph is None:
            return self._eager(self.static_batch)
        self.graph.replay()
        return self.outputs
