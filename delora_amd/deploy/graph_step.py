"""Whole-step HIP graph: capture projection -> normals -> CNN -> correspondences -> loss -> backward -> Adam once, replay it for
every batch -- ragged batches included.

All launches of the step -- torch's and the C-ABI kernels, which are enqueued on torch's current stream and never
synchronise -- are capturable.  What a graph needs is static ADDRESSES and SHAPES, not static contents: the 2B scans of a batch
are packed into one static point buffer of fixed capacity and addressed through CSR offsets that live in a static device tensor
(``step_geometry.PackedBatch``); the projection launch is sized for the per-scan capacity and bounds every scan by its offsets, so
scans of any length up to the capacity replay through the same graph.  Requirements: one sensor / image size per batch, a fixed
batch size, an optimiser with ``capturable=True``, world_size == 1 (DDP's bucketed all-reduce is left eager), no augmentation
or range normalisation.  ``GraphedStep`` runs the eager step whenever one of them does not hold (or capture fails).

Why: a 64x2048, B=8 step is ~180 launches (fp32) / ~210 (autocast).  In fp32 the GPU is busy 98 % of the step and the host needs
half the step to enqueue it, so capture buys nothing there; in half precision the step is 5 ms of GPU time against 4.3 ms of
host enqueue, and any host jitter shows.
"""
import os

import torch

from . import step_geometry


class GraphedStep:
    def __init__(self, trainer, example_batch, warmup=3, max_points=None):
        """``example_batch``: a list of sample dicts on the device (shapes the capture is warmed up with).  ``max_points``: capacity
        per scan of the static buffer (default: the longest scan of the example + 15 %, rounded up to 4096)."""
        self.trainer = trainer
        self.graph = None
        self.outputs = None
        self.fallback_steps = 0
        d0 = example_batch[0]
        self.B = len(example_batch)
        self.dataset = d0["dataset"]
        self.with_lists = d0.get("normal_list_1") is not None
        longest = max(d[k].shape[2] for d in example_batch for k in ("scan_1", "scan_2"))
        self.capacity = int(max_points) if max_points else ((int(longest * 1.15) + 4095) // 4096) * 4096
        dev = d0["scan_1"].device
        C = 6 if self.with_lists else 3
        self.pts = torch.zeros((C, 2 * self.B * self.capacity), dtype=torch.float32, device=dev)
        self.offs = torch.zeros((2 * self.B + 1,), dtype=torch.int32, device=dev)
        self.packed = step_geometry.PackedBatch(self.pts, self.offs, self.capacity, self.B, self.dataset, self.with_lists)
        cfg = trainer.config
        # a PackedBatch cannot be augmented or rescaled (Deployer.step rejects it) and one capture serves one sensor and one rank:
        # when a requirement does not hold every call runs the eager step on the caller's own list of dicts
        self.eligible = not (trainer.world_size != 1 or cfg["normalization_scaling"] or cfg["random_point_cloud_rotations"]
                             or len({d["dataset"] for d in example_batch}) != 1 or self.B != trainer.batch_size
                             or getattr(trainer, "grad_scaler", None) is not None)          # float16 loss scaling: left eager
        if not self.eligible:
            return
        for group in trainer.optimizer.param_groups:
            group["capturable"] = True
        for p_, st in trainer.optimizer.state.items():           # steps taken so far were counted on the host
            if torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                st["step"] = st["step"].to(p_.device)
        try:
            self.pack(example_batch)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._step(self.packed)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if os.environ.get("DL_GRAPH_DEBUG"):
                print("[graph_step] warm-up on the side stream finished; capturing", flush=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.outputs = self._step(self.packed)
            self.graph = g
        except Exception as e:                      # noqa: BLE001 -- any capture problem: stay eager, say so
            print(f"[delora_amd] HIP graph capture of the step failed ({type(e).__name__}: {e}); running eagerly")
            self.graph = None
            torch.cuda.synchronize()

    def _step(self, batch):
        tr = self.trainer
        tr.optimizer.zero_grad(set_to_none=True)
        ep = tr.new_epoch_losses()
        ep, T = tr.step(preprocessed_dicts=batch if isinstance(batch, step_geometry.PackedBatch) else [dict(d) for d in batch],
                        epoch_losses=ep)
        return ep, T

    @property
    def captured(self):
        return self.graph is not None

    def fits(self, batch):
        """Whether ``batch`` can go through the captured graph: same batch size, sensor and list kind, every scan within capacity."""
        if len(batch) != self.B:
            return False
        for d in batch:
            if d["dataset"] != self.dataset or (d.get("normal_list_1") is not None) != self.with_lists:
                return False
            if d["scan_1"].shape[2] > self.capacity or d["scan_2"].shape[2] > self.capacity:
                return False
        return True

    def pack(self, batch):
        """Copy the scans of ``batch`` (device or pinned host tensors) into the static point buffer and write the CSR offsets."""
        offs, o = [0], 0
        for d in batch:
            for k in ("1", "2"):
                scan = d["scan_" + k][0]
                n = scan.shape[1]
                self.pts[:3, o:o + n].copy_(scan[:3], non_blocking=True)
                if self.with_lists:
                    self.pts[3:6, o:o + n].copy_(d["normal_list_" + k][0], non_blocking=True)
                o += n
                offs.append(o)
        self.offs.copy_(torch.tensor(offs, dtype=torch.int32), non_blocking=False)

    def __call__(self, batch=None):
        """One training step on ``batch`` (None: the contents already in the static buffers).  Returns (epoch_losses, T) -- with a
        captured graph these are the graph's static output tensors, overwritten by the next call."""
        if batch is not None and (not self.eligible or not self.fits(batch)):
            self.fallback_steps += 1
            return self._step(batch)
        if not self.eligible:
            raise ValueError("this configuration cannot run on the packed static buffers (augmentation, range normalisation, mixed "
                             "sensors or several ranks): pass the batch itself")
        if batch is not None:
            self.pack(batch)
        if self.graph is None:
            return self._step(self.packed)
        self.graph.replay()
        return self.outputs
