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
    def __init__(self, trainer, example_batch, warmup=3, max_points=None, restore_after_warmup=True):
        """``example_batch``: a list of sample dicts on the device, or a ``PackedBatch`` of the packed feed (shapes the capture is
        warmed up with).  ``max_points``: capacity per scan of the static buffer (default: the longest scan of the example + 15 %,
        rounded up to 4096; the trainer passes the feed's own slot capacity when the batches come from ``PackedFeed``).
        ``restore_after_warmup``: the warm-up steps a capture needs are REAL optimisation steps on the example batch; with this flag
        (default) weights, Adam moments and step counters are put back afterwards, so that building a ``GraphedStep`` leaves the
        training trajectory untouched -- the run replays the example batch as its next step and continues exactly where an eager run
        would be (``Trainer`` switches to replay in the middle of an epoch)."""
        self.trainer = trainer
        self.graph = None
        self.outputs = None
        self.fallback_steps = 0
        self.replayed_steps = 0
        self.acc = self.acc_keys = None             # static accumulator of the step's metric vector, added to by every replay
        packed_example = isinstance(example_batch, step_geometry.PackedBatch)
        self.B = len(example_batch)
        if packed_example:
            self.dataset, self.with_lists = example_batch.dataset, example_batch.with_lists
            longest, dev = example_batch.max_points, example_batch.pts.device
            datasets = {self.dataset}
        else:
            d0 = example_batch[0]
            self.dataset = d0["dataset"]
            self.with_lists = d0.get("normal_list_1") is not None
            longest = max(d[k].shape[2] for d in example_batch for k in ("scan_1", "scan_2"))
            dev = d0["scan_1"].device
            datasets = {d["dataset"] for d in example_batch}
        self.capacity = int(max_points) if max_points else ((int(longest * 1.15) + 4095) // 4096) * 4096
        C = 6 if self.with_lists else 3
        self.pts = torch.zeros((C, 2 * self.B * self.capacity), dtype=torch.float32, device=dev)
        self.offs = torch.zeros((2 * self.B + 1,), dtype=torch.int32, device=dev)
        self.packed = step_geometry.PackedBatch(self.pts, self.offs, self.capacity, self.B, self.dataset, self.with_lists)
        # CSR offsets of a list batch reach the device through a small ring of page-locked staging vectors (an asynchronous copy: a
        # pageable source would make every step wait for the stream, and the host could never run ahead of the GPU)
        self._offs_ring = ([torch.zeros((2 * self.B + 1,), dtype=torch.int32).pin_memory() for _ in range(4)]
                           if dev.type == "cuda" and torch.cuda.is_available() else None)
        self._offs_events = [None] * 4
        self._offs_next = 0
        # a PackedBatch cannot be augmented or rescaled (Deployer.step rejects it) and one capture serves one sensor and one rank:
        # when a requirement does not hold every call runs the eager step on the caller's own list of dicts
        self.eligible = GraphedStep.config_eligible(trainer) and len(datasets) == 1 and self.B == trainer.batch_size
        if not self.eligible:
            return
        for group in trainer.optimizer.param_groups:
            group["capturable"] = True
        for p_, st in trainer.optimizer.state.items():           # steps taken so far were counted on the host
            if torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                st["step"] = st["step"].to(p_.device)
        snapshot = None
        try:
            self.pack(example_batch)
            snapshot = self._snapshot() if restore_after_warmup else None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ep_w = {}
                for _ in range(warmup):
                    ep_w, _ = self._step(self.packed)
                # the step's metrics are views of ONE stacked vector (Deployer._accumulate): the captured graph adds that vector to
                # a static accumulator of its own, so that a training loop pays nothing per step for its epoch sums
                self.acc_keys, base = self._metric_vector(ep_w)
                self.acc = torch.zeros_like(base) if base is not None else None
                del ep_w
            torch.cuda.current_stream().wait_stream(side)
            if snapshot is not None:
                self._restore(snapshot)
                snapshot = None
            torch.cuda.synchronize()
            if os.environ.get("DL_GRAPH_DEBUG"):
                print("[graph_step] warm-up on the side stream finished; capturing", flush=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.outputs = self._step(self.packed)
                if self.acc is not None:
                    keys, base = self._metric_vector(self.outputs[0])
                    if keys == self.acc_keys:
                        self.acc.add_(base)
                    else:
                        self.acc = self.acc_keys = None
            self.graph = g
        except Exception as e:                      # noqa: BLE001 -- any capture problem: stay eager, say so
            print(f"[delora_amd] HIP graph capture of the step failed ({type(e).__name__}: {e}); running eagerly")
            self.graph = None
            torch.cuda.synchronize()
            # (advisor, round 5) a failure in the warm-up or the capture set-up must not leave the warm-up's real steps behind: weights
            # and Adam moments go back to where the caller had them, and the optimiser back to its eager form
            if snapshot is not None:
                self._restore(snapshot)
                torch.cuda.synchronize()
            self.release_optimizer()

    def release_optimizer(self):
        """Undo what a capture asked of the optimiser (`capturable`, step counters on the device): the eager step and the checkpoints
        it writes are then those of a trainer that never built a graph.  Called when a capture fails or is dropped again."""
        opt = self.trainer.optimizer
        fused = any(bool(group.get("fused")) for group in opt.param_groups)
        for group in opt.param_groups:
            group["capturable"] = False
        if fused:
            return                                   # torch's single-kernel Adam keeps its step counters on the device in every mode
        for st in opt.state.values():
            if torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                st["step"] = st["step"].detach().to("cpu")

    @staticmethod
    def _metric_vector(ep):
        """(keys in element order, the stacked vector) when every tensor-valued entry of ``ep`` is an element of one 1-d tensor."""
        items = [(k, v) for k, v in ep.items() if torch.is_tensor(v)]
        if not items or any(v._base is None or v._base is not items[0][1]._base or v.dim() != 0 for _, v in items):
            return None, None
        base = items[0][1]._base
        if base.dim() != 1 or len(items) != base.numel():
            return None, None
        items.sort(key=lambda kv: kv[1].storage_offset())
        return tuple(k for k, _ in items), base

    def take_epoch_sums(self):
        """{metric: 0-d tensor} accumulated by the replays since the last call (a copy; the accumulator restarts at zero)."""
        if self.acc is None:
            return {}
        snap = self.acc.clone()
        self.acc.zero_()
        return {k: snap[i] for i, k in enumerate(self.acc_keys)}

    def _snapshot(self):
        """Copies of everything a training step changes: the parameters and the optimiser's per-parameter state (None for a parameter
        the optimiser has not stepped yet: its state is created by the warm-up and reset to "never stepped" values afterwards)."""
        opt = self.trainer.optimizer
        params = [p for g in opt.param_groups for p in g["params"]]
        return [(p, p.detach().clone(), {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()}
                 if p in opt.state and opt.state[p] else None) for p in params]

    def _restore(self, snapshot):
        opt = self.trainer.optimizer
        with torch.no_grad():
            for p, value, state in snapshot:
                p.copy_(value)
                now = opt.state.get(p)
                if not now:
                    continue
                for k, v in now.items():
                    if torch.is_tensor(v):
                        if state is not None and torch.is_tensor(state.get(k)):
                            v.copy_(state[k])
                        else:
                            v.zero_()                             # exp_avg, exp_avg_sq, step of a parameter that had never been stepped
                    elif state is not None and k in state:
                        now[k] = state[k]
        self.trainer.optimizer.zero_grad(set_to_none=True)

    @staticmethod
    def config_eligible(trainer):
        """Whether this trainer's configuration can run as a captured step at all (the batch-dependent requirements -- one sensor,
        the configured batch size -- are checked per batch): one rank (DDP's bucketed all-reduce is left eager), no augmentation or
        range normalisation (they work on the sample dicts), no float16 loss scaling (its skipped steps are host decisions)."""
        cfg = trainer.config
        return not (trainer.world_size != 1 or cfg["normalization_scaling"] or cfg["random_point_cloud_rotations"]
                    or getattr(trainer, "grad_scaler", None) is not None or cfg.get("use_jit"))

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
        if isinstance(batch, step_geometry.PackedBatch):
            return (batch.dataset == self.dataset and batch.with_lists == self.with_lists and batch.max_points <= self.capacity
                    and batch.pts.shape[0] == self.pts.shape[0] and batch.pts.shape[1] <= self.pts.shape[1])
        for d in batch:
            if d["dataset"] != self.dataset or (d.get("normal_list_1") is not None) != self.with_lists:
                return False
            if d["scan_1"].shape[2] > self.capacity or d["scan_2"].shape[2] > self.capacity:
                return False
        return True

    def pack(self, batch):
        """Copy the scans of ``batch`` into the static point buffer and write the CSR offsets.  A ``PackedBatch`` of the feed (already
        concatenated on the device, one batch ahead of the step) takes ONE strided device copy + the 2B+1 offsets; a list of dicts
        (device or pinned host tensors) one copy per scan and the offsets through the page-locked ring."""
        if isinstance(batch, step_geometry.PackedBatch):
            if batch.pts is not self.pts:
                self.pts[:, :batch.pts.shape[1]].copy_(batch.pts, non_blocking=True)
                self.offs.copy_(batch.offs, non_blocking=True)
            return
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
        if self._offs_ring is None:
            self.offs.copy_(torch.tensor(offs, dtype=torch.int32))
            return
        i = self._offs_next
        self._offs_next = (i + 1) % len(self._offs_ring)
        if self._offs_events[i] is not None:
            self._offs_events[i].synchronize()                  # four steps back: long finished unless the host runs far ahead
        self._offs_ring[i].copy_(torch.tensor(offs, dtype=torch.int32))
        self.offs.copy_(self._offs_ring[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._offs_events[i] = ev

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
        self.replayed_steps += 1
        return self.outputs
