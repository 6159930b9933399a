"""Host -> device feed at rate: a one-batch-ahead prefetcher.

The reference moves every tensor of a batch to the device synchronously inside the training loop
(src/deploy/trainer.py:62-66).  Here the next batch is pinned and copied on a side stream while the current step runs;
at 64x2048 a pair is 2 x ~141k points x 12 B = 3.4 MB (6.8 MB with stored normal lists), i.e. ~1-2 GB/s at 280 pairs/s
per GPU against 63 GB/s of PCIe Gen5 -- the copy disappears behind the step.
"""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device):
        self.loader, self.device = loader, device
        self.cuda = getattr(device, "type", str(device)) == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        if not self.cuda:
            for d in batch:
                for k, v in d.items():
                    if hasattr(v, "to"):
                        d[k] = v.to(self.device)
            return batch
        keep = []
        with torch.cuda.stream(self.stream):
            for d in batch:
                for k, v in d.items():
                    if torch.is_tensor(v):
                        if not v.is_cuda and not v.is_pinned():
                            v = v.pin_memory()
                        if not v.is_cuda:
                            keep.append(v)
                        d[k] = v.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        # the page-locked sources stay referenced until their copies have run (the same care as PackedFeed's per-slot offsets vector:
        # a source handed back to the host allocator at once may be rewritten before an asynchronous copy has read it)
        self._pending = [(e, k_) for e, k_ in getattr(self, "_pending", []) if not e.query()]
        self._pending.append((ev, keep))
        return batch

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for d in nxt:                       # tensors were allocated on the side stream: tell the allocator
                    for v in d.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(torch.cuda.current_stream(self.device))
            cur = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            yield cur


# ---------------------------------------------------------------------------------------------------------------------------------
# PackedFeed: worker processes decode whole batches straight into page-locked shared memory, in the layout the step's first kernel
# reads.
#
# What a step consumes is ONE planar point buffer ``pts [C, sumN]`` (the 2B scans of the batch back to back, C = 3, or 6 with the
# stored normal lists as extra rows) plus CSR offsets (deploy/step_geometry.PackedBatch).  torch's DataLoader hands a batch over as
# 16-32 separate tensors through per-tensor shared-memory files and re-pins every one of them (hipHostMalloc per tensor and batch:
# measured 23 ms per batch of 8 pairs, 0.63x of the resident rate at fp32 and 0.32x at bf16).  Here:
#   * a ring of batch SLOTS is allocated once in shared memory and page-locked once (hipHostRegister) -- no allocation, no pinning
#     and no pickling of tensor data afterwards;
#   * worker processes (forked after the slots exist) np.load the ``[M,3]`` files of a batch and write them TRANSPOSED straight into
#     their slot (one pass: decode -> planar slot), a scan shared by consecutive pairs once;
#   * the consumer starts ONE host-to-device copy per batch on a side stream, one batch ahead of the step, and recycles the slot when
#     the copy has finished.
# The order of the samples is torch's own (RandomSampler / DistributedSampler(drop_last) + BatchSampler), so an epoch visits what
# Trainer.make_dataloader's DataLoader would visit.  Reference: src/data/dataset.py:82-154, src/deploy/trainer.py:95-101.
import multiprocessing as _mp
import queue as _queue


def _fill_slot(dataset, out, indices, rows, capacity):
    """The scans of the samples ``indices`` into ``out [rows][capacity]`` (planar fp32: row c holds coordinate c of the batch's scans back
    to back; rows 3..5 the stored normals).  Returns (scan lengths, (seconds reading, seconds copying))."""
    import time as _t
    lengths, o, t_read, t_copy = [], 0, 0.0, 0.0
    planar = bool(getattr(dataset, "store_dataset_in_RAM", False))   # RAM: the [3,M] rows of the stored tensors; disk: [M,3] files
    for i in indices:
        t0 = _t.perf_counter()
        pair = dataset.load_pair_arrays(int(i))                      # [(xyz, normals | None)] x 2: [M,3] from disk, [3,M] from RAM
        t1 = _t.perf_counter()
        for xyz, nrm in pair:
            n = xyz.shape[1] if planar else xyz.shape[0]
            if o + n > capacity:
                raise ValueError(f"batch exceeds the slot capacity of {capacity} points (config feed_points_per_scan)")
            out[:3, o:o + n] = xyz[:3] if planar else xyz[:, :3].T       # decode -> planar slot in one pass
            if rows == 6:
                out[3:6, o:o + n] = nrm if planar else nrm.T
            lengths.append(n)
            o += n
        t_read += t1 - t0
        t_copy += _t.perf_counter() - t1
    return lengths, (t_read, t_copy)


def _feed_worker(dataset, slots, tasks, done, rows, capacity):
    """Worker loop: (batch id, sample indices, slot) -> the slot filled by ``_fill_slot`` + the scan lengths."""
    torch.set_num_threads(1)
    while True:
        task = tasks.get()
        if task is None:
            return
        bid, indices, slot = task
        try:
            lengths, spent = _fill_slot(dataset, slots[slot].numpy().reshape(rows, capacity), indices, rows, capacity)
            done.put((bid, slot, lengths, spent, None))
        except Exception as e:                                       # noqa: BLE001 -- reported to the consumer, which raises
            done.put((bid, slot, None, None, f"{type(e).__name__}: {e}"))


class PackedFeed:
    """Iterable over the batches of one epoch as ``PackedBatch`` objects on ``device`` (see the block comment above).

    ``dataset``: a ``PreprocessedPointCloudDataset`` over ONE dataset block (one sensor), on disk or held in RAM; ``batch_sampler``: an
    iterable of index lists per epoch (torch's BatchSampler); ``workers`` processes -- **0 = no processes at all**: the consumer
    itself decodes the next batch into a page-locked slot (the reference's ``num_dataloader_workers: 0`` taken literally; what it saves
    over the DataLoader is the per-tensor pinning and the 4-32 separate copies per batch: measured 44 ms -> see bench.py
    ``shipped_config``); ``points_per_scan``: upper bound of a stored list's length (the slot capacity is 2 * batch_size * that)."""

    def __init__(self, dataset, batch_sampler, batch_size, device, workers=4, points_per_scan=None, slots=None, ahead=None):
        from ..deploy.step_geometry import PackedBatch
        self._PackedBatch = PackedBatch
        self.dataset, self.batch_sampler, self.B, self.device = dataset, batch_sampler, int(batch_size), device
        self.cuda = getattr(device, "type", str(device)) == "cuda"
        self.rows = 6 if dataset.load_normals else 3
        self.dataset_name = dataset.config["datasets"][0]
        self.workers = max(0, int(workers))
        self.ahead = int(ahead) if ahead else self.workers + 2            # batches in flight
        n_slots = int(slots) if slots else self.ahead + 2
        cap = int(points_per_scan or dataset.max_points_per_scan())
        self.capacity = 2 * self.B * cap
        # the ring of slots: shared between the processes (allocated before the fork), page-locked for asynchronous copies
        self.slots = [torch.empty((self.rows * self.capacity,), dtype=torch.float32).share_memory_() for _ in range(n_slots)]
        self.pinned = False
        if self.cuda:
            rt = torch.cuda.cudart()
            self.pinned = all(int(rt.cudaHostRegister(s.data_ptr(), s.numel() * 4, 0)) == 0 for s in self.slots)
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.slot_offs = [torch.zeros((2 * self.B + 1,), dtype=torch.int32).pin_memory() for _ in range(n_slots)] if self.cuda else []
        ctx = _mp.get_context("fork")
        self.tasks, self.done = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_feed_worker, args=(dataset, self.slots, self.tasks, self.done, self.rows, self.capacity), daemon=True)
                      for _ in range(self.workers)]
        for p in self.procs:
            p.start()
        self.bytes_moved = 0
        self.host_seconds = {"wait_for_workers": 0.0, "upload_enqueue": 0.0, "reclaim": 0.0}      # where the consumer's time goes
        self._outstanding = 0                       # tasks handed to the workers whose result has not been received yet
        self._uploads = []                          # (event, slot): copies of an abandoned epoch still in flight
        self._generation = 0                        # one consumer at a time: starting an epoch invalidates the iterators before it

    def __len__(self):
        return len(self.batch_sampler)

    def close(self):
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        if self.cuda and self.pinned and self.slots:
            for ev, _ in self._uploads:                 # a copy still in flight reads the slot it is about to lose
                if ev is not None:
                    ev.synchronize()
            rt = torch.cuda.cudart()
            for s in self.slots:
                rt.cudaHostUnregister(s.data_ptr())
        self.procs, self.slots, self._uploads = [], [], []

    def __del__(self):
        try:
            if self.procs or self.slots:
                self.close()
        except Exception:                                            # noqa: BLE001 -- interpreter shutdown
            pass

    def _upload(self, slot, lengths):
        """Slot -> device (side stream); returns (PackedBatch, event after which the slot may be refilled)."""
        total = int(sum(lengths))
        host = self.slots[slot].view(self.rows, self.capacity)[:, :total]          # row stride = the slot capacity
        offs = [0]
        for n in lengths:
            offs.append(offs[-1] + n)
        if not self.cuda:
            return self._PackedBatch(host.clone(), torch.tensor(offs, dtype=torch.int32), max(lengths), self.B, self.dataset_name,
                                     self.rows == 6), None
        with torch.cuda.stream(self.stream):
            pts = torch.empty((self.rows, total), dtype=torch.float32, device=self.device)
            for c in range(self.rows):                                   # one contiguous DMA per coordinate row
                pts[c].copy_(host[c], non_blocking=True)
            # the CSR offsets go through a page-locked vector that belongs to the SLOT: it is rewritten only when the slot is, i.e. after
            # this upload's event.  (A temporary `torch.tensor(offs).pin_memory()` is handed back to torch's host allocator at once;
            # with batches arriving back to back from worker processes the next batch's offsets could land in the same block before
            # this copy had run -- round 6: one step in a few hundred trained on its neighbour's scan boundaries.)
            stage = self.slot_offs[slot]
            stage[:len(offs)] = torch.tensor(offs, dtype=torch.int32)
            offs_t = stage[:len(offs)].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.bytes_moved += pts.numel() * 4
        return self._PackedBatch(pts, offs_t, max(lengths), self.B, self.dataset_name, self.rows == 6), ev

    def _drain(self):
        """An epoch that was abandoned half-way (an exception in the training loop, a `break`) leaves tasks with the workers and copies
        in flight: collect them, so that the next epoch starts with every slot free and an empty result queue."""
        while self._outstanding > 0:
            try:
                self.done.get(timeout=120)
            except _queue.Empty:
                raise RuntimeError("PackedFeed: the worker processes do not answer")
            self._outstanding -= 1
        for ev, _ in self._uploads:
            if ev is not None:
                ev.synchronize()
        self._uploads = []

    def __iter__(self):
        """One epoch.  SINGLE consumer: the worker queues, the slots and the uploads in flight belong to the feed, not to the iterator, so
        starting a new epoch while an earlier iterator is still suspended takes them over -- the earlier iterator raises on its next
        ``next()`` instead of waiting for batches that were drained (advisor, round 4)."""
        self._generation += 1
        generation = self._generation
        self._drain()
        batches = iter(self.batch_sampler)
        free = list(range(len(self.slots)))
        busy = self._uploads                        # (event, slot): uploads in flight
        arrived = {}                                # batch id -> (slot, lengths)
        issued = consumed = 0
        exhausted = False
        staged = None                               # the uploaded batch waiting to be handed out (one ahead of the step)

        import time as _time

        def reclaim(block):
            t0 = _time.perf_counter()
            _reclaim(block)
            self.host_seconds["reclaim"] += _time.perf_counter() - t0

        def _reclaim(block):
            while busy and (block or busy[0][0] is None or busy[0][0].query()):
                ev, slot = busy.pop(0)
                if ev is not None:
                    ev.synchronize()
                free.append(slot)
                block = False

        def issue():
            nonlocal issued, exhausted
            while not exhausted and free and issued - consumed < (self.ahead if self.workers else 1):
                try:
                    idx = next(batches)
                except StopIteration:
                    exhausted = True
                    return
                slot = free.pop(0)
                if self.workers:
                    self.tasks.put((issued, [int(i) for i in idx], slot))
                    self._outstanding += 1
                else:                                                    # no worker processes: decode here, now
                    try:
                        lengths, spent = _fill_slot(self.dataset, self.slots[slot].numpy().reshape(self.rows, self.capacity),
                                                    [int(i) for i in idx], self.rows, self.capacity)
                    except Exception as e:                               # noqa: BLE001 -- same error type as a worker's report
                        raise RuntimeError(f"PackedFeed worker: {type(e).__name__}: {e}")
                    self.host_seconds["worker_read_files"] = self.host_seconds.get("worker_read_files", 0.0) + spent[0]
                    self.host_seconds["worker_transpose_into_slot"] = self.host_seconds.get("worker_transpose_into_slot", 0.0) + spent[1]
                    arrived[issued] = (slot, lengths)
                issued += 1

        def next_uploaded():
            nonlocal consumed
            if generation != self._generation:
                raise RuntimeError("PackedFeed: this iterator was abandoned -- a later iter() of the same feed took over its workers "
                                   "and slots (one consumer at a time)")
            if consumed >= issued:
                return None
            t0 = _time.perf_counter()
            while consumed not in arrived:
                try:
                    bid, slot, lengths, spent, err = self.done.get(timeout=120)
                except _queue.Empty:
                    raise RuntimeError("PackedFeed: no batch from the worker processes for 120 s")
                self._outstanding -= 1
                if err is not None:
                    raise RuntimeError("PackedFeed worker: " + err)
                self.host_seconds["worker_read_files"] = self.host_seconds.get("worker_read_files", 0.0) + spent[0]
                self.host_seconds["worker_transpose_into_slot"] = self.host_seconds.get("worker_transpose_into_slot", 0.0) + spent[1]
                arrived[bid] = (slot, lengths)
            slot, lengths = arrived.pop(consumed)
            consumed += 1
            t1 = _time.perf_counter()
            packed, ev = self._upload(slot, lengths)
            busy.append((ev, slot))
            self.host_seconds["wait_for_workers"] += t1 - t0
            self.host_seconds["upload_enqueue"] += _time.perf_counter() - t1
            return packed

        issue()
        staged = next_uploaded()
        while staged is not None:
            reclaim(block=not free and not exhausted)
            issue()
            cur = staged
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                cur.pts.record_stream(torch.cuda.current_stream(self.device))
                cur.offs.record_stream(torch.cuda.current_stream(self.device))
            staged = next_uploaded()                # the next batch's copy runs while the caller's step is enqueued
            yield cur
        reclaim(block=True)


def packed_feed_applicable(dataset, config, device):
    """Whether the training set can go through ``PackedFeed``: a CUDA device, the reference's dataset (on disk or in RAM) over ONE
    dataset block (one sensor per batch), no per-sample host preprocessing (augmentation / range normalisation work on the sample
    dicts).  Any worker count, 0 included (the consumer then decodes in-process): the DataLoader path re-pins every tensor of every
    batch and costs the host tens of milliseconds per step -- it remains for what the packed layout cannot express."""
    from .dataset import PreprocessedPointCloudDataset
    return (isinstance(dataset, PreprocessedPointCloudDataset) and len(config["datasets"]) == 1
            and int(config.get("num_dataloader_workers", 0)) >= 0 and bool(config.get("packed_feed", True))
            and not config.get("normalization_scaling") and not config.get("random_point_cloud_rotations")
            and getattr(device, "type", str(device)) == "cuda")


def make_packed_feed(dataset, config, device, batch_size, sampler=None, shuffle=True):
    """``PackedFeed`` over torch's own samplers: RandomSampler (or the given DistributedSampler) + BatchSampler(drop_last)."""
    if sampler is None:
        sampler = torch.utils.data.RandomSampler(dataset) if shuffle else torch.utils.data.SequentialSampler(dataset)
    batches = torch.utils.data.BatchSampler(sampler, batch_size=batch_size, drop_last=True)
    return PackedFeed(dataset, batches, batch_size, device, workers=int(config["num_dataloader_workers"]),
                      points_per_scan=config.get("feed_points_per_scan"), ahead=config.get("feed_batches_ahead"))
