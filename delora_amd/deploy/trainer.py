"""``Trainer``: optimiser, epochs, checkpoints, logging.  Mirror of src/deploy/trainer.py:18-186.

Differences that the MI355X design needs and the reference does not have:
  * data parallelism: when torch.distributed is initialised (``torchrun``; backend "nccl" = RCCL over xGMI) the
    model is wrapped in DistributedDataParallel and the sampler shards the dataset; the loss weights follow the
    sample's position in the GLOBAL batch so that N ranks x B samples reproduce one process with N*B samples;
  * metrics stay on the device and are read back once per epoch (the reference syncs five scalars per step);
  * mlflow / qqdm are optional: if they are not installed, metrics are printed and checkpoints still written.
Checkpoint files keep the reference's layout (trainer.py:155-161): epoch, model_state_dict, optimizer_state_dict,
loss, parameters.
"""
import os

import numpy as np
import torch

from ..data import feed
from . import deployer

try:                                  # optional third-party logging, absent in the build/bench image
    import mlflow
    import mlflow.pytorch
except ImportError:                   # pragma: no cover
    mlflow = None
try:
    import qqdm
except ImportError:                   # pragma: no cover
    qqdm = None


class Trainer(deployer.Deployer):

    def __init__(self, config, dataset=None, geometry_backend=None):
        super().__init__(config=config, dataset=dataset, geometry_backend=geometry_backend)
        self.training_bool = True
        self.raw_model = self.model
        if self.world_size > 1:
            self.model = self._wrap_ddp(self.raw_model)
        # torch.optim.Adam as the reference (src/deploy/trainer.py:23); on the GPU its single-kernel implementation: the
        # default multi-tensor one takes a per-tensor slow path for the channels-last trunk weights (65 launches, 0.4 ms)
        fused = torch.device(self.device).type == "cuda" and str(config.get("adam_impl", "fused")) == "fused"
        self.optimizer = torch.optim.Adam(params=self.raw_model.parameters(), lr=config["learning_rate"], **({"fused": True} if fused else {}))
        if config["checkpoint"]:
            checkpoint = torch.load(config["checkpoint"], map_location=self.device, weights_only=False)
            self.raw_model.load_state_dict(checkpoint["model_state_dict"])
            print("Model weights loaded from " + config["checkpoint"])
            self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
            print("Optimizer parameters loaded from " + config["checkpoint"])
            if self.grad_scaler is not None and checkpoint.get("grad_scaler_state_dict"):
                self.grad_scaler.load_state_dict(checkpoint["grad_scaler_state_dict"])     # float16: resume at the scale the run had reached
            config["unsupervised_at_start"] = True      # a pretrained model continues unsupervised (trainer.py:35-36)
        if config["inference_only"]:
            print("Config error: Inference only does not make sense during training. Changing to inference_only=False.")
            config["inference_only"] = False

    def _wrap_ddp(self, model):
        """DistributedDataParallel over the process group (backend "nccl" = RCCL over xGMI).  Buckets fill in backward order:
        heads, fc and the four 9.4 MB convolutions of layer4 -- 80 % of the 47.5 MB -- are ready in the first fraction of
        backward, layer3 next; layer2..conv1 (2.7 MB) only at its very end.  With 5 MB buckets the last, exposed all-reduce
        carries ~4 MB (a 25 MB cap leaves ~21 MB for it)."""
        ids = [self.device.index] if getattr(self.device, "type", "cpu") == "cuda" else None
        # the HIP trunk as three autograd Functions ([layer1+2] [layer3] [layer4]) instead of one: layer4's and layer3's weight
        # gradients (89 % of the bytes) reach the reducer while the rest of the backward still runs (tests/test_gpu_configs.py)
        from ..models import ring_conv
        ring_conv.TRUNK_SEGMENTS = self.config.get("trunk_segments", "layer")
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, gradient_as_bucket_view=True,
                                                         bucket_cap_mb=self.config.get("ddp_bucket_cap_mb", 5))

    @staticmethod
    def new_epoch_losses():
        return {"loss_epoch": 0.0, "loss_point_cloud_epoch": 0.0, "loss_field_of_view_epoch": 0.0,
                "loss_po2po_epoch": 0.0, "loss_po2pl_epoch": 0.0, "loss_pl2pl_epoch": 0.0,
                "visible_pixels_epoch": 0.0, "loss_yaw_pitch_roll_epoch": np.zeros(3), "loss_true_trafo_epoch": 0.0}

    def to_device(self, preprocessed_dicts):
        for d in preprocessed_dicts:
            for key, value in d.items():
                if hasattr(value, "to"):
                    d[key] = value.to(self.device, non_blocking=True)
        return preprocessed_dicts

    # ------------------------------------------------------------------------------------------ eager step or replayed graph
    def graph_policy(self):
        """config ``hip_graph``: ``true`` -> every eligible step is replayed as one captured HIP graph; ``false`` (the default, also when
        the key is absent -- the reference's YAML does not have it) -> eager; ``"auto"`` -> MEASURE: the first eager steps of a training
        phase are timed and the step is captured when the host needs as long per step as the GPU does (the reference's default operating
        point, ``batch_size: 1`` on 64x720: ~120 launches, 2.3 ms to enqueue for 2.3 ms of GPU work), then timed again and kept only if
        the replay is faster.  Round 6 (advisor): `auto` is opt-in -- its decision depends on wall-clock measurements, a captured Adam step
        differs from the eager one by an ulp, and a training run's trajectory must not depend on how busy the host was; what the default
        gives up is the 2 % the replay bought at batch 1 (2.31 -> 2.26 ms)."""
        from .graph_step import GraphedStep
        if getattr(self.device, "type", "cpu") != "cuda" or not GraphedStep.config_eligible(self):
            return "off"
        v = self.config.get("hip_graph", False)
        if isinstance(v, str):
            v = v.strip().lower()
        if v in (True, 1, "true", "on", "yes"):
            return "on"
        if v in (False, 0, None, "false", "off", "no"):
            return "off"
        return "auto"

    PROBE_SKIP, PROBE_STEPS, PROBE_HOST_BOUND = 8, 8, 0.8

    def _probe_step(self, phase, run_eager):
        """One eager step of the ``auto`` policy's measurement.  Returns the step's result.  After PROBE_SKIP untimed steps (first-call
        set-up, the caching allocator meeting the sizes of ragged batches), PROBE_STEPS steps are bracketed by a pair of events on the
        stream; the host clock measures the loop's PERIOD (step start to step start: the loader's time included) and the enqueue time
        of the step alone.  When the host is the bottleneck the GPU finishes each step right behind its last launch and idles until
        the next one arrives (period >= stream time); when the GPU is, the queue fills up, the events measure pure GPU time and the
        host's period is shorter."""
        import time
        st = self._graph_probe.setdefault(phase, {"n": 0, "host": 0.0, "events": [], "prev": None, "period_sum": 0.0, "periods": 0})
        timed = st["n"] >= self.PROBE_SKIP
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            t0 = time.perf_counter()
            if st["prev"] is not None:                  # (None at the first timed step and at the first step of an epoch: the gap between
                st["period_sum"] += t0 - st["prev"]     #  two epochs -- checkpoint, logging -- is not a step's period)
                st["periods"] += 1
            st["prev"] = t0
        out = run_eager()
        if timed:
            st["host"] += time.perf_counter() - t0
            b.record()
            st["events"].append((a, b))
        st["n"] += 1
        if len(st["events"]) >= self.PROBE_STEPS:
            st["events"][-1][1].synchronize()
            k = len(st["events"])
            gpu = sum(x.elapsed_time(y) for x, y in st["events"]) * 1e-3 / k
            enqueue = st["host"] / k
            period = st["period_sum"] / max(1, st["periods"])               # step start to step start, inside epochs only
            ratio = max(period, enqueue) / max(gpu, 1e-9)
            decision = "graph" if ratio >= float(self.config.get("hip_graph_auto_threshold", self.PROBE_HOST_BOUND)) else "eager"
            self._graph_decision[phase] = decision
            self.graph_probe_result[phase] = {"host_enqueue_ms": round(1e3 * enqueue, 3), "host_period_ms": round(1e3 * period, 3),
                                              "stream_ms": round(1e3 * gpu, 3), "ratio": round(ratio, 3), "decision": decision}
            if self.rank == 0:
                print(f"[delora_amd] hip_graph auto ({'unsupervised' if phase else 'identity'} phase): the host needs {1e3 * period:.2f} ms per "
                      f"step ({1e3 * enqueue:.2f} ms of it to enqueue the step), the stream {1e3 * gpu:.2f} ms -> "
                      + ("the step is host-bound: replaying it as ONE captured HIP graph" if decision == "graph"
                         else "the step is GPU-bound: staying eager"))
            del self._graph_probe[phase]
        return out

    def train_epoch(self, epoch, dataloader):
        epoch_losses = self.new_epoch_losses()
        # next batch is copied while this step runs (a PackedFeed does that itself and yields device-resident PackedBatch objects)
        iterator = dataloader if isinstance(dataloader, feed.PackedFeed) else feed.DevicePrefetcher(dataloader, self.device)
        show = self.rank == 0 and qqdm is not None
        if show:
            iterator = qqdm.qqdm(iterator, desc=qqdm.format_str("blue", "Epoch " + str(epoch)))
        every = max(1, int(self.config.get("progress_every", 50)))
        policy = self.graph_policy()
        if not hasattr(self, "_graph_decision"):
            self._graph_decision, self._graph_probe, self.graph_probe_result = {}, {}, {}
        for st in self._graph_probe.values():       # a measured period never spans an epoch boundary (checkpoint / logging time is not the step's)
            st["prev"] = None
        for tr_ in getattr(self, "_graph_trial", {}).values():
            tr_["prev"] = None
        self.graph_steps = getattr(self, "graph_steps", 0)
        if getattr(self, "_graphed", None) is not None:
            self._graphed.take_epoch_sums()             # (steps replayed outside an epoch, e.g. by a caller's own loop)
            if self._graphed_phase != bool(self.config["unsupervised_at_start"]):
                self._graphed = None                    # the training phase changed: the old capture (and its memory pool) goes now
        for counter, preprocessed_dicts in enumerate(iterator):
            phase = bool(self.config["unsupervised_at_start"])
            mode = "eager" if policy == "off" else ("graph" if policy == "on" else self._graph_decision.get(phase, "probe"))
            if mode == "graph":
                epoch_losses = self._graphed_step(preprocessed_dicts, epoch_losses)
                continue

            def run_eager(batch=preprocessed_dicts, ep=epoch_losses):
                self.optimizer.zero_grad(set_to_none=True)
                return self.step(preprocessed_dicts=batch, epoch_losses=ep, log_images_bool=False)[0]
            epoch_losses = self._probe_step(phase, run_eager) if mode == "probe" else run_eager()
            if show and counter % every == 0:          # one host sync per `every` steps instead of one per step
                iterator.set_infos({"loss": f'{float(epoch_losses["loss_epoch"]) / (counter + 1):.6f}',
                                    "loss_po2pl": f'{float(epoch_losses["loss_po2pl_epoch"]) / (counter + 1):.6f}',
                                    "loss_pl2pl": f'{float(epoch_losses["loss_pl2pl_epoch"]) / (counter + 1):.6f}'})
        return self._fold_graph_sums(epoch_losses)

    def _graphed_step(self, preprocessed_dicts, epoch_losses):
        """The step replayed as one captured HIP graph (deploy/graph_step.py; ragged batches go through static buffers of
        ``graph_max_points`` points per scan -- by default the packed feed's own slot capacity, else the example's longest scan + 15 %).
        Re-captured when the training phase changes.  A batch the capture cannot take (or a failed capture) runs eagerly."""
        from .graph_step import GraphedStep
        phase = bool(self.config["unsupervised_at_start"])
        g = getattr(self, "_graphed", None)
        if g is None or self._graphed_phase != phase:
            if g is not None:
                del self._graphed                       # the old capture's private memory pool goes before the new one is built
                g = None
            cap = self.config.get("graph_max_points") or getattr(self, "_feed_points_per_scan", None)
            if not cap and hasattr(self.dataset, "max_points_per_scan") and len(self.config["datasets"]) == 1:
                cap = self.dataset.max_points_per_scan()          # stored lists: at most one point per pixel of the preprocessing image
            g = self._graphed = GraphedStep(self, preprocessed_dicts, max_points=cap)
            self._graphed_phase = phase
            if self.graph_policy() == "auto" and g.captured:
                self.__dict__.setdefault("_graph_trial", {})[phase] = {"prev": None, "sum": 0.0, "n": 0}   # periods of the first replays (_judge_replay)
            if self.graph_policy() == "auto" and not g.captured:
                # a failed capture ends the experiment for this phase: plain eager steps from the caller's own batches, not eager steps
                # through the capture's full-capacity static buffers
                self._graph_decision[phase] = "eager"
                self.graph_probe_result.setdefault(phase, {})["decision"] = "eager (capture failed)"
                self._graphed = None
            if self.rank == 0:
                print(f"[delora_amd] training step captured as a HIP graph: {g.captured} ({'unsupervised' if phase else 'identity'} phase, "
                      f"{g.capacity} points per scan)")
        before = g.replayed_steps
        trial = self._graph_trial.get(phase) if hasattr(self, "_graph_trial") else None
        if trial is not None:
            import time
            now = time.perf_counter()
            if trial["prev"] is not None:
                trial["sum"] += now - trial["prev"]
                trial["n"] += 1
            trial["prev"] = now
        if not g.captured and getattr(self, "_graphed", None) is None:       # (auto: the capture just failed, see above)
            self.optimizer.zero_grad(set_to_none=True)
            return self.step(preprocessed_dicts=preprocessed_dicts, epoch_losses=epoch_losses, log_images_bool=False)[0]
        ep, _ = g(preprocessed_dicts)
        replayed = g.replayed_steps - before
        self.graph_steps += replayed
        if trial is not None and trial["n"] >= self.PROBE_STEPS:
            self._judge_replay(phase, trial["sum"] / trial["n"])
        if replayed and g.acc is not None:
            return epoch_losses                      # the replay added its metrics to the graph's own accumulator (folded in per epoch)
        for k, v in ep.items():                      # eager fallback / no accumulator: the outputs are (static) tensors, add their VALUES
            if torch.is_tensor(v):
                epoch_losses[k] = epoch_losses[k] + v
        return epoch_losses

    def _judge_replay(self, phase, period):
        """`auto` only: the host's period over the first replayed steps against the eager period the probe measured.  hipGraphLaunch
        of this ROCm release enqueues a captured step node by node -- measured on the reference's default batch-1 step: 0.9-1.6 ms to
        replay its ~120 kernel nodes against 2.2-2.6 ms to enqueue them eagerly, 2.2 against 2.3 ms on a loaded host -- so a replay
        that does not shorten the host's period by at least 3 % (``hip_graph_min_gain``) is dropped again; ``hip_graph_keep_if_slower``
        keeps it regardless."""
        del self._graph_trial[phase]
        res = self.graph_probe_result.get(phase, {})
        res["replay_host_period_ms"] = round(1e3 * period, 3)
        eager = res.get("host_period_ms")
        gain = float(self.config.get("hip_graph_min_gain", 0.03))
        keep = eager is None or 1e3 * period < (1.0 - gain) * max(eager, res.get("host_enqueue_ms", 0.0)) or bool(self.config.get("hip_graph_keep_if_slower", False))
        res["decision"] = "graph" if keep else "eager (replay measured, not faster)"
        if not keep:
            self._graph_decision[phase] = "eager"
            self._fold_pending = self._graphed.take_epoch_sums()      # metrics of the replayed steps: folded into the running epoch
            self._graphed.release_optimizer()                         # eager steps and checkpoints in the non-graph form again
            del self._graphed
            self._graphed = None
        if self.rank == 0:
            print(f"[delora_amd] hip_graph auto: replaying costs the host {1e3 * period:.2f} ms per step -> "
                  + ("keeping the captured graph" if keep else "no gain over the eager enqueue on this stack: back to the eager step"))

    def _fold_graph_sums(self, epoch_losses):
        pending = getattr(self, "_fold_pending", None)
        if pending:
            for k, v in pending.items():
                epoch_losses[k] = epoch_losses[k] + v
            self._fold_pending = None
        g = getattr(self, "_graphed", None)
        if g is not None:
            for k, v in g.take_epoch_sums().items():
                epoch_losses[k] = epoch_losses[k] + v
        return epoch_losses

    def _reduce_metrics(self, epoch_losses):
        keys = ("loss_epoch", "loss_point_cloud_epoch", "loss_po2po_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch",
                "visible_pixels_epoch")
        vec = torch.stack([torch.as_tensor(epoch_losses[k], dtype=torch.float32, device=self.device).reshape(()) for k in keys])
        if self.world_size > 1:
            # loss terms are already normalised by the global batch on every rank: the global value is the SUM.
            # visible_pixels is a COUNT of the last sample of the (global) batch (deployer.py:349-367): only the last
            # rank holds that sample, so the others contribute zero to the sum.
            if self.rank != self.world_size - 1:
                vec[-1] = 0.0
            torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.SUM)
        vals = (vec / max(self.steps_per_epoch_effective, 1)).tolist()
        return dict(zip(keys, vals))

    def save_checkpoint(self, path, epoch, loss):
        ck = {"epoch": epoch, "model_state_dict": self.raw_model.state_dict(),
              "optimizer_state_dict": self.optimizer.state_dict(), "loss": float(loss), "parameters": self.config}
        if self.grad_scaler is not None:            # an optional sixth key (float16 autocast only): the reference's five stay as they are
            ck["grad_scaler_state_dict"] = self.grad_scaler.state_dict()
        torch.save(ck, path)

    def make_dataloader(self):
        sampler = None
        if self.world_size > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.dataset, num_replicas=self.world_size,
                                                                      rank=self.rank, shuffle=True, drop_last=True)
        shuffle = bool(self.config.get("shuffle_training_data", True))          # the reference always shuffles (trainer.py:95-101)
        if feed.packed_feed_applicable(self.dataset, self.config, self.device):
            # the reference's on-disk training set with worker processes: batches are decoded straight into page-locked shared memory
            # in the layout of the step's first kernel and reach the GPU as ONE copy per batch (data/feed.py: PackedFeed); a captured
            # step takes such a batch with one device-side copy into its static buffers, sized for the feed's slots
            pf = feed.make_packed_feed(self.dataset, self.config, self.device, self.batch_size, sampler=sampler, shuffle=shuffle)
            self._feed_points_per_scan = pf.capacity // (2 * self.batch_size)
            return pf, sampler
        workers = int(self.config["num_dataloader_workers"])
        # worker processes decode whole batches ahead of the step (np.load + the [M,3] -> [1,3,M] transposition: ~3 ms per pair) and
        # stay alive between epochs; the loader's pinning thread copies each batch into page-locked memory, from where the
        # DevicePrefetcher's side stream takes it to the GPU while the previous step runs
        extra = {"prefetch_factor": int(self.config.get("dataloader_prefetch_factor", 4)), "persistent_workers": True} if workers > 0 else {}
        loader = torch.utils.data.DataLoader(dataset=self.dataset, batch_size=self.batch_size, shuffle=sampler is None and shuffle,
                                             sampler=sampler, collate_fn=Trainer.list_collate, drop_last=True,
                                             num_workers=workers, pin_memory=getattr(self.device, "type", "cpu") == "cuda", **extra)
        return loader, sampler

    def train(self, max_epochs=10000):
        dataloader, sampler = self.make_dataloader()
        self.steps_per_epoch_effective = len(dataloader)
        out_dir = self.config.get("checkpoint_dir", "/tmp")
        run = None
        if mlflow is not None and self.rank == 0:
            mlflow.set_experiment(self.config["experiment"])
            run = mlflow.start_run(run_name="Training: " + self.config["training_run_name"])
            for k, v in self.config.items():
                mlflow.log_param(k, v)
        if self.config.get("gc_freeze", True):
            # everything alive after the set-up (dataset index, module trees, the loader) moves to the cyclic collector's permanent
            # generation: a generation-2 pass over that heap stalls the host for ~0.1 s at random steps -- which a GPU-bound fp32
            # step hides and a 5 ms autocast step (or every rank behind a DDP all-reduce) does not
            import gc
            gc.collect()
            gc.freeze()
        self.history = []                           # per epoch: the reduced metrics + the phase they were computed in
        try:
            for epoch in range(max_epochs):
                if sampler is not None:
                    sampler.set_epoch(epoch)
                metrics = self._reduce_metrics(self.train_epoch(epoch=epoch, dataloader=dataloader))
                self.history.append(dict(metrics, epoch=epoch, unsupervised=bool(self.config["unsupervised_at_start"])))
                if self.rank == 0:
                    print("--------------------------")
                    print("Epoch Summary: " + format(epoch, "05d") + ", loss: " + str(metrics["loss_epoch"]) +
                          ", unsupervised: " + str(self.config["unsupervised_at_start"]) +
                          ", steps replayed as a HIP graph so far: " + str(getattr(self, "graph_steps", 0)))
                    names = {"loss": "loss_epoch", "loss point cloud": "loss_point_cloud_epoch", "loss po2po": "loss_po2po_epoch",
                             "loss po2pl": "loss_po2pl_epoch", "loss pl2pl": "loss_pl2pl_epoch", "visible pixels": "visible_pixels_epoch"}
                    if mlflow is not None:
                        for shown, key in names.items():
                            mlflow.log_metric(shown, float(metrics[key]), step=epoch)
                    else:
                        print({shown: metrics[key] for shown, key in names.items()})
                    # the reference writes the latest checkpoint every epoch (trainer.py:155-162) and keeps a copy every 5 (:164-173);
                    # `checkpoint_every` / `checkpoint_keep_every` (0 = never) thin that out for runs whose epochs take a second
                    every, keep = int(self.config.get("checkpoint_every", 1)), int(self.config.get("checkpoint_keep_every", 5))
                    latest = os.path.join(out_dir, self.config["training_run_name"] + "_latest_checkpoint.pth")
                    if (every > 0 and not epoch % every) or epoch == max_epochs - 1:
                        self.save_checkpoint(latest, epoch, metrics["loss_epoch"])
                        if mlflow is not None:
                            mlflow.log_artifact(latest)
                    if keep > 0 and not epoch % keep:
                        self.save_checkpoint(os.path.join(out_dir, self.config["training_run_name"] + "_checkpoint_epoch_" + str(epoch) + ".pth"),
                                             epoch, metrics["loss_epoch"])
                # identity pre-training ends once its loss is small (trainer.py:184-186); all ranks see the same reduced value
                if not self.config["unsupervised_at_start"] and metrics["loss_epoch"] < 1e-2:
                    self.config["unsupervised_at_start"] = True
                    if self.rank == 0:
                        print("Loss has decreased sufficiently. Switching to unsupervised mode.")
        finally:
            if run is not None:
                mlflow.end_run()
            if isinstance(dataloader, feed.PackedFeed) and self.rank == 0:
                steps = max(1, len(self.history) * len(dataloader))
                self.feed_report = {"page_locked_slots": bool(dataloader.pinned), "workers": dataloader.workers,
                                    "consumer_host_ms_per_step": {k: round(1e3 * v / steps, 3) for k, v in dataloader.host_seconds.items()}}
                print("[delora_amd] packed feed: " + str(self.feed_report))
            if hasattr(dataloader, "close"):
                dataloader.close()                  # PackedFeed: worker processes and page-locked slots
        return self.history
