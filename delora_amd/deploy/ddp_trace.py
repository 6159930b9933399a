"""Timeline of the gradient all-reduce of a DistributedDataParallel rank (measurement aid of bench.py --gpus N and of the multi-rank tests;
nothing in the training path imports it).

The path's only exchange is the bucketed all-reduce of the 47.5 MB of CNN gradients (reference: none -- it trains on one device;
SURVEY.md section 8e).  What a scaling run has to explain is how much of that exchange is EXPOSED, i.e. not hidden behind the rest of
the backward pass.  ``DdpTimeline.attach`` registers a communication hook that performs the default all-reduce and records, per bucket,
  * an event on the compute stream when the bucket is handed over (its last gradient has been produced),
  * an event when the bucket's all-reduce has completed (recorded in the future's callback: torch runs it under a stream that has
    waited for the collective),
and a hook on the parameter whose gradient is produced LAST (the stem's first convolution) marks the end of the backward computation.
``exposed_ms`` of a step = completion of the last bucket - end of the backward computation (clamped at 0): the time the optimiser has to
wait for the network.  Events are read after a synchronisation, so recording costs the host ~10 us per bucket and nothing on the GPU.
On a CPU process group (gloo tests) host clocks take the place of the events."""
import time

import torch
import torch.distributed as dist


class DdpTimeline:
    def __init__(self, keep_steps=64):
        self.keep_steps = int(keep_steps)
        self.steps = []                 # per step: {"t0": ev, "last_grad": ev, "buckets": [(index, bytes, ready_ev, done_ev)]}
        self._cur = None
        self.enabled = True
        self.cuda = False
        self.world = 1

    # ------------------------------------------------------------------------------------------------------------ recording
    def _now(self):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return time.perf_counter()

    def attach(self, ddp_model, last_grad_param=None, process_group=None):
        """Register the hook on ``ddp_model`` (a DistributedDataParallel).  ``last_grad_param``: the parameter whose gradient is produced
        last in the backward pass (default: the first parameter of the wrapped module)."""
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = list(ddp_model.module.parameters())
        self.cuda = bool(params) and params[0].is_cuda
        p_last = last_grad_param if last_grad_param is not None else params[0]

        def on_last_grad(_p):
            if self.enabled and self._cur is not None:
                self._cur["last_grad"] = self._now()
        p_last.register_post_accumulate_grad_hook(on_last_grad)
        timeline = self

        def hook(state, bucket):
            buf = bucket.buffer()
            rec = None
            if timeline.enabled and timeline._cur is not None:
                rec = [bucket.index(), buf.numel() * buf.element_size(), timeline._now(), None]
                timeline._cur["buckets"].append(rec)
            buf.div_(timeline.world)
            fut = dist.all_reduce(buf, group=process_group, async_op=True).get_future()

            def done(f):
                if rec is not None:
                    rec[3] = timeline._now()
                return f.value()[0]
            return fut.then(done)
        ddp_model.register_comm_hook(None, hook)
        return self

    def begin_step(self):
        if not self.enabled:
            return
        self._cur = {"t0": self._now(), "last_grad": None, "buckets": []}

    def end_step(self):
        """Call after the optimiser step has been enqueued."""
        if not self.enabled or self._cur is None:
            return
        self._cur["t1"] = self._now()
        self.steps.append(self._cur)
        self._cur = None
        if len(self.steps) > self.keep_steps:
            del self.steps[0]

    # ------------------------------------------------------------------------------------------------------------ reading
    def _ms(self, a, b):
        if a is None or b is None:
            return None
        return a.elapsed_time(b) if self.cuda else 1e3 * (b - a)

    def summary(self, last=16):
        """Mean over the last ``last`` recorded steps (synchronises the device): exposed all-reduce time, the bucket timeline relative to
        the start of the step, bytes."""
        if self.cuda:
            torch.cuda.synchronize()
        rows = [s for s in self.steps[-int(last):] if s["buckets"] and s.get("last_grad") is not None]
        if not rows:
            return {"steps": 0}
        exposed, bwd_end, step_ms, tail_bytes = [], [], [], []
        nb = max(len(s["buckets"]) for s in rows)
        ready = [[] for _ in range(nb)]
        done = [[] for _ in range(nb)]
        size = [0] * nb
        for s in rows:
            t_last = self._ms(s["t0"], s["last_grad"])
            bwd_end.append(t_last)
            step_ms.append(self._ms(s["t0"], s["t1"]))
            fin = [self._ms(s["t0"], b[3]) for b in s["buckets"] if b[3] is not None]
            exposed.append(max(0.0, max(fin) - t_last) if fin else None)
            tail_bytes.append(sum(b[1] for b in s["buckets"] if self._ms(s["last_grad"], b[2]) is not None and self._ms(s["last_grad"], b[2]) >= -1e-3))
            for j, b in enumerate(s["buckets"]):
                ready[j].append(self._ms(s["t0"], b[2]))
                if b[3] is not None:
                    done[j].append(self._ms(s["t0"], b[3]))
                size[j] = b[1]

        def mean(v):
            v = [x for x in v if x is not None]
            return round(sum(v) / len(v), 3) if v else None
        return {"steps": len(rows), "exposed_allreduce_ms": mean(exposed), "backward_end_ms": mean(bwd_end), "step_enqueued_to_optimizer_ms": mean(step_ms),
                "bytes_per_step": int(sum(size)), "bytes_handed_over_at_or_after_the_last_gradient": int(sum(tail_bytes) / len(tail_bytes)),
                "buckets": [{"index": j, "bytes": int(size[j]), "ready_ms": mean(ready[j]), "done_ms": mean(done[j])} for j in range(nb)],
                "clock": "HIP events on the compute stream" if self.cuda else "host clock (CPU process group)"}
