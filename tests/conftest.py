import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is a chain of small torch ops: on the GPU box's many-core host the default thread count (all cores)
    # makes it an order of magnitude slower than 16 threads (bench.py measured 190 s vs ~10 s per pair).
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:            # pragma: no cover
        pass


@pytest.fixture(autouse=True)
def _module_switches_do_not_leak():
    """The kernels' module-level switches are process state: `Trainer._wrap_ddp` cuts the trunk into three autograd Functions for the rest
    of the process (ring_conv.TRUNK_SEGMENTS = "layer": same kernels, other slab boundaries in the merged weight-gradient launches), and a
    later test would silently run another summation order than it does on its own -- a long training run then follows another
    trajectory (round 5: the convergence test gave other figures inside the full suite than alone).  Restored after every test."""
    from delora_amd.models import ring_conv
    names = ("TRUNK_SEGMENTS", "USE_WINOGRAD", "USE_WINOGRAD_WGRAD", "WGRAD_BATCHED", "BACKWARD_TRACE", "ALLOC_SKEW", "WGRAD_PENDING_MAX_BYTES")
    saved = {n: getattr(ring_conv, n) for n in names if hasattr(ring_conv, n)}
    yield
    for n, v in saved.items():
        setattr(ring_conv, n, v)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Table of the deviations the parity tests measured (tests/util.py: measured()); also kept as JSON for profiles/."""
    try:
        from tests import util
    except Exception:          # pragma: no cover
        return
    if not util.MEASURED:
        return
    import json
    print("\n==== measured parity deviations (value / bound) ====")
    for k in sorted(util.MEASURED):
        v = util.MEASURED[k]
        print(f"  {k:78s} {v['value']:.6g}" + (f" / {v['bound']:.6g}" if v["bound"] is not None else ""))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.json"), "w") as f:
            json.dump(util.MEASURED, f, indent=1, sort_keys=True)
    except OSError:            # pragma: no cover
        pass
