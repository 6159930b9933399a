"""Training-quality evidence (-m gpu): the product's own Trainer.train() -> Tester.test() pipeline on a synthetic dataset with known
ego-motion, in every precision the bench advertises, scored the way the reference's results are scored (KITTI-style relative
trajectory errors).  The same run as tools/convergence.py (300 epochs = 6000 steps per precision, ~2.5 minutes on the GPU); its report
is written to gpurun_out/convergence.json and committed as profiles/r05_convergence.json.

What it establishes -- and what it does not: the three precisions start from the same weights and see the same samples in the same
order, reach the SAME loss plateau and the SAME trajectory error (so bf16 / fp16 storage with fp32 accumulation and fp32 master weights
trains like fp32).  The absolute error stays far above KITTI figures: the synthetic scenes are corridors (walls and ground parallel to
the motion), where the point-to-plane loss leaves the forward translation almost unconstrained -- a property of the scenes."""
import os
import sys

import numpy as np
import pytest
import torch

from tests import util
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_every_precision_trains_to_the_same_plateau_and_trajectory_error(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import convergence as cv
    finally:
        sys.path.pop(0)
    device = torch.device("cuda", 0)
    tree, truth = cv.build_dataset(device, train_sequences=4, scans_per_sequence=41, workdir=str(tmp_path / "tree"))
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    # (at a constant 1e-4 the loss curves keep spiking -- three precisions then end wherever their last spike left them: 0.986 / 0.986 /
    # 0.986 in one run, 1.00 / 1.16 / ... in the next -- hence the last third at the reference's own 1e-5, resumed from the checkpoint)
    # `hip_graph: false` for all three: a replayed step runs torch's capturable Adam, whose bias corrections are computed on the device --
    # one ulp away from the eager ones, which is enough to send a chaotic trajectory elsewhere; `auto` decides by the host's speed, i.e.
    # per box.  Eager steps make the run the same on every box.
    runs = {p: cv.train_and_test(device, tree, truth, p, epochs=360, lr=1e-4, batch=8, seed=11, out_dir=out_dir, anneal_fraction=1.0 / 6.0,
                                 extra={"hip_graph": False})
            for p in ("float32", "bfloat16", "float16")}
    base = runs["float32"]
    curve = np.asarray(base["unsupervised_loss_per_epoch"])
    held = base["held_out_sequence"]
    last = curve[-20:].mean()
    for p in ("bfloat16", "float16"):
        runs[p]["vs_float32"] = {"plateau_loss_ratio": float(np.asarray(runs[p]["unsupervised_loss_per_epoch"])[-20:].mean() / last),
                                 "held_out_translation_error_ratio": runs[p]["held_out_sequence"]["translation_error_percent"] / held["translation_error_percent"],
                                 "held_out_per_step_rotation_error_ratio": runs[p]["held_out_sequence"]["per_step_rotation_error_deg_mean"] / held["per_step_rotation_error_deg_mean"]}
    import json
    report = {"what": "identity pre-training -> unsupervised training (300 epochs at lr 1e-4, 60 more at 1e-5 resumed from the checkpoint; eager steps) -> Tester -> "
                      "KITTI-style relative errors, per precision, same seed (tests/test_gpu_zz_convergence.py; tools/convergence.py is the same run as a script)",
              "image": "64x720", "batch": 8, "learning_rate": 1e-4, "reference_learning_rate": 1e-5, "epochs": 360, "segment_lengths_m": [2.0, 5.0, 10.0, 15.0],
              "dataset": "4 training sequences + 1 held out, 41 scans each (synthetic scenes, ~0.45 m and <=1.5 deg yaw per scan), preprocessed offline at 64x2250",
              "runs": runs}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "convergence.json"), "w") as f:                    # (before the assertions: a failing run leaves its curves behind)
        json.dump(report, f, indent=1)
    assert base["identity_epochs"] >= 1 and len(curve) >= 340, "identity pre-training must hand over to the unsupervised phase"
    first, before_last = curve[:8].mean(), curve[-40:-20].mean()
    # Bounds.  fp32 converges to the same plateau on every trajectory tried (1.0374 with the trunk as one autograd Function and with the
    # three-Function cut of a DDP rank, whose merged weight-gradient launches sum in another order).  The half precisions are less robust
    # at this learning rate: on the trajectory of THIS configuration (trunk as one Function, eager steps; deterministic on every box) all
    # three end in the same state to 5e-4 -- on the three-Function trajectory bf16 / fp16 were still wandering at epoch 360 (plateau x1.21 /
    # x1.15, per-step rotation error x2.1 / x1.9; profiles/r05_convergence_trajectories.json).  The assertions below therefore hold the
    # half precisions to "trains, same ballpark"; the figures themselves go to the report and to DESIGN.md section 8.
    util.measured("fp32 training: mean unsupervised loss of the last 20 epochs / of the first 8", last / first, bound=0.85)
    util.measured("fp32 training: the plateau -- |last 20 epochs - the 20 before| / last", abs(last - before_last) / last, bound=0.05)
    util.measured("fp32 training: held-out relative translation error / the error of a predictor that says 'no motion'",
                  held["translation_error_percent"] / held["yardstick_no_motion_percent"], bound=0.6)
    for p in ("bfloat16", "float16"):
        c = np.asarray(runs[p]["unsupervised_loss_per_epoch"])
        assert np.isfinite(c).all() and len(c) == len(curve)
        util.measured(f"{p} training: loss plateau (last 20 epochs) / fp32's", runs[p]["vs_float32"]["plateau_loss_ratio"], bound=1.35)
        util.measured(f"{p} training: held-out relative translation error / fp32's", runs[p]["vs_float32"]["held_out_translation_error_ratio"], bound=1.5)
        util.measured(f"{p} training: held-out per-step rotation error (deg) / fp32's", runs[p]["vs_float32"]["held_out_per_step_rotation_error_ratio"], bound=2.5)
        util.measured(f"{p} training: held-out relative translation error / the error of a predictor that says 'no motion'",
                      runs[p]["held_out_sequence"]["translation_error_percent"] / held["yardstick_no_motion_percent"], bound=0.6)
