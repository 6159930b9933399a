#!/usr/bin/env python3
"""Headline benchmark: DeLORA training scan-pairs/s on synthetic KITTI-shaped 64x2048 input.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full training step over a batch of B scan pairs per GPU (BASELINE.json configs[1]: 64x2048, B=8,
fp32): spherical projection of the 2B raw scans, per-pixel normals from the projected images, the pose CNN forward,
quaternion -> T, exact nearest-neighbour correspondences, the fused point-to-plane/plane-to-plane loss, backward
through loss and CNN, gradient all-reduce (N>1) and the Adam update.  Inputs (raw scan point lists) are resident in
HBM when the timed region starts; the timed steps ROTATE over `--rotate` distinct batches (ragged scan lengths) so
that no step finds the previous step's intermediates of the same data in a cache.  Rank 0 prints ONE JSON line
(contract in the task statement) that also carries
  "roofline":     the dominant kernel, k_wino_conv (fused Winograd F(2x2,3x3) on the fp32 matrix cores; 36 % of the step):
                  multiply-adds issued / its duration, measured in the timed steps themselves (begin/end timestamps on HIP
                  events attached to every launch), vs the 157.3 TFLOP/s fp32 MFMA peak
  "roofline_loss": the fused ICP loss kernel (HBM-bound): algorithmic bytes / its duration behind a cache flush (cold: operands from
                  HBM; with clean and with dirty foreign lines in the infinity cache) AND in the timed steps (warm), vs 8 TB/s
  "roofline_cnn": every stride-1 layer shape and pass, one launch each
  "long_run":     the same loop for `--long-steps` (default 200) steps
  "feed":         the same steps fed from pinned host memory through DataLoader + DevicePrefetcher (H2D in the loop)
  "cpu_baseline": the reference-like step (stored normal lists, B=1) evaluated by the CPU oracle on a bounded sample;
  "cpu_baseline_online_normals": the GPU workload itself (normals computed in the step) on the CPU
  "kernels":      per-launch HIP-event times and achieved bandwidth of every geometry kernel (back-to-back launches).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s measured copy rate)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32: 256 flop/clk/CU x 256 CUs x 2.4 GHz)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 matrix peak (v_mfma_f32_32x32x16_bf16)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="scan pairs per GPU per step")
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--rotate", type=int, default=8, help="distinct batches the steps rotate over (HBM resident)")
    ap.add_argument("--long-steps", type=int, default=200, help="extra timed run of this many steps (0 = skip); N=1 only")
    ap.add_argument("--feed-steps", type=int, default=64, help="steps of the host-fed leg (0 = skip); N=1 only")
    ap.add_argument("--alloc-skew", type=int, default=-1, help="A/B: base-address skew (bytes) of the trunk's activation tensors (-1: library default)")
    ap.add_argument("--trunk-segments", default="", help="A/B: how the trunk is cut into autograd Functions (mono / layer / block; default: the library's choice)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes (roofline.traffic then comes from the committed bundle)")
    ap.add_argument("--no-profile", action="store_true", help="no event-carrying launches in the timed steps (no roofline object)")
    ap.add_argument("--conv-table", action="store_true", help="also time every stride-1 layer shape and pass back to back (roofline_cnn)")
    ap.add_argument("--autocast-steps", type=int, default=20, help="timed steps of the extra bf16-autocast leg of the fp32 run (0 = skip); N=1 only")
    ap.add_argument("--disk-pairs", type=int, default=32, help="scan pairs of the synthetic on-disk sequence of the `feed_disk` leg (0 = skip); N=1 only")
    ap.add_argument("--disk-workers", type=int, default=2, help="DataLoader worker processes of the `feed_disk` leg")
    ap.add_argument("--variant-steps", type=int, default=20, help="timed steps of the extra fp32 legs `shipped_image` (64x720, the reference's "
                    "default KITTI image) and `untrained_network` (randomly initialised heads: whole-image search) (0 = skip); N=1 only")
    ap.add_argument("--point-order", default="raster", choices=("raster", "firing", "shuffled"),
                    help="order of the points inside a raw scan: raster = ring after ring, the order of the reference's stored point lists "
                         "(src/preprocessing/preprocesser.py:60-67; the default since round 5), firing = the order a spinning sensor delivers, "
                         "shuffled = a random permutation (rounds 1-4: the worst case of the projection's vote; still a row of `kernels`)")
    ap.add_argument("--ddp-steps", type=int, default=20, help="timed steps of the `ddp_rank` leg: one rank wrapped in DistributedDataParallel over a "
                    "one-rank RCCL group, fp32 and bf16, against the unwrapped step (0 = skip); N=1 only")
    ap.add_argument("--shipped-steps", type=int, default=200, help="timed batch-1 steps of the `shipped_config` leg: the reference's default operating "
                    "point (unmodified YAML: 64x720, batch 1, stored lists), eager / graph / product loop from disk (0 = skip); N=1 only")
    ap.add_argument("--amp", default="", help="optional autocast dtype for the CNN (bfloat16/float16); default fp32 = parity mode")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--cnn", default="", help="CNN implementation override (config key cnn_impl)")
    ap.add_argument("--miopen-benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True (exhaustive MIOpen find)")
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one captured HIP graph (single GPU, one batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=2, help="pairs evaluated by the CPU baseline (stored-normals leg)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch CPU threads of the baseline (more is slower on a 256-thread host: "
                    "the step is a chain of small ops; 128 threads measured 190 s/pair vs ~10 s/pair)")
    ap.add_argument("--kernel-reps", type=int, default=50)
    return ap.parse_args(argv)


def build_config(args, device):
    from delora_amd import config as cfgmod
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfg["datasets"] = ["kitti"]
    cfgmod.degrees_to_radians(cfg)
    cfg["kitti"]["data_identifiers"] = cfg["kitti"]["training_identifiers"]
    cfg["kitti"]["vertical_cells"], cfg["kitti"]["horizontal_cells"] = args.height, args.width
    cfg.update(device=device, batch_size=args.batch, unsupervised_at_start=True, inference_only=False, checkpoint=None,
               training_run_name="bench", run_name="bench", mode="training")
    if args.amp:
        cfg["amp_dtype"] = args.amp
    if args.channels_last:
        cfg["channels_last"] = True
    if getattr(args, "cnn", ""):
        cfg["cnn_impl"] = args.cnn
    return cfg


def make_batch(args, rank, device=None):
    """B synthetic pairs for this rank (seed = 1000*config + global sample index; config = 2)."""
    from delora_amd.data import synthetic
    samples = []
    for j in range(args.batch):
        s1, s2, T = synthetic.make_pair(2000 + rank * args.batch + j, rings=args.height, azimuth_steps=2250,
                                        point_order=getattr(args, "point_order", "raster"))       # (the tools pass ad-hoc argument objects)
        d = {"dataset": "kitti", "scan_1": torch.from_numpy(s1).unsqueeze(0), "scan_2": torch.from_numpy(s2).unsqueeze(0),
             "normal_list_1": None, "normal_list_2": None}
        if device is not None:
            d["scan_1"], d["scan_2"] = d["scan_1"].to(device), d["scan_2"].to(device)
        samples.append(d)
    return samples


def derived_batches(base, count, rank, shuffled=False):
    """`count` distinct host batches from one ray-cast batch (ray casting costs ~0.4 s per pair): batch 0 is `base`, batch k
    is the same scenes seen under a different heading -- both scans of a pair rotated about the vertical axis by the same
    angle, which moves every point to another pixel column and keeps the pair's relative motion small -- with a different
    random 0-3 % of the points dropped (ragged scan lengths); the remaining points keep their order (``shuffled``: re-permuted)."""
    out = [[{k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()} for d in base]]
    for k in range(1, count):
        rng = np.random.default_rng(77000 + 100 * rank + k)
        batch = []
        for d in base:
            yaw = 2.0 * np.pi * k / count + rng.uniform(-0.2, 0.2)
            c, s = np.cos(yaw), np.sin(yaw)
            R = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
            e = dict(d)
            for name in ("scan_1", "scan_2"):
                pts = d[name][0]
                n = pts.shape[1]
                keep = rng.permutation(n)[: n - int(rng.integers(0, max(1, int(0.03 * n))))]
                keep = torch.from_numpy(keep if shuffled else np.sort(keep))     # dropped points leave the order of the others alone
                e[name] = (R @ pts[:, keep]).unsqueeze(0).contiguous()
            batch.append(e)
        out.append(batch)
    return out


def pin_batches(batches):
    for b in batches:
        for d in b:
            for k, v in d.items():
                if torch.is_tensor(v) and not v.is_cuda:
                    d[k] = v.pin_memory()
    return batches


def to_device(batch, device):
    return [{k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in d.items()} for d in batch]


def identity_pretrained_state(model):
    """Put the randomly initialised network into the state the reference's own first training phase leaves it in
    (identity fitting until its loss < 1e-2, src/deploy/trainer.py:184-186): it predicts T = I.  Done by zeroing the last
    linear layer of both heads and biasing the quaternion to (0,0,0,1); every other weight keeps its random value, so
    the forward/backward cost is unchanged.  Without this an untrained network predicts a random rotation, and the
    correspondence search of every step degenerates to its exhaustive fallback -- a regime real training never sees."""
    with torch.no_grad():
        rot, tra = model.fully_connected_rotation[-1], model.fully_connected_translation[-1]
        rot.weight.zero_(); rot.bias.copy_(torch.tensor([0.0, 0.0, 0.0, 1.0]))
        tra.weight.zero_(); tra.bias.zero_()


def cnn_impl_in_use(trainer, args):
    """Which convolution path the timed steps ran (models/resnet_modified.py decides per call)."""
    from delora_amd.models import ring_conv
    r = trainer.raw_model.resnet
    x = torch.zeros(1, device=trainer.device)
    if not args.amp and r.hip_trunk_applicable((args.batch, args.height, args.width // 4, r.layer1[0].conv1.in_channels), x):
        return "hip trunk: fp32 MFMA, " + ("fused Winograd F(2x2,3x3) + direct implicit GEMM" if ring_conv.USE_WINOGRAD else "direct implicit GEMM")
    if args.amp:
        with torch.autocast("cuda", dtype=getattr(torch, args.amp)):
            if r.hip_half_applicable(torch.zeros((args.batch, 8, args.height, args.width), device=trainer.device)) is not None:
                return f"hip trunk: {args.amp} MFMA (v_mfma_f32_32x32x16), LDS-DMA implicit GEMM + transposing-read weight gradient; fp32 stem"
    return "modules: library convolutions + fused ring ops"


def autocast_leg(args, device, host_batches, batches, timed_region, tree=None):
    """The same training step with the CNN in half precision (config key amp_dtype = torch.autocast around the model call: the
    mixed-precision mode BASELINE.json configs[4] names; the reference itself trains in fp32 only): a second trainer on the same batches, the trunk on
    the bf16 MFMA kernels (csrc/convh.hip, wgradh.hip).  Reported next to the fp32 headline, never as it."""
    import argparse
    from delora_amd.deploy.trainer import Trainer
    from delora_amd.data.dataset import ListDataset
    a2 = argparse.Namespace(**vars(args))
    a2.amp = "bfloat16"
    cfg = build_config(a2, device)
    torch.manual_seed(1234)
    trainer = Trainer(cfg, dataset=ListDataset([d for b in host_batches for d in b]))
    identity_pretrained_state(trainer.raw_model)
    counter = {"i": 0}

    def step():
        batch = batches[counter["i"] % len(batches)]
        counter["i"] += 1
        trainer.optimizer.zero_grad(set_to_none=True)
        ep, _ = trainer.step(preprocessed_dicts=[dict(s) for s in batch], epoch_losses=trainer.new_epoch_losses())
        return ep
    for _ in range(max(3, args.warmup)):
        step()
    import gc
    gc.collect()
    gc.freeze()                                                  # as Trainer.train does after its set-up (see main)
    el, ep = timed_region(args.autocast_steps, step)
    out_disk = None
    if tree is not None:                                         # the 5 ms step fed from the on-disk sequence (see disk_feed_leg)
        def step_on(batch):                                       # (a PackedBatch of the feed)
            trainer.optimizer.zero_grad(set_to_none=True)
            return trainer.step(preprocessed_dicts=batch, epoch_losses=trainer.new_epoch_losses())[0]
        try:
            out_disk = disk_feed_leg(a2, cfg, device, tree, step_on, timed_region, args.batch * args.autocast_steps / el, max(args.autocast_steps, 40))
        except Exception as e:                                   # noqa: BLE001 -- the leg is informative
            out_disk = {"error": f"{type(e).__name__}: {e}"}
    out_graph = None
    try:
        from delora_amd.deploy.graph_step import GraphedStep
        graphed = GraphedStep(trainer, batches[0])
        if graphed.captured:
            def gstep():
                batch = batches[counter["i"] % len(batches)]
                counter["i"] += 1
                return graphed(batch)[0]
            for _ in range(3):
                gstep()
            elg, epg = timed_region(args.autocast_steps, gstep)
            out_graph = {"value": round(args.batch * args.autocast_steps / elg, 3), "ms_per_step": round(1e3 * elg / args.autocast_steps, 3),
                         "final_loss": float(epg["loss_epoch"]), "eager_fallback_steps": graphed.fallback_steps,
                         "note": "the same steps replayed as ONE captured HIP graph; the rotating ragged batches go through static buffers "
                                 "(deploy/graph_step.py, config key hip_graph)"}
    except Exception as e:                                       # noqa: BLE001 -- the leg is informative
        out_graph = {"error": f"{type(e).__name__}: {e}"}
    return {"dtype": "bfloat16", "steps": args.autocast_steps, "value": round(args.batch * args.autocast_steps / el, 3), "unit": "scan-pairs/s",
            "ms_per_step": round(1e3 * el / args.autocast_steps, 3), "final_loss": float(ep["loss_epoch"]), "cnn_impl": cnn_impl_in_use(trainer, a2),
            "hip_graph": out_graph, "feed_disk": out_disk,
            "note": "autocast(bfloat16) around the pose CNN: fp32 stem, layer1-4 on v_mfma_f32_32x32x16_bf16 with bf16 activations, fp32 "
                    "accumulation, fp32 master weights and weight gradients; geometry kernels, loss and Adam unchanged (fp32)"}


def variant_leg(args, device, host_batches, batches, timed_region, steps, width=None, pretrained=True):
    """The fp32 training step on a variant of the workload, a second trainer on the same raw scans: ``width`` = another image width
    (720: the reference's shipped KITTI image, config/config_datasets.yaml:21 -- feature maps 180 / 90 / 45 / 23 pixels wide, i.e.
    overhanging tiles and a stride-2 layer on an odd width), ``pretrained`` False = the randomly initialised network of a run from
    scratch (`unsupervised_at_start: True` without a checkpoint): it predicts a random rotation, so every step's correspondence search
    is the whole-image walk."""
    import argparse
    from delora_amd.deploy.trainer import Trainer
    from delora_amd.data.dataset import ListDataset
    a2 = argparse.Namespace(**vars(args))
    a2.amp = ""
    if width:
        a2.width = int(width)
    cfg = build_config(a2, device)
    torch.manual_seed(1234)
    trainer = Trainer(cfg, dataset=ListDataset([d for b in host_batches for d in b]))
    if pretrained:
        identity_pretrained_state(trainer.raw_model)
    counter = {"i": 0}

    def step():
        batch = batches[counter["i"] % len(batches)]
        counter["i"] += 1
        trainer.optimizer.zero_grad(set_to_none=True)
        ep, _ = trainer.step(preprocessed_dicts=[dict(s) for s in batch], epoch_losses=trainer.new_epoch_losses())
        return ep
    for b in batches:                                            # priming: every ragged batch once (allocator), as main() does
        trainer.optimizer.zero_grad(set_to_none=True)
        trainer.step(preprocessed_dicts=[dict(s) for s in b], epoch_losses=trainer.new_epoch_losses())
    el, ep = timed_region(steps, step)
    return {"steps": steps, "value": round(args.batch * steps / el, 3), "unit": "scan-pairs/s", "ms_per_step": round(1e3 * el / steps, 3),
            "image": f"{a2.height}x{a2.width}", "final_loss": float(ep["loss_epoch"]), "cnn_impl": cnn_impl_in_use(trainer, a2),
            "network_state": "identity-pretrained" if pretrained else "random initialisation (no identity pre-training)"}


def make_disk_tree(args):
    """A synthetic SEQUENCE written in the reference's on-disk training format (<path>/<seq>/scans/<idx>.npy,
    src/preprocessing/preprocesser.py:64-68; xyz only -- the normals of this workload are computed online)."""
    import tempfile
    from delora_amd.data import synthetic
    tmp = tempfile.mkdtemp(prefix="delora_feed_")
    t0 = time.perf_counter()
    scans, _ = synthetic.make_sequence(4000, args.disk_pairs + 1, rings=args.height, azimuth_steps=2250, point_order=args.point_order)
    synthetic.write_tree(tmp, scans, sequence=0)
    nbytes = sum(os.path.getsize(os.path.join(tmp, "00", "scans", f)) for f in os.listdir(os.path.join(tmp, "00", "scans")))
    return {"path": tmp, "bytes": nbytes, "generation_s": round(time.perf_counter() - t0, 1)}


def disk_feed_leg(args, cfg, device, tree, run_step, timed_region, resident_pairs_s, steps):
    """SURVEY.md 8f-1, for real: the tree of `make_disk_tree` is read back by `PreprocessedPointCloudDataset` (src/data/dataset.py:82-154:
    not kept in RAM) and fed the way `Trainer.train` feeds it when `num_dataloader_workers` > 0 (data/feed.py: PackedFeed -- worker
    processes decode the files straight into page-locked shared batch slots in the layout of the step's first kernel, consecutive pairs
    share their scan, ONE host-to-device copy per batch on a side stream, one batch ahead) -- shuffled, as a training set is read.
    The timed steps train on it; reported against the resident rate of the same step."""
    from delora_amd.data import feed as feedmod
    from delora_amd.data.dataset import PreprocessedPointCloudDataset
    dcfg = dict(cfg)
    dcfg["kitti"] = dict(cfg["kitti"], preprocessed_path=tree["path"], data_identifiers=[0])
    dcfg.update(store_dataset_in_RAM=False, num_dataloader_workers=args.disk_workers, load_normal_lists=False)
    if os.environ.get("DELORA_FEED_AHEAD"):
        dcfg["feed_batches_ahead"] = int(os.environ["DELORA_FEED_AHEAD"])
    ds = PreprocessedPointCloudDataset(dcfg)
    assert feedmod.packed_feed_applicable(ds, dcfg, device), "the bench's on-disk leg must take the product's packed feed"
    pf = feedmod.make_packed_feed(ds, dcfg, device, args.batch, shuffle=True)
    try:
        def epochs():
            while True:
                for b in pf:
                    yield b
        # the yardstick: the same step on the same data, resident -- one epoch of the feed's batches kept on the device (taken BEFORE the
        # endless iterator starts: a feed serves one consumer at a time)
        kept = [b for b in pf]
        it = epochs()

        def fed_step():
            return run_step(next(it))
        for _ in range(max(3, 2 * len(pf))):                     # workers up, page cache and allocator primed with this data's sizes
            fed_step()
        k = {"i": 0}

        def resident_step():
            k["i"] += 1
            return run_step(kept[k["i"] % len(kept)])
        for _ in range(3):
            resident_step()
        el_r, _ = timed_region(steps, resident_step)              # A - B - A: resident, fed, resident (clock and thermal drift cancel)
        pf.bytes_moved = 0
        for k_ in pf.host_seconds:
            pf.host_seconds[k_] = 0.0
        el, _ = timed_region(steps, fed_step)
        rate = args.batch * steps / el
        host_ms = {k_: round(1e3 * v_ / steps, 3) for k_, v_ in pf.host_seconds.items()}
        for _ in range(3):
            resident_step()
        el_r2, _ = timed_region(steps, resident_step)
        resident_same = args.batch * steps / (0.5 * (el_r + el_r2))
        del kept
        return {"steps": steps, "value": round(rate, 3), "unit": "scan-pairs/s", "ms_per_step": round(1e3 * el / steps, 3),
                "resident_same_data": round(resident_same, 3), "vs_resident": round(rate / resident_same, 4),
                "vs_headline_workload": round(rate / resident_pairs_s, 4), "feed_GB_s": round(pf.bytes_moved / el / 1e9, 3),
                "dataset": f"{args.disk_pairs} consecutive pairs of one synthetic sequence, {tree['bytes'] / 1e6:.0f} MB on disk, xyz only, the "
                           f"reference's layout, store_dataset_in_RAM False", "workers": args.disk_workers, "shuffle": True,
                "slots_page_locked": bool(pf.pinned), "consumer_host_ms_per_step": host_ms, "generation_s": tree["generation_s"],
                "note": "PreprocessedPointCloudDataset -> PackedFeed (worker processes decode into page-locked shared batch slots; one "
                        "H2D copy per batch, one batch ahead) -> the same training step; the files sit in the page cache after the first "
                        "epoch, as a training set that fits in RAM does"}
    finally:
        pf.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's DEFAULT operating point: config/*.yaml unmodified -- 64x720 images, batch_size 1, stored point + normal lists read
# from <preprocessed_path>/<seq>/{scans,normals}/ (config/hyperparameters.yaml:3, config_datasets.yaml:21, src/deploy/trainer.py:43-91).
# ~100 launches for ~1 ms of GPU work: the eager step is bound by the host's enqueue time, which is what `hip_graph: auto` removes.
def shipped_tree(device, n_pairs, seed=4100):
    """A synthetic sequence through the OFFLINE PREPROCESSING of the reference (projection at horizontal_cells_preprocessing = 2250 +
    normals, src/preprocessing/preprocesser.py:52-68: raster-ordered [M,3] point and normal lists) into a temporary tree."""
    import tempfile
    from delora_amd import config as cfgmod
    from delora_amd.data import synthetic
    from delora_amd.preprocessing.preprocesser import Preprocesser
    tmp = tempfile.mkdtemp(prefix="delora_shipped_")
    t0 = time.perf_counter()
    scans, poses = synthetic.make_sequence(seed, n_pairs + 1, rings=64, azimuth_steps=2250, point_order="firing")    # raw scans as a driver delivers them
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfgmod.degrees_to_radians(cfg)
    cfg["device"] = device
    cfg["kitti"]["preprocessed_path"] = tmp
    Preprocesser(cfg).preprocess_scans(scans, "kitti", 0)
    torch.cuda.synchronize()
    nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
    return {"path": tmp, "pairs": n_pairs, "bytes": nbytes, "generation_s": round(time.perf_counter() - t0, 1), "poses": poses}


def shipped_trainer(device, tree_path, batch, workers=0, hip_graph="auto", amp=""):
    """A Trainer on this repository's YAML as it is, plus what a user has to set anyway (where the data are) and the leg's variables."""
    from delora_amd import config as cfgmod
    from delora_amd.deploy.trainer import Trainer
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfgmod.degrees_to_radians(cfg)
    cfg["kitti"]["preprocessed_path"] = tree_path
    cfg["kitti"]["data_identifiers"] = [0]
    cfg.update(device=device, batch_size=batch, unsupervised_at_start=True, inference_only=False, checkpoint=None, mode="training",
               training_run_name="bench_shipped", run_name="bench_shipped", num_dataloader_workers=workers, hip_graph=hip_graph)
    if amp:
        cfg["amp_dtype"] = amp
    torch.manual_seed(1234)
    trainer = Trainer(cfg)
    identity_pretrained_state(trainer.raw_model)
    return trainer


def shipped_resident_batches(trainer, count):
    """`count` batches of the trainer's dataset, on the device, as the lists of sample dicts the DataLoader would deliver."""
    B = trainer.batch_size
    n = len(trainer.dataset)
    return [trainer.to_device([trainer.dataset[(k * B + j) % n] for j in range(B)]) for k in range(count)]


def shipped_config_leg(args, device, timed_region, enqueue, steps_b1=200, steps_b8=40, pairs=32):
    """Every way the reference's default operating point can run here, B = 1 (the YAML's value) and B = 8: the eager step and the
    replayed graph on HBM-resident batches (rotating), and the PRODUCT LOOP -- `Trainer.train_epoch` with `hip_graph: auto`, which
    measures and decides by itself -- fed from the on-disk tree through `PackedFeed` (2 workers) and through the YAML's literal
    `num_dataloader_workers: 0` (in-process DataLoader + DevicePrefetcher).  Each entry carries the host's enqueue time per step."""
    import gc, shutil
    from delora_amd.deploy.graph_step import GraphedStep
    tree = shipped_tree(device, pairs)
    out = {"image": "64x720", "data": f"{pairs} consecutive pairs of one synthetic sequence, preprocessed offline at 64x2250 (stored point + "
                                       f"normal lists, raster order, {tree['bytes'] / 1e6:.0f} MB on disk)",
           "reference": "config/hyperparameters.yaml:3 (batch_size 1), config/config_datasets.yaml:21 (64x720), src/deploy/trainer.py:43-91"}
    try:
        for B, steps in ((1, steps_b1), (8, steps_b8)):
            leg = {}
            trainer = shipped_trainer(device, tree["path"], B, workers=0, hip_graph="off")
            batches = shipped_resident_batches(trainer, 8 if B == 1 else 4)
            k = {"i": 0}

            def eager_step():
                k["i"] += 1
                trainer.optimizer.zero_grad(set_to_none=True)
                return trainer.step(preprocessed_dicts=[dict(d) for d in batches[k["i"] % len(batches)]], epoch_losses=trainer.new_epoch_losses())[0]
            for _ in range(len(batches) + 3):
                eager_step()
            gc.collect()
            el, ep = timed_region(steps, eager_step)
            leg["eager_resident"] = {"ms_per_step": round(1e3 * el / steps, 4), "value": round(B * steps / el, 2),
                                     "host_enqueue_ms_per_step": round(enqueue["ms_per_step"], 4), "final_loss": float(ep["loss_epoch"])}
            graphed = GraphedStep(trainer, batches[0])
            if graphed.captured:
                def graph_step():
                    k["i"] += 1
                    return graphed(batches[k["i"] % len(batches)])[0]
                for _ in range(5):
                    graph_step()
                el, ep = timed_region(steps, graph_step)
                leg["graph_resident"] = {"ms_per_step": round(1e3 * el / steps, 4), "value": round(B * steps / el, 2),
                                         "host_enqueue_ms_per_step": round(enqueue["ms_per_step"], 4), "final_loss": float(ep["loss_epoch"]),
                                         "eager_fallback_steps": graphed.fallback_steps}
                # the GPU's own time for one replay: the same batch replayed back to back (no packing, the host far ahead)
                for _ in range(3):
                    graphed(None)
                el, _ = timed_region(steps, lambda: graphed(None)[0])
                leg["graph_replay_only_ms"] = round(1e3 * el / steps, 4)
            del graphed, batches, trainer
            gc.collect()
            torch.cuda.empty_cache()
            for name, workers in (("product_loop_packed_feed_2_workers", 2), ("product_loop_yaml_default_0_workers", 0)):
                trainer = shipped_trainer(device, tree["path"], B, workers=workers, hip_graph="auto")
                loader, _ = trainer.make_dataloader()
                per_epoch = len(loader)
                try:
                    for e in range(max(2, -(-32 // per_epoch))):        # the probe of `auto` (16 eager steps + 9 replayed ones), page cache, workers up
                        trainer.train_epoch(e, loader)
                    n_ep = max(1, steps // per_epoch)
                    count = {"e": 100}

                    def epoch_step():
                        count["e"] += 1
                        return trainer.train_epoch(count["e"], loader)
                    before = trainer.graph_steps
                    el, ep = timed_region(n_ep, epoch_step)
                    n = n_ep * per_epoch
                    leg[name] = {"ms_per_step": round(1e3 * el / n, 4), "value": round(B * n / el, 2), "steps": n,
                                 "host_ms_per_step": round(enqueue["ms_per_step"] / per_epoch, 4),
                                 "hip_graph_auto": trainer.graph_probe_result.get(True), "graph_replayed_steps": trainer.graph_steps - before,
                                 "feed": type(loader).__name__}
                finally:
                    if hasattr(loader, "close"):
                        loader.close()
                    del loader, trainer
                    gc.collect()
                    torch.cuda.empty_cache()
            out[f"batch_{B}"] = leg
    finally:
        shutil.rmtree(tree["path"], ignore_errors=True)
    return out


def live_kernel_sum(batch, mode="graph", steps=40, timeout_s=300):
    """Summed kernel time of ONE step of the default operating point, measured by this run: a rocprofv3 --kernel-trace child over
    tools/shipped_step.py (the resident step, eager or replayed); the trace is cut at the projection's first kernel, the median
    steady-state step gives `kernel_sum_ms` (sum of kernel durations) and `step_wall_ms` (start of a step to start of the next)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_ksum_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "k", "--", sys.executable,
               os.path.join(ROOT, "tools", "shipped_step.py"), str(batch), mode, str(steps)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        rows = sorted(csv.DictReader(open(files[0])), key=lambda r_: int(r_["Start_Timestamp"]))
        return kernel_sum_of_trace(rows, steps)
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_sum_of_trace(rows, steps):
    """Median over the last `steps` - 1 steps of a kernel trace (rows sorted by start time; a step starts at `k_fill_words` followed by
    `k_project_scatter`, i.e. at the projection): (sum of kernel durations, start-to-start wall time, kernels per step)."""
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_project_scatter")]
    if len(starts) < 4:
        return None
    starts = starts[-min(len(starts), steps):]
    busy, wall, count = [], [], []
    for a, b in zip(starts[:-1], starts[1:]):
        busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]))
        wall.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
        count.append(b - a)
    return {"kernel_sum_ms": round(float(np.median(busy)) * 1e-6, 4), "step_wall_ms": round(float(np.median(wall)) * 1e-6, 4),
            "kernels_per_step": int(np.median(count)), "steps_in_trace": len(busy)}


def ddp_rank_leg(args, device, host_batches, batches, timed_region, enqueue, steps=20):
    """What DistributedDataParallel adds to ONE rank, measured without a second GPU: a process group of one rank over RCCL
    (backend "nccl"), the model wrapped exactly as `Trainer._wrap_ddp` wraps it on an 8-GPU node (trunk cut into three autograd
    Functions, 5 MB buckets as gradient views, one all-reduce launch per bucket -- RCCL's single-rank kernels), against the same steps
    unwrapped, fp32 and bf16.  The step time tells what the reducer costs the GPU side, `host_enqueue_ms_per_step` whether a DDP rank's
    HOST keeps up with its GPU (a rank whose enqueue time exceeds its GPU step is host-bound before the network is even involved)."""
    import argparse, gc
    from delora_amd.deploy.trainer import Trainer
    from delora_amd.data.dataset import ListDataset
    from delora_amd.models import ring_conv
    own_group = not torch.distributed.is_initialized()
    if own_group:
        torch.distributed.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=device)
    out = {"process_group": "1 rank, backend nccl (RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version()) + ")"}
    segments = ring_conv.TRUNK_SEGMENTS
    try:
        for amp in ("", "bfloat16"):
            leg = {}
            for wrapped in (False, True):
                a2 = argparse.Namespace(**vars(args))
                a2.amp = amp
                cfg = build_config(a2, device)
                ring_conv.TRUNK_SEGMENTS = segments
                torch.manual_seed(1234)
                trainer = Trainer(cfg, dataset=ListDataset([d for b in host_batches for d in b]))
                identity_pretrained_state(trainer.raw_model)
                if wrapped:
                    trainer.model = trainer._wrap_ddp(trainer.raw_model)
                k = {"i": 0}

                def step():
                    k["i"] += 1
                    trainer.optimizer.zero_grad(set_to_none=True)
                    return trainer.step(preprocessed_dicts=[dict(d) for d in batches[k["i"] % len(batches)]], epoch_losses=trainer.new_epoch_losses())[0]
                for _ in range(len(batches) + 2):
                    step()
                el, ep = timed_region(steps, step)
                leg["ddp" if wrapped else "plain"] = {"ms_per_step": round(1e3 * el / steps, 3), "host_enqueue_ms_per_step": round(enqueue["ms_per_step"], 3),
                                                     "final_loss": float(ep["loss_epoch"])}
                del trainer
                gc.collect()
                torch.cuda.empty_cache()
            leg["ddp_host_bound"] = bool(leg["ddp"]["host_enqueue_ms_per_step"] > leg["plain"]["ms_per_step"])
            leg["ddp_step_overhead_ms"] = round(leg["ddp"]["ms_per_step"] - leg["plain"]["ms_per_step"], 3)
            out["fp32" if not amp else "bf16"] = leg
    finally:
        ring_conv.TRUNK_SEGMENTS = segments
        if own_group:
            torch.distributed.destroy_process_group()
    out["note"] = ("ddp_host_bound = the wrapped rank's host needs longer to enqueue a step than the unwrapped step takes on the GPU; the all-reduce "
                   "itself moves no data here (one rank) -- its xGMI time is arithmetic in DESIGN.md, not a measurement")
    return out


def conv_table(args, device, reps=10):
    """The trunk's convolution kernels on the layer shapes of this workload, one launch each: HIP-event time, direct-equivalent
    TFLOP/s (2*9*C*K flop per output pixel) and, for the matrix-core roofline, the multiply-adds actually issued (Winograd
    F(2x2,3x3) issues 16 per 2x2 outputs and tap instead of 36) against the 157.3 TFLOP/s fp32 MFMA peak."""
    from delora_amd.models import ring_conv as rc
    B, H, W = args.batch, args.height, args.width // 4
    rows = []

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    tot_ms, tot_flop, tot_mfma = 0.0, 0.0, 0.0
    for name, h, w, c in (("layer1 64ch", H, W, 64), ("layer2 128ch", H, W // 2, 128), ("layer3 256ch", H, W // 4, 256),
                          ("layer4 512ch", H // 2, W // 8, 512)):
        x = torch.randn((B, h, w, c), device=device)
        wt = (torch.randn((c, c, 3, 3), device=device) * 0.02).contiguous(memory_format=torch.channels_last)
        g = torch.randn((B, h, w, c), device=device)
        flop = 2.0 * B * h * w * c * c * 9
        uf, ub = rc.wino_weights(wt)
        legs = (("forward (Winograd)", lambda: rc.wino_conv(x, uf, c, act=1, epilogue=rc.EPI_ACT), 1 / 2.25),
                ("input gradient (Winograd)", lambda: rc.wino_conv(g, ub, c, act=1, epilogue=rc.EPI_DACT, dsrc=x), 1 / 2.25),
                ("weight gradient (direct)", lambda: rc.wgrad_nhwc(x, g, 3), 1.0))
        for leg, fn, mfma_share in legs:
            ms = timed(fn)
            rows.append({"layer": name, "pass": leg, "ms": round(ms, 4), "TFLOPs_direct_equivalent": round(flop / ms * 1e-9, 1),
                         "TFLOPs_mfma_issued": round(flop * mfma_share / ms * 1e-9, 1),
                         "frac_mfma_peak": round(flop * mfma_share / ms * 1e-9 / 157.3, 3)})
            tot_ms += ms; tot_flop += flop; tot_mfma += flop * mfma_share
    return {"bound": "mfma", "peak": 157.3, "unit": "TFLOP/s", "achieved": round(tot_mfma / tot_ms * 1e-9, 1),
            "frac": round(tot_mfma / tot_ms * 1e-9 / 157.3, 3), "achieved_direct_equivalent": round(tot_flop / tot_ms * 1e-9, 1),
            "note": "stride-1 3x3 layers of the trunk (85 % of the network's multiplications), one launch per layer shape and pass",
            "layers": rows}


def conv_roofline(rows, args, step_ms, prof_steps=None):
    """The `roofline` object of the JSON line from the launch profile of the timed steps (delora_amd/_lib.py: profile_end): the
    kernel family with the largest share of the step, its algorithmic flop / its summed kernel time against the dense MFMA peak
    of its arithmetic type, one row per layer shape with the HBM traffic of the committed PMC passes next to the compulsory
    bytes.  Also returns the whole profile (every instrumented kernel) as per-step rows."""
    nsteps = prof_steps or args.steps
    fam = {}
    for r in rows:
        k = r["name"].split(" ")[0]
        f = fam.setdefault(k, {"ms": 0.0, "flop": 0.0, "launches": 0})
        f["ms"] += r["ms"]; f["flop"] += r["flop"]; f["launches"] += r["launches"]
    dom = max(fam, key=lambda k: fam[k]["ms"])
    half = dom in ("k_convh", "k_wgradh")
    peak = MFMA_BF16_PEAK_TFLOPS if half else MFMA_F32_PEAK_TFLOPS
    traffic = pmc_layer_traffic(dom)
    layers = []
    for r in sorted((r for r in rows if r["name"].startswith(dom + " ")), key=lambda r: -r["ms"]):
        per = r["ms"] / r["launches"]
        t = traffic.get(r["name"])
        layers.append({"launch": r["name"], "launches_per_step": round(r["launches"] / nsteps, 2), "ms_per_launch": round(per, 5),
                       "TFLOPs": round(r["flop"] / r["ms"] * 1e-9, 1), "frac": round(r["flop"] / r["ms"] * 1e-9 / peak, 4),
                       "compulsory_MB_per_launch": round(r["bytes"] / r["launches"] / 1e6, 2),
                       "traffic_MB_per_launch": None if t is None else round(t / 1e6, 2)})
    d = fam[dom]
    tf = d["flop"] / d["ms"] * 1e-9
    known = [(l["traffic_MB_per_launch"], l["launches_per_step"]) for l in layers if l["traffic_MB_per_launch"] is not None]
    what = {"k_wino_conv": "fused Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32: the stride-1 3x3 layers of the pose CNN, forward and input gradient",
            "k_wino_wgrad": "Winograd-domain weight gradient of the stride-1 3x3 layers on v_mfma_f32_32x32x2_f32",
            "k_wgrad_f32": "direct weight gradient on v_mfma_f32_32x32x2_f32", "k_conv_f32": "direct implicit-GEMM convolution on v_mfma_f32_32x32x2_f32",
            "k_convh": "half-precision implicit-GEMM convolution (LDS DMA) on v_mfma_f32_32x32x16_bf16",
            "k_wgradh": "half-precision weight gradient (transposing LDS reads) on v_mfma_f32_32x32x16_bf16"}[dom]
    roof = {"kernel": f"{dom} ({what}; the largest share of the step)", "bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(tf / peak, 4),
            "traffic": None if not known else int(1e6 * sum(t * n for t, n in known) / sum(n for _, n in known)),
            "launches_per_step": round(d["launches"] / nsteps, 2), "ms_per_launch": round(d["ms"] / d["launches"], 5),
            "ms_per_step": round(d["ms"] / nsteps, 4), "share_of_step": round(d["ms"] / nsteps / step_ms, 3),
            "algorithmic_flop_per_launch": round(d["flop"] / d["launches"]), "layers": layers,
            "note": "achieved = flop the algorithm issues on the matrix cores (Winograd: 16 multiply-adds per 2x2 output tile and (c,k) pair = a direct "
                    "convolution's / 2.25; direct kernels: 36) / kernel time; begin/end timestamps on HIP events attached to every launch "
                    "(hipExtLaunchKernelGGL) of the K timed steps, summed per layer shape; peak = dense MFMA peak of the arithmetic type "
                    "(MI355X_MICROARCH.md: 157.3 TFLOP/s fp32, 2500 bf16/fp16); traffic = bytes beyond the L2 per launch (launch-weighted mean of the "
                    "layer rows) from rocprofv3 --pmc FETCH_SIZE (x2: gfx950 correction) and --pmc WRITE_SIZE passes, see traffic_source",
            "traffic_source": LIVE_PMC["source"]}
    if dom == "k_wino_conv":
        roof["achieved_direct_equivalent"] = round(2.25 * tf, 1)
    prof = [{"launch": r["name"], "launches_per_step": round(r["launches"] / nsteps, 2), "ms_per_step": round(r["ms"] / nsteps, 4),
             "TFLOPs": round(r["flop"] / r["ms"] * 1e-9, 1), "compulsory_GB_s": round(r["bytes"] / r["ms"] * 1e-6, 0)}
            for r in sorted(rows, key=lambda r: -r["ms"])]
    return roof, {"families_ms_per_step": {k: round(v["ms"] / nsteps, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}, "rows": prof}


LIVE_PMC = {"rows": None, "source": None}      # filled once by live_pmc_traffic()


def live_pmc_traffic(mode, timeout_s=240):
    """HBM-side bytes per convolution launch MEASURED BY THIS RUN: two rocprofv3 child processes (`--kernel-trace --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE`, separate passes as MI355X_MICROARCH.md prescribes; no other trace domain) over tools/conv_layers.py,
    which launches every (kernel, pass, layer shape) of the trunk once at the bench's batch size; the dispatches are mapped back to
    the launch-profile row names by tools/conv_layers_pmc.py (x2 read correction of gfx950).  Returns {row: bytes per launch} or
    None (no rocprofv3, nested under a profiler, child failure) -- the caller then falls back to the committed bundle."""
    import glob, importlib.util, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    order = os.path.join(tmp, "order.json")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "-o", "c", "--",
                   sys.executable, os.path.join(ROOT, "tools", "conv_layers.py"), mode, order]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
        csvs = [glob.glob(os.path.join(tmp, c, "**", "*counter_collection.csv"), recursive=True) for c in ("FETCH_SIZE", "WRITE_SIZE")]
        if not (csvs[0] and csvs[1] and os.path.exists(order)):
            return None
        spec = importlib.util.spec_from_file_location("conv_layers_pmc", os.path.join(ROOT, "tools", "conv_layers_pmc.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        rows = mod.traffic_rows(json.load(open(order)), csvs[0][0], csvs[1][0])
        return {name: int(v["hbm_bytes_per_launch"]) for name, v in rows.items() if v["hbm_bytes_per_launch"] > 0}
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_layer_traffic(family):
    """HBM bytes per launch by profile-row name: measured by this run (live_pmc_traffic) when available, else from the newest
    committed PMC bundle (profiles/r*_conv_hbm_pmc.json: {"launches": {row name: {"hbm_bytes_per_launch": ...}}}); {} when absent."""
    import glob
    if LIVE_PMC["rows"]:
        return {k: v for k, v in LIVE_PMC["rows"].items() if k.startswith(family + " ")}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_hbm_pmc.json")))
    out = {}
    for f in files[-1:]:
        try:
            for name, v in json.load(open(f)).get("launches", {}).items():
                if name.startswith(family + " "):
                    out[name] = int(v["hbm_bytes_per_launch"])
            LIVE_PMC["source"] = "committed " + os.path.relpath(f, ROOT) + " (rocprofv3 --pmc passes of tools/prof_round.sh; not measured by this run)"
        except (ValueError, KeyError, TypeError):
            pass
    return out


def pmc_conv_traffic():
    """Mean HBM bytes per k_wino_conv launch over the four layer shapes (forward + input gradient) from the committed PMC
    passes of tools/conv_harness (profiles/*_conv_hbm_pmc.json); None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_hbm_pmc.json")))
    if not files:
        return None
    try:
        return int(json.load(open(files[-1]))["k_wino_conv"]["hbm_bytes_per_launch_mean"])
    except (KeyError, ValueError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC profile of this workload (profiles/*_geometry_pmc.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction); None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_geometry_pmc.json")))
    if not files:
        return None
    try:
        return int(json.load(open(files[-1]))["kernels"][kernel]["hbm_bytes_per_launch"])
    except (KeyError, ValueError):
        return None


def kernel_table(trainer, batch, reps):
    """Per-launch time of each geometry entry point on the bench batch: `reps` back-to-back launches between two HIP
    events (launch gaps included), with the algorithmic bytes of DESIGN.md; and the loss kernel alone behind a flush of
    the infinity cache (cold)."""
    from delora_amd import geometry as G
    cfg = trainer.config
    sensor = trainer.img_projection.sensor("kitti")
    B, HW = len(batch), sensor.H * sensor.W
    prepared = trainer.geo.prepare(batch, sensor, trainer._normal_params("kitti"))
    img, nrm = prepared["images"], prepared["normals"]
    tgt_pk, tgt_n_pk = prepared["packed"][:, 0], prepared["normals_packed"][:, 0]
    # a realistic pose (small residual motion) for the correspondence / loss kernels
    T_small = torch.eye(4, device=trainer.device).repeat(B, 1, 1)
    T_small[:, 0, 3] = 0.4
    pts = torch.cat([torch.cat((s["scan_1"][0], s["scan_2"][0]), dim=1) for s in batch], dim=1).contiguous()
    lengths = [n for s in batch for n in (s["scan_1"].shape[2], s["scan_2"].shape[2])]
    offs = trainer.geo._offsets_for(lengths, pts.device)
    n_pts = int(sum(lengths))
    nn, _, match = G.nn_correspond(img[:, 1], nrm[:, 1], tgt_pk, tgt_n_pk, T_small, sensor)
    flags = G.loss_flags(cfg)
    terms, counts = G.icp_loss(T_small, img[:, 1], nrm[:, 1], match, nn, flags)
    M = int((nn >= 0).sum())
    K = int(counts[:, 0].sum())
    kept = int((prepared["pix2pt"] >= 0).sum())

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    rows = []

    def row(name, ms, nbytes, bound, note):
        rows.append({"kernel": name, "ms": round(ms, 5), "algorithmic_MB": round(nbytes / 1e6, 3),
                     "GB_s": round(nbytes / ms / 1e6, 1), "frac_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                     "bound": bound, "note": note})

    row("dl_project", timed(lambda: G.project(pts, offs, max(lengths), sensor, want_kept=False)), 12 * n_pts + 36 * 2 * B * HW, "hbm",
        f"{2 * B} scans, {n_pts} points -> {2 * B}x{sensor.H}x{sensor.W}; 12 B/point + 36 B/pixel (planar + packed image, map); points in the "
        f"bench's order (--point-order, default raster: ring after ring, as the reference's stored lists, src/preprocessing/preprocesser.py:60-67)")
    # the same points randomly permuted inside every scan (the input order of rounds 1-4): every vote touches another line of the key plane
    g_perm = torch.Generator(device="cpu").manual_seed(11)
    perm = torch.cat([o + torch.randperm(n, generator=g_perm) for o, n in zip(np.cumsum([0] + lengths[:-1]).tolist(), lengths)]).to(pts.device)
    pts_shuffled = pts[:, perm].contiguous()
    row("dl_project/shuffled-points", timed(lambda: G.project(pts_shuffled, offs, max(lengths), sensor, want_kept=False)), 12 * n_pts + 36 * 2 * B * HW,
        "hbm", "the same points randomly permuted inside every scan (rounds 1-4 fed the bench this order): the worst case of the vote's atomics")
    del perm, pts_shuffled
    row("dl_normals", timed(lambda: G.normals(prepared["stacked"].view(2 * B, 4, sensor.H, sensor.W), want_packed=True)), 40 * 2 * B * HW, "valu",
        "7x11 stencil + 3x3 eigen solve; 40 B/pixel (read xyz, write planar + packed normals)")
    row("dl_nn_correspond", timed(lambda: G.nn_correspond(img[:, 1], nrm[:, 1], tgt_pk, tgt_n_pk, T_small, sensor)), 28 * B * HW + 12 * B * HW, "l2+valu",
        f"{M} queries, residual motion 0.4 m")
    # the random-pose regime (an untrained network): every query falls back to the tile walk over the whole image
    g = torch.Generator().manual_seed(5)
    q = torch.randn((B, 4), generator=g)
    from delora_amd.models.model_parts import GeometryHandler
    T_rand = GeometryHandler.get_transformation_matrix_quaternion(torch.randn((B, 3), generator=g), q, torch.device("cpu")).to(trainer.device)
    row("dl_nn_correspond/random-pose", timed(lambda: G.nn_correspond(img[:, 1], nrm[:, 1], tgt_pk, tgt_n_pk, T_rand, sensor)),
        28 * B * HW + 12 * B * HW, "l2+valu", "random rotations + 1 m translations (worst case of the exact search)")
    # a network in mid-training: a few degrees of roll / pitch error and a few decimetres (most queries keep a bound, but a large one:
    # the regime of the windowed packets of pass B)
    q_tilt = torch.cat([0.03 * torch.randn((B, 3), generator=g), torch.ones((B, 1))], dim=1)
    T_tilt = GeometryHandler.get_transformation_matrix_quaternion(0.2 * torch.randn((B, 3), generator=g), q_tilt, torch.device("cpu")).to(trainer.device)
    row("dl_nn_correspond/tilted-pose", timed(lambda: G.nn_correspond(img[:, 1], nrm[:, 1], tgt_pk, tgt_n_pk, T_tilt, sensor)),
        28 * B * HW + 12 * B * HW, "l2+valu", "rotations of a few degrees about random axes + 0.2 m translations (a network in mid-training)")
    # the pose the network itself produces at this point of the run (what the search of the timed steps sees): a network that
    # trains on unrelated random scenes does not converge to the true motion, so the residual the search faces is larger
    # than the 0.4 m of the row above
    with torch.no_grad():
        t_net, q_net = trainer.raw_model(prepared["stacked"])
        T_net = GeometryHandler.get_transformation_matrix_quaternion(t_net.float(), q_net.float(), trainer.device)
        qn = q_net.float() / q_net.float().norm(dim=1, keepdim=True)
        pose = {"translation_m_mean": round(float(t_net.float().norm(dim=1).mean()), 4),
                "rotation_deg_mean": round(float(torch.rad2deg(2 * torch.acos(qn[:, 3].abs().clamp(max=1.0))).mean()), 4)}
    row("dl_nn_correspond/network-pose", timed(lambda: G.nn_correspond(img[:, 1], nrm[:, 1], tgt_pk, tgt_n_pk, T_net, sensor)),
        28 * B * HW + 12 * B * HW, "l2+valu", f"the pose the network outputs after the timed steps: |t| = {pose['translation_m_mean']} m, "
        f"rotation {pose['rotation_deg_mean']} deg (mean over the batch); this is the regime of the search inside the timed steps")
    row("dl_icp_loss_fwd", timed(lambda: G.icp_loss(T_small, img[:, 1], nrm[:, 1], match, nn, flags)), 52 * M, "hbm",
        f"both launches (stream + reduce); {M} source points with a correspondence, {K} pairs; 52 B/point")
    # cold: the loss kernel alone with none of its operands in a cache.  Two ways of getting there, and they differ by 1.6x:
    #   clean  1 GiB of unrelated writes, then 1 GiB of unrelated READS (another buffer): the read pass pushes the dirty lines out, the
    #          256 MiB infinity cache ends up full of clean unrelated lines -- the kernel's reads are the only HBM traffic while it runs
    #   dirty  the 1 GiB of writes alone (rounds 2-3 measured this): the cache is full of somebody else's DIRTY lines, every line the
    #          kernel brings in forces a write-back, and the kernel is charged for 54 MB of writes it did not ask for
    # (tools/loss_cold.py -> profiles/r04_loss_cold.txt: a read-only flush gives the same time as write-then-read, i.e. reads do
    # allocate in the infinity cache and the clean flush is a real one)
    flush = torch.empty((256 * 1024 * 1024,), dtype=torch.float32, device=trainer.device)
    flush2 = torch.zeros((256 * 1024 * 1024,), dtype=torch.float32, device=trainer.device)
    sink = torch.zeros((1,), device=trainer.device)

    def cold_run(with_reads):
        timers = G.LossTimers(reserve=16)
        G.LOSS_TIMER_FACTORY = timers.new
        try:
            for i in range(12):
                flush.fill_(float(i))
                if with_reads:
                    sink.add_(flush2.sum())
                G.icp_loss(T_small, img[:, 1], nrm[:, 1], match, nn, flags)
            torch.cuda.synchronize()
        finally:
            G.LOSS_TIMER_FACTORY = None
        ms = timers.elapsed_ms()[2:]
        timers.close()
        return ms

    cold, dirty = cold_run(True), cold_run(False)
    del flush, flush2
    return rows, {"M": M, "K": K, "kept": kept, "loss_cold_ms": float(np.mean(cold)), "loss_cold_min_ms": float(np.min(cold)),
                  "loss_cold_dirty_ms": float(np.mean(dirty)), "network_pose": pose}


def _cpu_step(orc, model, opt, cfg, lists, images):
    from delora_amd.models.model_parts import GeometryHandler
    opt.zero_grad()
    t, q = model(images[0], images[1])
    T = GeometryHandler.get_transformation_matrix_quaternion(t, q, torch.device("cpu"))
    out, _ = orc.step_losses([lists], T, lambda_po2pl=cfg["lambda_po2pl"], normal_loss=cfg["normal_loss"])
    out["loss_pc"].sum().backward()
    opt.step()


def cpu_baseline(args, cfg):
    """The training step on the host with the CPU oracle (torch CPU ops + scipy cKDTree) + the CNN on CPU threads, B=1 (the
    reference's own batch size), same 64x2048 synthetic generator, bounded sample.
      cpu_baseline                 the reference's step (src/deploy/trainer.py:43-91, deployer.py:237-375): point and normal
                                   lists come from disk (computed offline, untimed here), the step re-projects them, runs the
                                   network, KD-tree correspondences, losses, backward, Adam
      cpu_baseline_online_normals  the GPU workload itself: normals computed inside the step from the projected images."""
    from oracle import delora_oracle as orc
    from delora_amd.models.model import OdometryModel
    from delora_amd.data import synthetic
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    cores = torch.get_num_threads()
    ccfg = dict(cfg)
    ccfg["device"] = torch.device("cpu")
    ccfg.pop("cnn_impl", None)
    torch.manual_seed(0)
    model = OdometryModel(ccfg)
    opt = torch.optim.Adam(model.parameters(), lr=cfg["learning_rate"])
    sensor = orc.Sensor(args.height, args.width, cfg["kitti"]["vertical_field_of_view"], cfg["horizontal_field_of_view"])
    pre = orc.Sensor(args.height, 2250, cfg["kitti"]["vertical_field_of_view"], cfg["horizontal_field_of_view"])
    side = cfg["kitti"]["neighborhood_side_length"]
    kw = dict(side=side, epsilon_range=cfg["epsilon_range"], min_neighbors=cfg["min_num_points_in_neighborhood_to_determine_point_class"])
    t_stored, t_online = 0.0, 0.0
    online_pairs = 1
    for k in range(args.cpu_pairs):
        s1, s2, _ = synthetic.make_pair(2000 + k, rings=args.height, azimuth_steps=2250)
        # offline preprocessing of the reference (bin/preprocess_data.py -> scans/normals lists on disk), untimed
        stored = {}
        for name, s in (("1", s1), ("2", s2)):
            img, _, _, _, _ = orc.project_to_img(torch.from_numpy(s).view(1, 3, -1), pre)
            nrm, has, pts = orc.compute_normal_vectors(img.clone(), pre, **kw)
            stored["scan_" + name] = pts.t().contiguous().view(1, 3, -1)
            stored["normal_list_" + name] = nrm.t().contiguous().view(1, 3, -1)
        t0 = time.perf_counter()
        img1, img2, lists = orc.filter_to_projected(stored, sensor)                 # deployer.py:252-267
        _cpu_step(orc, model, opt, cfg, lists, [img1.unsqueeze(0), img2.unsqueeze(0)])
        t_stored += time.perf_counter() - t0
        if k < online_pairs:
            t0 = time.perf_counter()
            lists, images = {}, []
            for name, s in (("1", s1), ("2", s2)):
                img, _, _, _, _ = orc.project_to_img(torch.from_numpy(s).view(1, 3, -1), sensor)
                nrm, has, pts = orc.compute_normal_vectors(img.clone(), sensor, **kw)
                lists["scan_" + name] = pts.t().contiguous().view(1, 3, -1)
                lists["normal_list_" + name] = nrm.t().contiguous().view(1, 3, -1)
                images.append(img)
            _cpu_step(orc, model, opt, cfg, lists, images)
            t_online += time.perf_counter() - t0
    base = {"unit": "scan-pairs/s", "cores": cores, "kind": "port"}
    stored_leg = dict(base, value=round(args.cpu_pairs / t_stored, 4), s_per_pair=round(t_stored / args.cpu_pairs, 3),
                      sample=f"{args.cpu_pairs} pairs, B=1, {args.height}x{args.width}, stored point+normal lists (offline preprocessing untimed, as the reference), "
                             f"oracle re-projection + CNN fwd/bwd + cKDTree + losses + Adam on {cores} threads")
    online_leg = dict(base, value=round(online_pairs / t_online, 4), s_per_pair=round(t_online / online_pairs, 3),
                      sample=f"{online_pairs} pair, B=1, {args.height}x{args.width}, normals computed inside the step (the GPU workload) on {cores} threads")
    return stored_leg, online_leg


def launch_plan(gpus, environ, device_count, argv, free_port=None):
    """What a `bench.py --gpus N` invocation has to do before it may measure anything (pure: tested on the CPU).

    Returns ("run", None) when this process is a rank of a correctly sized job (or N = 1), ("spawn", command) when it was
    started as ONE plain process for N > 1 GPUs -- it then re-executes itself under torch.distributed.run with N ranks, so a
    `python bench.py --gpus 8` can never print a single-rank number labelled as an 8-GPU run -- and raises SystemExit when the
    request cannot be honoured (WORLD_SIZE disagrees with --gpus, fewer visible GPUs than ranks without the share-GPU test hook)."""
    share = environ.get("DELORA_BENCH_SHARE_GPU") == "1"
    if gpus < 1:
        raise SystemExit(f"--gpus {gpus}: need at least one GPU")
    if "WORLD_SIZE" in environ and "RANK" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: launch one rank per GPU (torch.distributed.run --nproc-per-node {gpus})")
    if device_count < gpus and not share:
        raise SystemExit(f"--gpus {gpus} but only {device_count} GPU(s) visible: refusing to run several ranks on one device "
                         f"(DELORA_BENCH_SHARE_GPU=1 is the test hook that allows it)")
    if gpus == 1 or ("WORLD_SIZE" in environ and "RANK" in environ):
        return "run", None
    port = environ.get("MASTER_PORT") or str(free_port() if free_port else 29500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", port] + list(argv)
    return "spawn", cmd


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    args = parse()
    action, cmd = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), [os.path.abspath(__file__)] + sys.argv[1:], _free_port)
    if action == "spawn":
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        print(f"bench.py: --gpus {args.gpus} started as one process: re-launching as {args.gpus} ranks ({' '.join(cmd[1:8])} ...)", file=sys.stderr)
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    t_start = time.perf_counter()
    # The contract is ONE JSON line on stdout.  Libraries print there too -- the dataset's and the trainer's progress lines, and RCCL's
    # version banner (a C-level printf at communicator set-up that lands BEHIND the JSON line once Python's buffer is flushed) -- so for
    # the whole run file descriptor 1 is pointed at stderr, and the result goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def trace(what):                                    # DELORA_BENCH_TRACE=1: milestones of every rank on stderr (where does a run spend / lose its time)
        if os.environ.get("DELORA_BENCH_TRACE"):
            print(f"[bench rank {rank}/{world} +{time.perf_counter() - t_start:7.2f}s] {what}", file=sys.stderr, flush=True)
    # test hooks (never set in production): run all ranks on one GPU over gloo to exercise the N>1 code path on a 1-GPU box
    if os.environ.get("DELORA_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("DELORA_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend=backend)
    trace("process group up")
    from delora_amd.deploy.trainer import Trainer
    from delora_amd.data.dataset import ListDataset
    from delora_amd.data.feed import DevicePrefetcher
    if args.miopen_benchmark:
        torch.backends.cudnn.benchmark = True
    cfg = build_config(args, device)
    if args.trunk_segments:
        from delora_amd.models import ring_conv
        ring_conv.TRUNK_SEGMENTS = args.trunk_segments
    if args.alloc_skew >= 0:
        from delora_amd.models import ring_conv
        ring_conv.ALLOC_SKEW = args.alloc_skew
    torch.manual_seed(1234)
    host_batches = pin_batches(derived_batches(make_batch(args, rank), max(1, args.rotate), rank, shuffled=args.point_order == "shuffled"))
    batches = [to_device(b, device) for b in host_batches]
    trace("batches generated")
    trainer = Trainer(cfg, dataset=ListDataset([d for b in host_batches for d in b]))
    identity_pretrained_state(trainer.raw_model)
    trace("trainer built")
    counter = {"i": 0}
    # N > 1: the gradient all-reduce is the path's only exchange; its timeline (per bucket: handed over / completed; the part that is not
    # hidden behind the backward pass) is recorded in a pass of its own AFTER the timed steps (deploy/ddp_trace.py).  The hook is
    # registered now -- DistributedDataParallel takes one hook per model -- and does what the default one does (divide, all-reduce).
    timeline = None
    if world > 1:
        from delora_amd.deploy.ddp_trace import DdpTimeline
        timeline = DdpTimeline()
        timeline.enabled = False
        timeline.attach(trainer.model, last_grad_param=trainer.raw_model.resnet.conv1.weight)

    def run_step(batch=None):
        if batch is None:
            batch = batches[counter["i"] % len(batches)]
            counter["i"] += 1
        trainer.optimizer.zero_grad(set_to_none=True)
        ep = trainer.new_epoch_losses()
        from delora_amd.deploy.step_geometry import PackedBatch
        if timeline is not None:
            timeline.begin_step()
        ep, T = trainer.step(preprocessed_dicts=batch if isinstance(batch, PackedBatch) else [dict(s) for s in batch], epoch_losses=ep)
        if timeline is not None:
            timeline.end_step()
        return ep

    # Priming (set-up, not warm-up): one step on each of the distinct ragged batches, so that torch's caching allocator has seen every
    # tensor size of the rotation before the W warm-up steps -- a first visit of a batch costs hipMalloc calls (host stalls, device
    # synchronisation): with W = 5 < 8 batches the first timed steps were such visits (autocast step: 11 ms instead of 5 ms of enqueue)
    for b in batches:
        run_step(b)
        trace("priming step enqueued")
    # As Trainer.train does after its set-up: everything alive now (batches, datasets, module trees) moves to the collector's permanent
    # generation.  A generation-2 pass over that heap stalled the host for ~0.1 s at random steps -- invisible behind a 14.6 ms GPU-bound
    # step, 5 ms per step of a 20-step autocast region (1500 pairs/s read as 750).
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        run_step()
    trace("warm-up enqueued")
    graphed = None
    if args.graph and world == 1:
        from delora_amd.deploy.graph_step import GraphedStep
        graphed = GraphedStep(trainer, batches[0])
        if graphed.captured:
            eager_step = run_step

            def run_step(batch=None):                              # ragged batches through the captured graph's static buffers
                if batch is None:
                    batch = batches[counter["i"] % len(batches)]
                    counter["i"] += 1
                return graphed(batch)[0]
    # in-situ timing of the streaming loss kernel (k_icp_loss) in every timed step: the launch carries a pair of HIP
    # events that receive the kernel's own begin/end timestamps (dl_icp_loss_partial_timed) on the launch stream
    from delora_amd import geometry as G
    timers = G.LossTimers(reserve=args.steps + 8)
    loss_timed = (graphed is None or not graphed.captured) and not args.no_profile      # event-carrying launches cannot be captured into a graph

    enqueue = {}

    def timed_region(steps, step_fn):
        return _timed_region(steps, step_fn)

    def _timed_region(steps, step_fn):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        ep = None
        per_step = []
        for _ in range(steps):
            t1 = time.perf_counter()
            ep = step_fn()
            per_step.append(round(1e3 * (time.perf_counter() - t1), 1))
        if os.environ.get("DELORA_BENCH_STEP_TIMES"):
            print("host ms per step:", per_step, file=sys.stderr)
        enqueue["ms_per_step"] = 1e3 * (time.perf_counter() - t0) / steps      # host time to enqueue the steps (no sync yet)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=device, dtype=torch.float64)
            every = [torch.zeros_like(tt) for _ in range(world)]
            torch.distributed.all_gather(every, tt)                    # per-rank wall time of the region (reported min / max)
            per_rank = [float(t.item()) for t in every]
            enqueue["rank_ms_per_step"] = [round(1e3 * t / steps, 3) for t in per_rank]
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            el = float(tt.item())
        return el, ep

    # in-situ timing of the convolution kernels the same way: every launch of the timed steps carries its own begin/end
    # timestamps (dl_profile_begin / dl_profile_end: one row per kernel, pass and layer shape)
    from delora_amd import _lib
    conv_prof = None
    dominant = "k_convh" if args.amp else "k_wino_conv"          # the family with the largest share of the step (conv_profile below)
    can_profile = (graphed is None or not graphed.captured) and "hip trunk" in cnn_impl_in_use(trainer, args) and not args.no_profile
    # An event-carrying launch (hipExtLaunchKernelGGL with start/stop events) costs the HOST ~0.25 ms, and it keeps the host from
    # running ahead of the GPU.  With the 26 + 1 such launches in EVERY timed step the 14.7 ms fp32 step of a fresh box was host-bound
    # (492 pairs/s against 545 with --no-profile, same box, back to back).  The timed region therefore carries events in every
    # EVENT_EVERY-th step only (dl_profile_pause in between): 5 of the default 20 steps, 130 Winograd launches.  The 5 ms autocast
    # step and a DDP rank have no slack at all: there the K timed steps carry no events and the rooflines come from a second pass.
    # Round 6: the K timed steps of the headline carry NO events in any mode (review: event-carrying launches perturbed exactly the number
    # being reported); the rooflines are read from a second pass over the same steps.
    EVENT_EVERY = 4
    in_timed = False
    evented = {"steps": 0}

    def run_step_sampled():
        on = counter["i"] % EVENT_EVERY == 0
        if can_profile:
            _lib.profile_pause(not on)
        G.LOSS_TIMER_FACTORY = timers.new if (on and loss_timed) else None
        evented["steps"] += int(on)
        return run_step()
    if can_profile:
        _lib.profile_begin(int(args.steps) * 64, dominant)         # the steps time this kernel family only
        _lib.profile_pause(True)
    counter["i"] = 0
    elapsed, ep = timed_region(args.steps, run_step_sampled if in_timed and (can_profile or loss_timed) else run_step)
    host_enqueue_ms = enqueue["ms_per_step"]
    rank_ms = enqueue.get("rank_ms_per_step")
    trace("timed region done")
    roofline_pass = f"every {EVENT_EVERY}th of the K timed steps ({evented['steps']} steps)"
    if not in_timed and (can_profile or loss_timed):
        counter["i"] = 0
        timed_region(args.steps, run_step_sampled)
        roofline_pass = (f"every {EVENT_EVERY}th step of a second pass over the same K steps ({evented['steps']} steps; the timed steps of this "
                         f"mode carry no events: see bench.py)")
    G.LOSS_TIMER_FACTORY = None
    if can_profile:
        conv_prof, untimed = _lib.profile_end()
        assert untimed == 0, f"{untimed} launches were not timed: raise the profile capacity"
    ddp_timeline = None
    if timeline is not None:
        timeline.enabled = True
        n_tl = max(4, min(16, args.steps))
        timed_region(n_tl, run_step)
        timeline.enabled = False
        mine = timeline.summary(last=n_tl)
        mine["rank"] = rank
        mine["feed_wait_ms_per_step"] = 0.0          # the timed steps read HBM-resident batches; the disk-fed legs (N = 1) measure the feed
        every = [None] * world
        torch.distributed.all_gather_object(every, mine)
        ex = [r.get("exposed_allreduce_ms") for r in every if r.get("exposed_allreduce_ms") is not None]
        ddp_timeline = {"steps": n_tl, "exposed_allreduce_ms": ({"min": min(ex), "max": max(ex), "mean": round(sum(ex) / len(ex), 3)} if ex else None),
                        "per_rank": every,
                        "note": "a pass of its own after the timed steps: per bucket of the gradient all-reduce the time it was handed over and the "
                                "time it completed (ms from the start of the step, HIP events), the end of the backward computation (the stem's first "
                                "convolution receives its gradient last), and exposed = completion of the last bucket - end of the backward computation"}
        trace("DDP timeline recorded")
    final_loss = float(ep["loss_epoch"])
    pairs = world * args.batch * args.steps
    ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    result = {
        "metric": "training scan-pairs/sec, KITTI 64x2048 range images", "value": round(pairs / elapsed, 3), "unit": "scan-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if not args.amp else args.amp,
        "data": "synthetic",
        "config": {"workload": f"KITTI-shaped {args.height}x{args.width}, batch={args.batch} pairs/GPU, raw scans of ~141k points (ragged, {args.point_order} order), "
                               f"steps rotate over {len(batches)} distinct HBM-resident batches, online normals, ResNet pose CNN (11.9M params, "
                               f"identity-pretrained state), Adam; BASELINE configs[1]",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}", "cnn": "fp32" if not args.amp else "autocast " + args.amp,
                   "cnn_impl": cnn_impl_in_use(trainer, args),
                   "channels_last": bool(args.channels_last), "hip_graph": bool(graphed is not None and graphed.captured),
                   "distinct_batches": len(batches), "priming_steps": len(batches)},
        "final_loss": final_loss, "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
        "rccl_ranks": ranks_seen, "collective_backend": (backend if world > 1 else None),
        "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None),
        "visible_gpus": torch.cuda.device_count(),
        "rank_ms_per_step": ({"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms} if rank_ms else None),
        "ddp_timeline": ddp_timeline,
    }
    if rank == 0:
        rows, counts = kernel_table(trainer, batches[0], args.kernel_reps)
        if timers.used == 0 and world == 1:                    # graph mode: measure the same launch right after the timed region
            # (one process only: under DDP a training step is a collective -- rank 0 stepping alone would wait for the others forever)
            G.LOSS_TIMER_FACTORY = timers.new
            for _ in range(5):
                trainer.optimizer.zero_grad(set_to_none=True)
                trainer.step(preprocessed_dicts=[dict(s) for s in batches[0]], epoch_losses=trainer.new_epoch_losses())
            torch.cuda.synchronize()
            G.LOSS_TIMER_FACTORY = None
        alg = next(r for r in rows if r["kernel"] == "dl_icp_loss_fwd")
        loss_ms = float(np.mean(timers.elapsed_ms())) if timers.used else float(alg["ms"])     # (no in-step timer: the back-to-back figure)
        timers.close()
        live_bytes = 52 * counts["M"]
        warm = live_bytes / loss_ms / 1e6
        cold = live_bytes / counts["loss_cold_ms"] / 1e6
        dirty = live_bytes / counts["loss_cold_dirty_ms"] / 1e6
        result["roofline_loss"] = {
            "kernel": "k_icp_loss (dl_icp_loss_partial: fused transform + residuals + reduction, 13 planes streamed)",
            "bound": "hbm", "regime_in_step": "infinity-cache (the search kernel has just written/read the 54 MB of operands)",
            # the headline of this object is the COLD figure: operands from HBM (what "HBM roofline" means), caches full of CLEAN unrelated
            # lines; the in-step launch finds its operands in the 256 MiB Infinity Cache and is reported next to it, and so is the launch
            # behind a cache full of DIRTY unrelated lines (rounds 2-3's "cold"), which pays for their write-back
            "achieved": round(cold, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(cold / HBM_PEAK_GBS, 4),
            "frac_warm": round(warm / HBM_PEAK_GBS, 4), "frac_cold": round(cold / HBM_PEAK_GBS, 4),
            "frac_cold_behind_dirty_cache": round(dirty / HBM_PEAK_GBS, 4),
            "achieved_cold": round(cold, 1), "achieved_in_step": round(warm, 1), "traffic": pmc_traffic("k_icp_loss"),
            "ms_per_launch_in_step": round(loss_ms, 5), "ms_per_launch_cold": round(counts["loss_cold_ms"], 5),
            "ms_per_launch_cold_min": round(counts["loss_cold_min_ms"], 5),
            "ms_per_launch_cold_behind_dirty_cache": round(counts["loss_cold_dirty_ms"], 5), "ms_per_launch_back_to_back": alg["ms"],
            "algorithmic_bytes": live_bytes,
            "note": "frac = frac_cold: the launch after 1 GiB of unrelated writes followed by 1 GiB of unrelated reads (no operand in any "
                    "cache, the infinity cache full of clean lines: the kernel's own reads are the only HBM traffic); "
                    "frac_cold_behind_dirty_cache: after the 1 GiB of writes alone -- every line the kernel brings in evicts a dirty one, "
                    "so 54 MB of foreign write-back share the HBM with it (what rounds 2-3 reported as cold); frac_warm: kernel "
                    "begin/end timestamps on HIP events attached to the launch (hipExtLaunchKernelGGL) in the K timed steps, where the "
                    "operands sit in the 256 MiB infinity cache; 52 B x source points with a correspondence, vs the 8 TB/s HBM peak"}
        result["network_pose_after_timed_steps"] = counts["network_pose"]
        if conv_prof:
            if world == 1 and not args.no_live_pmc:
                LIVE_PMC["rows"] = live_pmc_traffic(args.amp or "float32")
                if LIVE_PMC["rows"]:
                    LIVE_PMC["source"] = ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes over "
                                          "tools/conv_layers.py (one launch per kernel / pass / layer shape at the bench's batch size)")
            result["roofline"], _ = conv_roofline(conv_prof, args, result["ms_per_step"], prof_steps=max(1, evented["steps"]))
            result["roofline"]["measured_in"] = roofline_pass
            if world == 1:
                # every instrumented convolution launch, in a few more steps (single process only: under DDP a step is collective)
                PROFILE_STEPS = 4
                _lib.profile_begin(PROFILE_STEPS * 200)
                timed_region(PROFILE_STEPS, run_step)
                rows_all, untimed = _lib.profile_end(1024)
                _, result["conv_profile"] = conv_roofline(rows_all, args, result["ms_per_step"], prof_steps=PROFILE_STEPS)
                result["conv_profile"]["note"] = (f"kernel begin/end timestamps of every convolution launch in {PROFILE_STEPS} further steps; "
                                                  "the timed steps instrument the dominant family only, in every fourth step")
        else:
            result["roofline"] = result["roofline_loss"]
        result["kernels"] = rows
        if world == 1 and not args.amp and "hip trunk" in result["config"]["cnn_impl"] and args.conv_table:
            result["roofline_cnn"] = conv_table(args, device)
        if world == 1:
            if args.long_steps > 0:
                el, _ = timed_region(args.long_steps, run_step)
                result["long_run"] = {"steps": args.long_steps, "ms_per_step": round(1e3 * el / args.long_steps, 3),
                                      "value": round(args.batch * args.long_steps / el, 3)}
            if args.feed_steps > 0 and (graphed is None or not graphed.captured):
                loader = torch.utils.data.DataLoader(dataset=trainer.dataset, batch_size=args.batch, shuffle=False,
                                                     collate_fn=Trainer.list_collate, drop_last=True, num_workers=0)
                moved = {"bytes": 0}

                def epochs():
                    while True:
                        for b in DevicePrefetcher(loader, device):
                            yield b
                it = epochs()

                def fed_step():
                    b = next(it)
                    moved["bytes"] += sum(v.numel() * v.element_size() for d in b for v in d.values() if torch.is_tensor(v))
                    return run_step(b)
                for _ in range(3):
                    fed_step()
                moved["bytes"] = 0
                el, _ = timed_region(args.feed_steps, fed_step)
                result["feed"] = {"steps": args.feed_steps, "value": round(args.batch * args.feed_steps / el, 3), "unit": "scan-pairs/s",
                                  "ms_per_step": round(1e3 * el / args.feed_steps, 3), "feed_GB_s": round(moved["bytes"] / el / 1e9, 3),
                                  "MB_per_pair": round(moved["bytes"] / (args.batch * args.feed_steps) / 1e6, 3),
                                  "note": "same steps with every batch read from pinned host memory through DataLoader + DevicePrefetcher "
                                          "(async H2D one batch ahead on a side stream)"}
            tree = None
            if args.feed_steps > 0 and args.disk_pairs >= args.batch and (graphed is None or not graphed.captured):
                tree = make_disk_tree(args)
                result["feed_disk"] = disk_feed_leg(args, cfg, device, tree, run_step, timed_region, result["value"], args.feed_steps)
            if not args.amp and args.autocast_steps > 0 and (graphed is None or not graphed.captured):
                result["autocast"] = autocast_leg(args, device, host_batches, batches, timed_region, tree)
            if tree is not None:
                import shutil
                shutil.rmtree(tree["path"], ignore_errors=True)
            if not args.amp and args.variant_steps > 0 and (graphed is None or not graphed.captured):
                if args.width != 720:
                    result["shipped_image"] = variant_leg(args, device, host_batches, batches, timed_region, args.variant_steps, width=720)
                    result["shipped_image"]["note"] = ("the same step on the reference's shipped KITTI image size (config/config_datasets.yaml:21: "
                                                       "64x720; same raw scans): the HIP stem + trunk with overhanging tiles, not the headline")
                result["untrained_network"] = variant_leg(args, device, host_batches, batches, timed_region, args.variant_steps, pretrained=False)
                result["untrained_network"]["note"] = ("the headline workload with the network as torch initialises it (a run from scratch with "
                                                       "unsupervised_at_start: True): random poses, so the exact search walks the whole image for "
                                                       "every query; the headline uses the state the reference's identity pre-training leaves")
            if not args.amp and args.shipped_steps > 0 and (graphed is None or not graphed.captured):
                try:
                    sc = shipped_config_leg(args, device, timed_region, enqueue, steps_b1=args.shipped_steps, steps_b8=max(8, args.shipped_steps // 5))
                    ks = None if args.no_live_pmc else live_kernel_sum(1, "graph")
                    if ks:
                        b1 = sc["batch_1"]
                        ks["product_loop_vs_kernel_sum"] = round(b1["product_loop_packed_feed_2_workers"]["ms_per_step"] / ks["kernel_sum_ms"], 3)
                        if "graph_resident" in b1:
                            ks["graph_resident_vs_kernel_sum"] = round(b1["graph_resident"]["ms_per_step"] / ks["kernel_sum_ms"], 3)
                        ks["note"] = ("rocprofv3 --kernel-trace child over tools/shipped_step.py 1 graph: sum of the kernel durations of one "
                                      "replayed batch-1 step (median steady-state step) and its start-to-start time inside the trace")
                    sc["batch_1"]["kernel_sum"] = ks
                    result["shipped_config"] = sc
                except Exception as e:                               # noqa: BLE001 -- the leg is informative
                    result["shipped_config"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.amp and args.ddp_steps > 0 and (graphed is None or not graphed.captured):
                try:
                    result["ddp_rank"] = ddp_rank_leg(args, device, host_batches, batches, timed_region, enqueue, steps=args.ddp_steps)
                except Exception as e:                               # noqa: BLE001 -- the leg is informative
                    result["ddp_rank"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_cpu_baseline:
                result["cpu_baseline"], result["cpu_baseline_online_normals"] = cpu_baseline(args, cfg)
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    trace("at the final barrier")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
