"""Host-side logic that needs no GPU: config surface, model layout, the step's loss bookkeeping against the reference
vectors (geometry evaluated by the oracle backend), and data parallelism over 2 gloo ranks."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import util
from tests.conftest import ROOT


def _small_cfg(H, W, B, **over):
    gm = util.load_golden("model_small")
    opts = dict(factor_fewer_resnet_channels=int(gm["cfg::factor_fewer_resnet_channels"]), resnet_outputs=int(gm["cfg::resnet_outputs"]),
                unsupervised_at_start=True, inference_only=False, batch_size=B)
    opts.update(over)
    cfg = util.repo_config(H, W, device="cpu", **opts)
    sd = {k[4:]: torch.from_numpy(v) for k, v in gm.items() if k.startswith("sd::")}
    return cfg, sd


def _samples(g, B):
    out = []
    for j in range(B):
        s = {k: torch.from_numpy(g[f"s{j}::{k}"]) for k in ("scan_1", "scan_2", "normal_list_1", "normal_list_2")}
        s["dataset"] = "kitti"
        out.append(s)
    return out


def test_config_files_have_the_reference_surface():
    from delora_amd import config as cfgmod
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    for key in ("horizontal_field_of_view", "epsilon_range", "min_num_points_in_neighborhood_to_determine_point_class",
                "datasets", "device", "store_dataset_in_RAM", "num_dataloader_workers", "unsupervised_at_start",
                "inference_only", "use_jit", "batch_size", "learning_rate", "lambda_po2pl", "use_dropout",
                "random_point_cloud_rotations", "normal_loss", "point_to_point_loss", "point_to_plane_loss",
                "plane_to_plane_loss", "po2po_alone", "normalization_scaling", "activation_fct", "resnet_outputs",
                "pre_feature_extraction", "layers", "factor_fewer_resnet_channels", "use_single_mlp_at_output", "experiment"):
        assert key in cfg, key
    k = cfg["kitti"]
    assert (k["vertical_cells"], k["horizontal_cells"], k["horizontal_cells_preprocessing"]) == (64, 720, 2250)
    assert k["vertical_field_of_view"] == [-24.5, 2.0] and k["neighborhood_side_length"] == [7, 11]
    assert cfg["learning_rate"] == 1e-5 and cfg["batch_size"] == 1 and cfg["normal_loss"] == "squared"
    cfgmod.degrees_to_radians(cfg)
    assert np.isclose(cfg["kitti"]["vertical_field_of_view"][0], -24.5 * np.pi / 180.0)


def test_reference_import_names_resolve():
    import delora_amd.compat  # noqa: F401
    import deploy.trainer, losses.icp_losses, models.model, preprocessing.normal_computation, utility.projection  # noqa: E401,F401
    assert deploy.trainer.Trainer.__mro__[1].__name__ == "Deployer"


def test_model_matches_reference_on_cpu():
    from delora_amd.models.model import OdometryModel
    g = util.load_golden("model_small")
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 1)
    m = OdometryModel(cfg)
    m.load_state_dict(sd)
    with torch.no_grad():
        t, q = m(torch.from_numpy(g["image_1"]).unsqueeze(0), torch.from_numpy(g["image_2"]).unsqueeze(0))
    assert np.allclose(t.numpy(), g["translation"], atol=1e-6) and np.allclose(q.numpy(), g["quaternion"], atol=1e-6)


def test_geometry_handler_matches_reference_vectors():
    from delora_amd.models.model_parts import GeometryHandler
    g = util.load_golden("geometry")
    T = GeometryHandler.get_transformation_matrix_quaternion(torch.from_numpy(g["t"]), torch.from_numpy(g["q"]), torch.device("cpu"))
    assert np.allclose(T.numpy(), g["T"], atol=1e-6)


@pytest.mark.parametrize("name", ["b1", "b2"])
def test_step_bookkeeping_matches_reference(name):
    """Trainer.step with the oracle geometry backend reproduces the reference step: poses, loss terms (incl. the
    (B-j)/B weighting), gradient norms."""
    from delora_amd.deploy.trainer import Trainer
    g = util.load_golden("step_" + name)
    B = len(g["picks"])
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), B)
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    tr.raw_model.load_state_dict(sd)
    ep = tr.new_epoch_losses()
    tr.optimizer.zero_grad()
    ep, T = tr.step(preprocessed_dicts=_samples(g, B), epoch_losses=ep)
    assert np.allclose(T.detach().numpy(), g["T"], atol=1e-5)
    for key in ("loss_epoch", "loss_point_cloud_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch"):
        assert np.isclose(float(ep[key]), g["ep::" + key], rtol=2e-5), key
    assert int(ep["visible_pixels_epoch"]) == int(g["ep::visible_pixels_epoch"])
    for k, p in tr.raw_model.named_parameters():
        assert np.isclose(float(p.grad.double().norm()), float(g["gradnorm::" + k]), rtol=1e-3, atol=1e-9), k


def _ddp_worker(rank, world, port, return_dict):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from delora_amd.deploy.trainer import Trainer
        torch.set_num_threads(2)
        g = util.load_golden("step_b2")
        cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 1)
        tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
        tr.raw_model.load_state_dict(sd)
        assert tr.world_size == 2 and tr.rank == rank
        ep = tr.new_epoch_losses()
        tr.optimizer.zero_grad()
        ep, T = tr.step(preprocessed_dicts=[_samples(g, 2)[rank]], epoch_losses=ep)
        tr.steps_per_epoch_effective = 1
        red = tr._reduce_metrics(ep)
        grads = {k: p.grad.clone() for k, p in tr.raw_model.named_parameters()}
        after = {k: v.clone() for k, v in tr.raw_model.state_dict().items()}
        return_dict[rank] = {"T": T.detach().clone(), "grads": grads, "after": after, "reduced": red,
                             "local_loss": float(ep["loss_epoch"])}
    finally:
        torch.distributed.destroy_process_group()


def _timeline_worker(rank, world, port, return_dict):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from delora_amd.deploy.ddp_trace import DdpTimeline
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(), torch.nn.Linear(256, 8))
        ref = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(), torch.nn.Linear(256, 8))
        ref.load_state_dict(net.state_dict())
        ddp = torch.nn.parallel.DistributedDataParallel(net, bucket_cap_mb=0.1)
        tl = DdpTimeline().attach(ddp, last_grad_param=net[0].weight)
        x = torch.randn((4, 64), generator=torch.Generator().manual_seed(10 + rank))
        for _ in range(3):
            ddp.zero_grad(set_to_none=True)
            tl.begin_step()
            ddp(x).square().sum().backward()
            tl.end_step()
        # the hook must do what DDP's default does: the averaged gradient of the two ranks' losses
        xs = [torch.randn((4, 64), generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
        sum(ref(v).square().sum() for v in xs).backward()
        worst = max(float((p.grad - q.grad / world).abs().max()) for p, q in zip(net.parameters(), ref.parameters()))
        return_dict[rank] = {"summary": tl.summary(), "worst": worst}
    finally:
        torch.distributed.destroy_process_group()


def test_ddp_timeline_records_buckets_and_keeps_the_default_reduction():
    """deploy/ddp_trace.py on a 2-rank gloo group: the communication hook averages the gradients exactly as DDP's default does, every
    bucket of a step has a hand-over and a completion time, the bytes add up to the model's gradients, and the exposed part of the
    all-reduce is a non-negative number of milliseconds (bench.py --gpus N reports this per rank)."""
    ret = mp.Manager().dict()
    mp.spawn(_timeline_worker, args=(2, 36500 + (os.getpid() % 2000), ret), nprocs=2, join=True)
    nbytes = 4 * (64 * 256 + 256 + 256 * 256 + 256 + 256 * 8 + 8)
    for r in (0, 1):
        assert ret[r]["worst"] < 1e-6
        s = ret[r]["summary"]
        assert s["steps"] == 3 and s["bytes_per_step"] == nbytes and len(s["buckets"]) >= 2
        assert s["exposed_allreduce_ms"] is not None and s["exposed_allreduce_ms"] >= 0.0 and s["backward_end_ms"] > 0
        assert all(b["done_ms"] >= b["ready_ms"] for b in s["buckets"])


def test_two_ranks_reproduce_the_single_process_batch():
    """2 ranks x B=1 (gloo) == 1 process x B=2: same poses, same averaged gradients, same Adam update, same loss --
    i.e. the loss weights follow the sample's index in the GLOBAL batch."""
    from delora_amd.deploy.trainer import Trainer
    g = util.load_golden("step_b2")
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 2)
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    tr.raw_model.load_state_dict(sd)
    ep = tr.new_epoch_losses()
    tr.optimizer.zero_grad()
    ep, T = tr.step(preprocessed_dicts=_samples(g, 2), epoch_losses=ep)
    single_grads = {k: p.grad.clone() for k, p in tr.raw_model.named_parameters()}
    single_after = tr.raw_model.state_dict()
    port = 29500 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_ddp_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        assert torch.allclose(ret[r]["T"][0], T[r].detach(), atol=1e-6)
        for k in single_grads:
            # DDP averages; the step scales each rank's loss by world_size so that the average equals the global sum.
            # (tolerance: the CNN runs with batch 1 per rank vs batch 2, i.e. different CPU conv summation orders)
            a, b = ret[r]["grads"][k], single_grads[k]
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-4 * float(b.abs().max())), k
        for k in single_after:
            # Adam's first step is +-lr per weight: only weights whose gradient is ~0 may move differently
            off = (ret[r]["after"][k] - single_after[k]).abs() > 2e-7
            assert float(off.float().mean()) < 0.01, k
        assert np.isclose(ret[r]["reduced"]["loss_epoch"], float(ep["loss_epoch"]), rtol=1e-5)
    assert np.isclose(ret[0]["local_loss"] + ret[1]["local_loss"], float(ep["loss_epoch"]), rtol=1e-5)


def test_short_batch_is_rejected_explicitly():
    from delora_amd.deploy.trainer import Trainer
    g = util.load_golden("step_b2")
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 2)
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    with pytest.raises(ValueError):
        tr.step(preprocessed_dicts=_samples(g, 1), epoch_losses=tr.new_epoch_losses())


def test_product_deployer_defaults_to_the_hip_backend_and_refuses_cpu():
    from delora_amd import _lib
    from delora_amd.deploy.deployer import Deployer
    from delora_amd.deploy.step_geometry import HipStepGeometry
    g = util.load_golden("step_b1")
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 1)
    dep = Deployer(cfg, dataset=util.ListDataset([]))
    assert isinstance(dep.geo, HipStepGeometry)
    with pytest.raises(_lib.DeloraHipError):                    # CPU tensors: no fallback
        dep.step(preprocessed_dicts=_samples(g, 1), epoch_losses=None)


def test_preprocessed_dataset_reads_reference_layout(tmp_path):
    from delora_amd.data.dataset import PreprocessedPointCloudDataset
    root = tmp_path / "seq"
    for sub in ("scans", "normals"):
        os.makedirs(root / "03" / sub)
    rng = np.random.default_rng(0)
    for i in range(4):
        np.save(root / "03" / "scans" / f"{i:06d}.npy", rng.normal(size=(50 + i, 3)).astype(np.float32))
        np.save(root / "03" / "normals" / f"{i:06d}.npy", rng.normal(size=(50 + i, 3)).astype(np.float32))
    cfg = util.repo_config(16, 128)
    cfg["kitti"]["preprocessed_path"] = str(root)
    cfg["kitti"]["data_identifiers"] = [3]
    ds = PreprocessedPointCloudDataset(cfg)
    assert len(ds) == 3                                         # pairs (t, t+1)
    s = ds[1]
    assert s["scan_1"].shape == (1, 3, 51) and s["scan_2"].shape == (1, 3, 52) and s["dataset"] == "kitti"
    assert set(s) >= {"index", "index_dataset", "index_sequence", "index_scan", "dataset", "scan_1", "scan_2",
                      "normal_list_1", "normal_list_2"}


def test_xyz_only_tree_and_consecutive_pair_reuse(tmp_path, monkeypatch):
    """a sequence tree without normals/ is an xyz-only dataset (normals are then estimated online: half the bytes per
    pair), and the scan shared by two consecutive samples is decoded ONCE when the samples are visited in order -- also through a
    DataLoader with worker processes (each worker decodes whole batches)."""
    from delora_amd.data import dataset as dsmod, synthetic
    scans = [np.random.default_rng(i).normal(size=(3, 40 + i)).astype(np.float32) for i in range(9)]
    synthetic.write_tree(str(tmp_path), scans, sequence=0)
    cfg = util.repo_config(16, 128)
    cfg["kitti"]["preprocessed_path"] = str(tmp_path)
    cfg["kitti"]["data_identifiers"] = [0]
    ds = dsmod.PreprocessedPointCloudDataset(cfg)
    assert len(ds) == 8 and not ds.load_normals
    loads = []
    real = np.load
    monkeypatch.setattr(dsmod.np, "load", lambda path, *a, **k: (loads.append(os.path.basename(str(path))), real(path, *a, **k))[1])
    got = [ds[i] for i in range(8)]
    assert len(loads) == 9, loads                                # 9 scans for 8 consecutive pairs, not 16
    for i, s in enumerate(got):
        assert s["normal_list_1"] is None and s["normal_list_2"] is None
        assert torch.equal(s["scan_1"][0], torch.from_numpy(scans[i])) and torch.equal(s["scan_2"][0], torch.from_numpy(scans[i + 1]))
    monkeypatch.undo()
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, collate_fn=list, drop_last=True, num_workers=2,
                                         prefetch_factor=2, persistent_workers=True)
    for epoch in range(2):
        seen = [int(s["index"]) for batch in loader for s in batch]
        assert seen == list(range(8))
    batch = next(iter(loader))
    assert torch.equal(batch[3]["scan_2"][0], torch.from_numpy(scans[4]))
    # a tree WITH normals keeps the reference's behaviour, and a mismatch between the two directories is an error
    synthetic.write_tree(str(tmp_path / "n"), scans[:3], sequence=0, normals=[-s for s in scans[:3]])
    cfg["kitti"]["preprocessed_path"] = str(tmp_path / "n")
    ds2 = dsmod.PreprocessedPointCloudDataset(cfg)
    assert len(ds2) == 2 and torch.equal(ds2[1]["normal_list_2"][0], torch.from_numpy(-scans[2]))
    os.remove(tmp_path / "n" / "00" / "normals" / "000002.npy")
    with pytest.raises(Exception):
        dsmod.PreprocessedPointCloudDataset(cfg)
    assert not dsmod.PreprocessedPointCloudDataset(dict(cfg, load_normal_lists=False)).load_normals
    # mixed trees -- one sequence with stored normals, one without -- are rejected in either order instead of silently dropping the
    # stored normals of the sequences scanned first (advisor, round 4); load_normal_lists: false trains all of them xyz-only
    synthetic.write_tree(str(tmp_path / "m"), scans[:3], sequence=0, normals=[-s for s in scans[:3]])
    synthetic.write_tree(str(tmp_path / "m"), scans[3:6], sequence=1)
    cfg["kitti"]["preprocessed_path"] = str(tmp_path / "m")
    for order in ([0, 1], [1, 0]):
        cfg["kitti"]["data_identifiers"] = order
        with pytest.raises(Exception, match="mixed trees"):
            dsmod.PreprocessedPointCloudDataset(cfg)
    assert len(dsmod.PreprocessedPointCloudDataset(dict(cfg, load_normal_lists=False))) == 4


@pytest.mark.parametrize("with_normals", [False, True])
def test_packed_feed_delivers_the_batches_of_the_sampler_in_the_steps_layout(tmp_path, with_normals):
    """data/feed.py: PackedFeed -- worker processes decode the reference's on-disk files straight into shared batch slots, planar
    [C, sumN] + CSR offsets in the order b0.scan_1, b0.scan_2, b1.scan_1 ... (what dl_project reads): every batch of the sampler, in
    order, with exactly the bytes of the files, over two epochs, with fewer slots than batches (slots are recycled)."""
    from delora_amd.data import dataset as dsmod, feed, synthetic
    rng = np.random.default_rng(3)
    scans = [rng.normal(size=(3, 30 + 7 * i)).astype(np.float32) for i in range(11)]
    normals = [rng.normal(size=s.shape).astype(np.float32) for s in scans] if with_normals else None
    synthetic.write_tree(str(tmp_path), scans, sequence=0, normals=normals)
    cfg = util.repo_config(16, 128)
    cfg["kitti"]["preprocessed_path"] = str(tmp_path)
    cfg["kitti"]["data_identifiers"] = [0]
    cfg["num_dataloader_workers"] = 2
    ds = dsmod.PreprocessedPointCloudDataset(cfg)
    assert not feed.packed_feed_applicable(ds, cfg, torch.device("cpu"))          # a CUDA feed: the CPU keeps the DataLoader
    order = [[3, 0], [9, 4], [1, 2], [7, 8], [5, 6]]
    pf = feed.PackedFeed(ds, order, 2, torch.device("cpu"), workers=2, points_per_scan=200, slots=3, ahead=2)
    try:
        for epoch in range(2):
            got = list(pf)
            assert len(got) == len(order)
            for batch, idx in zip(got, order):
                assert len(batch) == 2 and batch.with_lists == with_normals and batch.dataset == "kitti"
                want = [scans[i + d] for i in idx for d in (0, 1)]
                offs = batch.offs.tolist()
                assert offs == list(np.cumsum([0] + [w.shape[1] for w in want])) and batch.max_points == max(w.shape[1] for w in want)
                assert torch.equal(batch.pts[:3], torch.from_numpy(np.concatenate(want, axis=1)))
                if with_normals:
                    assert torch.equal(batch.pts[3:], torch.from_numpy(np.concatenate([normals[i + d] for i in idx for d in (0, 1)], axis=1)))
        # an epoch abandoned after its first batch (an exception in the training loop) leaves tasks and results behind: the next epoch
        # must not see them
        it = iter(pf)
        next(it)
        del it
        again = list(pf)
        assert [b.offs.tolist() for b in again] == [b.offs.tolist() for b in got]
        # one consumer at a time: an iterator that is still alive when a later epoch starts is stale and says so (it used to wait
        # 120 s for batches the later epoch had drained)
        stale = iter(pf)
        next(stale)
        assert len(list(pf)) == len(order)
        with pytest.raises(RuntimeError, match="abandoned"):
            next(stale)
        # no worker processes at all (num_dataloader_workers: 0 taken literally: the consumer decodes in-process) and a dataset held in
        # RAM ([3,M] tensors instead of [M,3] files) deliver the same batches
        for ram in (False, True):
            ds0 = dsmod.PreprocessedPointCloudDataset(dict(cfg, store_dataset_in_RAM=ram))
            inproc = feed.PackedFeed(ds0, order, 2, torch.device("cpu"), workers=0, points_per_scan=200, slots=2)
            assert inproc.procs == []
            same = list(inproc)
            assert [b.offs.tolist() for b in same] == [b.offs.tolist() for b in got] and all(torch.equal(a.pts, b.pts) for a, b in zip(same, got))
            inproc.close()
        # a batch beyond the slot capacity is an error of the feed, not a silent truncation
        small = feed.PackedFeed(ds, [[0, 1]], 2, torch.device("cpu"), workers=1, points_per_scan=20)
        with pytest.raises(RuntimeError):
            list(small)
        small.close()
    finally:
        pf.close()


def test_synthetic_sequence_is_a_chain_of_small_motions():
    from delora_amd.data import synthetic
    scans, poses = synthetic.make_sequence(5, 4, rings=8, azimuth_steps=90)
    assert len(scans) == 4 and all(s.shape[0] == 3 and s.dtype == np.float32 and s.shape[1] > 300 for s in scans)
    for a, b in zip(poses[:-1], poses[1:]):
        rel = np.linalg.inv(a) @ b
        assert 0.1 < np.linalg.norm(rel[:3, 3]) < 1.0 and np.degrees(np.arccos(np.clip((np.trace(rel[:3, :3]) - 1) / 2, -1, 1))) < 3.0


@pytest.mark.parametrize("name", ["tower_relu", "single_mlp"])
def test_model_architecture_switches_match_reference(name):
    """pre_feature_extraction / use_single_mlp_at_output / relu: same parameter names, same outputs as the reference."""
    from delora_amd.models.model import OdometryModel
    g = util.load_golden("model_" + name)
    over = {k[5:]: (v.item() if v.shape == () else v) for k, v in g.items() if k.startswith("cfg::")}
    over = {k: (str(v) if k == "activation_fct" else (bool(v) if k in ("pre_feature_extraction", "use_single_mlp_at_output") else int(v)))
            for k, v in over.items()}
    cfg = util.repo_config(16, 128, device="cpu", **over)
    m = OdometryModel(cfg)
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd)
    with torch.no_grad():
        t, q = m(torch.from_numpy(g["image_1"]), torch.from_numpy(g["image_2"]))
    assert np.allclose(t.numpy(), g["translation"], atol=2e-6) and np.allclose(q.numpy(), g["quaternion"], atol=2e-6)


def test_normalized_step_matches_reference():
    """normalization_scaling (deployer.py:222-235,344-346) with lambda_po2pl = 10: scans divided by the mean range,
    translation rescaled afterwards."""
    from delora_amd.deploy.trainer import Trainer
    g = util.load_golden("step_b1_norm")
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 1, normalization_scaling=True, lambda_po2pl=float(g["lambda_po2pl"]))
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    tr.raw_model.load_state_dict(sd)
    ep = tr.new_epoch_losses()
    tr.optimizer.zero_grad()
    ep, T = tr.step(preprocessed_dicts=_samples(g, 1), epoch_losses=ep)
    assert np.allclose(T.detach().numpy(), g["T"], rtol=1e-4, atol=1e-5)
    for key in ("loss_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch"):
        assert np.isclose(float(ep[key]), g["ep::" + key], rtol=5e-5), key


def test_pose_integration_matches_reference():
    """utility.poses.compute_poses / write_poses_to_text_file against the reference's trajectory (bit-exact, incl. the file)."""
    from delora_amd.utility import poses
    g = util.load_golden("poses")
    P = poses.compute_poses([t.reshape(1, 4, 4) for t in g["transformations"]])
    assert np.array_equal(P, g["poses"])
    import tempfile
    fn = tempfile.mktemp()
    poses.write_poses_to_text_file(fn, P)
    assert open(fn).read().encode() == g["text"].tobytes()
    bad = np.eye(4); bad[:3, :3] *= 1.5
    assert not poses.check_validity_so3(bad[:3, :3])


def test_tester_writes_kitti_pose_files(tmp_path):
    """Checkpoint -> Tester over a two-pair 'sequence' (oracle geometry backend on CPU) -> pose file = compute_poses(T)."""
    from delora_amd.deploy.tester import Tester
    from delora_amd.deploy.trainer import Trainer
    from delora_amd.utility import poses
    g = util.load_golden("step_b2")
    cfg, sd = _small_cfg(int(g["H"]), int(g["W"]), 1)
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    tr.raw_model.load_state_dict(sd)
    ck = str(tmp_path / "m.pth")
    tr.save_checkpoint(ck, 0, 0.0)
    cfg2, _ = _small_cfg(int(g["H"]), int(g["W"]), 1)
    cfg2.update(checkpoint=ck, inference_only=True, output_dir=str(tmp_path), run_name="t", mode="testing")
    cfg2["kitti"]["data_identifiers"] = [9]
    samples = []
    for j, s in enumerate(_samples(g, 2)):
        s.update(index=j, index_dataset=0, index_sequence=0, index_scan=j)
        samples.append(s)
    te = Tester(cfg2, dataset=util.ListDataset(samples), geometry_backend=util.OracleStepGeometry())
    te.test()
    assert len(te.written) == 1 and te.written[0]["poses_text"].endswith("t_poses_text_file_kitti_09.txt")
    T = np.load(te.written[0]["transformations"])
    assert T.shape == (2, 1, 4, 4) and np.allclose(T[:, 0], g["T"], atol=1e-5)
    P = np.load(te.written[0]["poses"])
    assert np.allclose(P, poses.compute_poses(list(T)))
    assert len(open(te.written[0]["poses_text"]).read().strip().splitlines()) == 3


def test_prefetcher_passes_batches_through_on_cpu():
    from delora_amd.data.feed import DevicePrefetcher
    batches = [[{"a": torch.ones(3) * i, "name": "x"}] for i in range(4)]
    got = list(DevicePrefetcher(batches, torch.device("cpu")))
    assert len(got) == 4 and all(float(b[0]["a"][0]) == i for i, b in enumerate(got))
    assert list(DevicePrefetcher([], torch.device("cpu"))) == []


def _mixed_setup(device):
    """Two samples from different sensor blocks: kitti 16x128 (vFoV -24.5..2) and darpa 16x96 (vFoV +-22.5)."""
    from delora_amd.data import synthetic
    g = util.load_golden("step_b2")
    cfg, sd = _small_cfg(16, 128, 2)
    cfg["device"] = torch.device(device)
    cfg["datasets"] = ["kitti", "darpa"]
    cfg["darpa"]["vertical_cells"], cfg["darpa"]["horizontal_cells"] = 16, 96
    cfg["darpa"]["vertical_field_of_view"] = [-22.5 * np.pi / 180.0, 22.5 * np.pi / 180.0]
    s_k = _samples(g, 1)[0]
    d1, d2, _ = synthetic.make_pair(5150, rings=16, azimuth_steps=120, vfov_deg=(-22.5, 22.5))
    s_d = {"scan_1": torch.from_numpy(d1).unsqueeze(0), "scan_2": torch.from_numpy(d2).unsqueeze(0),
           "normal_list_1": None, "normal_list_2": None, "dataset": "darpa"}
    return cfg, sd, [s_k, s_d]


def _terms_of(cfg, sd, samples, backend, device="cpu"):
    from delora_amd.deploy.deployer import Deployer
    import copy
    c = copy.deepcopy(cfg)
    c["batch_size"] = len(samples)
    dep = Deployer(c, dataset=util.ListDataset([]), geometry_backend=backend)
    dep.model.load_state_dict({k: v.to(device) for k, v in sd.items()})
    batch = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in s.items()} for s in samples]
    with torch.no_grad():
        ep, T = dep.step(preprocessed_dicts=batch, epoch_losses=None)
    return dep.last_step["loss_terms"].cpu(), T.cpu(), dep.last_step["losses"]


def test_mixed_sensor_batch_equals_per_sample_results():
    """BASELINE config 5 semantics: a batch mixing two image geometries gives every sample exactly the result it has in a
    batch of its own (sub-batch per geometry), and loss_pc uses the (B-j)/B weights over the sample order."""
    cfg, sd, samples = _mixed_setup("cpu")
    terms, T, losses = _terms_of(cfg, sd, samples, util.OracleStepGeometry())
    for j, smp in enumerate(samples):
        t1, T1, _ = _terms_of(cfg, sd, [smp], util.OracleStepGeometry())
        assert torch.allclose(terms[j], t1[0], rtol=1e-6, atol=1e-8) and torch.allclose(T[j], T1[0], atol=1e-7)
    c = terms[:, 0] + cfg["lambda_po2pl"] * terms[:, 1] + terms[:, 2]
    assert np.isclose(float(losses["loss_pc"]), float((2 * c[0] + 1 * c[1]) / 2), rtol=1e-6)


def _python_sources(*roots):
    out = []
    for root in roots:
        p = os.path.join(ROOT, root)
        if os.path.isfile(p):
            out.append(p)
            continue
        for d, _, files in os.walk(p):
            out += [os.path.join(d, f) for f in files if f.endswith(".py")]
    return out


def test_product_code_never_touches_the_oracle_or_the_reference_tree():
    """The oracle is test infrastructure: nothing under delora_amd/ or bin/ may import it, bench.py may do so only inside its
    cpu_baseline leg, and nothing that runs on the GPU box may read /root/reference (only tests/golden/make_golden.py,
    which generated the committed fixtures, does)."""
    import ast
    for path in _python_sources("delora_amd", "bin"):
        tree = ast.parse(open(path).read(), filename=path)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{path} imports the oracle"
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:                                            # module level and every function except cpu_baseline
        if isinstance(node, ast.FunctionDef) and node.name == "cpu_baseline":
            continue
        for sub in ast.walk(node):
            if isinstance(sub, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in sub.names] if isinstance(sub, ast.Import) else [sub.module or ""]
                assert not any(n == "oracle" or n.startswith("oracle.") for n in names), "bench.py imports the oracle outside cpu_baseline"
    runtime = _python_sources("delora_amd", "bin", "bench.py", "__graft_entry__.py", "tests", "oracle")
    for path in runtime:
        if path.endswith(os.path.join("golden", "make_golden.py")) or path.endswith("test_host_logic.py"):
            continue
        code = open(path).read()
        tree = ast.parse(code, filename=path)
        strings = [n.value for n in ast.walk(tree) if isinstance(n, ast.Constant) and isinstance(n.value, str)]
        docstrings = {ast.get_docstring(n) for n in ast.walk(tree)
                      if isinstance(n, (ast.Module, ast.FunctionDef, ast.ClassDef)) and ast.get_docstring(n)}
        for s in strings:
            if "/root/reference" in s:
                assert any(s.strip() == (d or "").strip() or s.strip() in (d or "") for d in docstrings), \
                    f"{path} uses /root/reference outside a docstring"


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_ring_ops_on_cpu_are_the_reference_op_sequence(act):
    """On CPU tensors the fused ring ops evaluate the reference's separate ops (resnet_modified.py:97-102,159-177)."""
    import torch.nn.functional as F
    from delora_amd.models.ring_ops import ring_act_pad, ring_act_pool_pad
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((2, 3, 5, 16), generator=gen)
    res = torch.randn((2, 3, 5, 18), generator=gen)
    f = torch.tanh if act == "tanh" else torch.relu
    wrap = lambda v: F.pad(v, (1, 1, 0, 0), mode="circular")          # noqa: E731
    assert torch.equal(ring_act_pad(x, act, pad=True, residual=res), wrap(f(x + res[..., 1:-1])))
    assert torch.equal(ring_act_pad(x, act, pad=False), f(x))
    want = wrap(F.max_pool2d(wrap(f(x)), kernel_size=3, stride=(1, 2), padding=(1, 0)))
    assert torch.equal(ring_act_pool_pad(x, act), want)


def test_tf_quaternion_restatements_agree_with_scipy():
    """quaternion_from_matrix / quaternion_matrix (restated tf.transformations, (x,y,z,w)) against scipy's Rotation on
    random rotations, the three trace <= 0 branches and the identity; the two functions invert each other."""
    from scipy.spatial.transform import Rotation
    from delora_amd.ros_utils import odometry
    rng = np.random.default_rng(5)
    mats = list(Rotation.random(50, random_state=7).as_matrix())
    mats += [np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0]), np.eye(3)]
    mats += list(Rotation.from_rotvec(rng.normal(size=(20, 3)) * 1e-4).as_matrix())       # near identity
    mats += list(Rotation.from_rotvec([[np.pi - 1e-3, 0, 0], [0, np.pi - 1e-3, 0], [0, 0, np.pi - 1e-3]]).as_matrix())
    for R in mats:
        M = np.eye(4)
        M[:3, :3] = R
        q = odometry.quaternion_from_matrix(M)
        assert np.isclose(np.linalg.norm(q), 1.0, atol=1e-12)
        ref = Rotation.from_matrix(R).as_quat()
        assert min(np.abs(q - ref).max(), np.abs(q + ref).max()) < 1e-9
        assert np.allclose(odometry.quaternion_matrix(q)[:3, :3], R, atol=1e-9)
    assert np.array_equal(odometry.quaternion_matrix([0.0, 0.0, 0.0, 0.0]), np.identity(4))


def test_scan_filter_of_the_inference_node():
    from delora_amd.ros_utils import odometry
    scan = np.array([[[1.0, 0.0, 2.0, 0.1, 5.0], [1.0, 3.0, 0.0, 0.1, 0.0], [1.0, 3.0, 2.0, 0.1, 1.0]]], dtype=np.float32)
    kept = odometry.filter_scans(scan)                 # exact-zero coordinate in points 1, 2, 4; point 3 closer than 0.3 m
    assert kept.shape == (1, 3, 1) and np.array_equal(kept[0, :, 0], [1.0, 1.0, 1.0])


@pytest.mark.parametrize("normalize", [False, True])
def test_scan_to_scan_odometry_core_on_cpu(normalize):
    """The ROS-free core of the inference node with the oracle's projection injected: nothing for the first scan, then the
    network's pose for (previous, current) in metres, and an integrated pose that is the product of the steps."""
    from delora_amd.data import synthetic
    from delora_amd.models import model as model_module, model_parts
    from delora_amd.ros_utils import odometry
    gm = util.load_golden("model_small")
    cfg = util.repo_config(16, 128, device="cpu", factor_fewer_resnet_channels=int(gm["cfg::factor_fewer_resnet_channels"]),
                           resnet_outputs=int(gm["cfg::resnet_outputs"]), normalization_scaling=normalize)
    cfg["integrate_odometry"] = True
    torch.manual_seed(11)
    net = model_module.OdometryModel(config=cfg)
    o_sensor = util.oracle_sensor(16, 128, cfg["kitti"]["vertical_field_of_view"], cfg["horizontal_field_of_view"])

    def project_pair(previous, current, sensor):
        return torch.cat([util.orc.project_to_img(c.cpu(), o_sensor)[0][:, [0, 1, 2, 3]] for c in (previous, current)], dim=0)

    odo = odometry.ScanToScanOdometry(cfg, model=net, project_pair=project_pair)
    scans = []
    for i in range(3):
        s1, s2, _ = synthetic.make_pair(300 + i, rings=16, azimuth_steps=140)
        scans.append(s1[None] if i % 2 == 0 else s2[None])
    assert odo.push(scans[0]) is None
    handler = model_parts.GeometryHandler(config=cfg)
    chain = np.eye(4)
    for k in (1, 2):
        out = odo.push(scans[k])
        prev = torch.from_numpy(odometry.filter_scans(scans[k - 1]))[:, :3]
        cur = torch.from_numpy(odometry.filter_scans(scans[k]))[:, :3]
        scale = 1.0
        if normalize:
            scale = float(torch.mean(torch.cat((torch.norm(cur, dim=1), torch.norm(prev, dim=1)), dim=1)))
            prev, cur = prev / scale, cur / scale
        imgs = project_pair(prev, cur, None)
        with torch.no_grad():
            t, q = net.eval()(imgs[0:1], imgs[1:2])
            T = handler.get_transformation_matrix_quaternion(translation=t, quaternion=q, device="cpu")[0].double().numpy()
        assert np.allclose(out["translation"], t[0].double().numpy() * scale, rtol=1e-5, atol=1e-7)
        assert np.allclose(odometry.quaternion_matrix(out["quaternion"])[:3, :3], T[:3, :3], atol=1e-6)
        step = odometry.quaternion_matrix(out["quaternion"])
        step[:3, 3] = out["translation"]
        chain = chain @ step
        assert np.allclose(out["T_0_t"], chain, atol=1e-9)
        assert np.allclose(out["global_translation"], chain[:3, 3], atol=1e-9)


def test_rosnode_config_has_the_reference_cli_fields(tmp_path):
    from delora_amd import config as config_module
    cfg = config_module.rosnode_config("ckpt.pth", "kitti", "/velodyne_points", "velodyne", True, config_dir=os.path.join(ROOT, "config"))
    assert cfg["datasets"] == ["kitti"] and cfg["lidar_topic"] == "/velodyne_points" and cfg["lidar_frame"] == "velodyne"
    assert cfg["integrate_odometry"] is True and cfg["checkpoint"] == "ckpt.pth" and cfg["use_dropout"] is False
    assert abs(cfg["horizontal_field_of_view"][1] - np.deg2rad(179.9)) < 1e-6          # radians, as the node expects


def _synthetic_samples(n, H=16, az=140):
    from delora_amd.data import synthetic
    out = []
    for j in range(n):
        s1, s2, _ = synthetic.make_pair(700 + j, rings=H, azimuth_steps=az)
        out.append({"dataset": "kitti", "scan_1": torch.from_numpy(s1).unsqueeze(0), "scan_2": torch.from_numpy(s2).unsqueeze(0),
                    "normal_list_1": None, "normal_list_2": None})
    return out


def _ddp_worker_n(rank, world, per_rank, port, identity_phase, return_dict):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from delora_amd.deploy.trainer import Trainer
        torch.set_num_threads(2)
        cfg, sd = _small_cfg(16, 128, per_rank, unsupervised_at_start=not identity_phase)
        tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
        tr.raw_model.load_state_dict(sd)
        ep = tr.new_epoch_losses()
        tr.optimizer.zero_grad()
        mine = _synthetic_samples(world * per_rank)[rank * per_rank:(rank + 1) * per_rank]
        ep, T = tr.step(preprocessed_dicts=mine, epoch_losses=ep)
        tr.steps_per_epoch_effective = 1
        return_dict[rank] = {"T": T.detach().clone(), "grads": {k: p.grad.clone() for k, p in tr.raw_model.named_parameters()},
                             "loss": float(ep["loss_epoch"]), "loss_pc": float(ep["loss_point_cloud_epoch"]),
                             "visible_local": float(ep["visible_pixels_epoch"]), "reduced": tr._reduce_metrics(ep),
                             "after": {k: v.clone() for k, v in tr.raw_model.state_dict().items()}}
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("identity_phase", [False, True])
def test_three_ranks_times_two_samples_equal_one_batch_of_six(identity_phase):
    """Odd world size, more than one sample per rank, online normals, both training phases: 3 ranks x B=2 (gloo) give the
    poses, the summed loss and the averaged gradients of ONE process with B=6 (the (Bg - j)/Bg weights follow the global
    sample index; the identity phase fits the last sample of the GLOBAL batch only)."""
    from delora_amd.deploy.trainer import Trainer
    world, per_rank = 3, 2
    cfg, sd = _small_cfg(16, 128, world * per_rank, unsupervised_at_start=not identity_phase)
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    tr.raw_model.load_state_dict(sd)
    ep = tr.new_epoch_losses()
    tr.optimizer.zero_grad()
    ep, T = tr.step(preprocessed_dicts=_synthetic_samples(world * per_rank), epoch_losses=ep)
    single = {k: p.grad.clone() for k, p in tr.raw_model.named_parameters()}
    ret = mp.Manager().dict()
    mp.spawn(_ddp_worker_n, args=(world, per_rank, 31500 + (os.getpid() % 2000), identity_phase, ret), nprocs=world, join=True)
    for r in range(world):
        assert torch.allclose(ret[r]["T"], T[r * per_rank:(r + 1) * per_rank].detach(), atol=2e-6)
        for k, b in single.items():
            a = ret[r]["grads"][k]
            assert torch.allclose(a, b, rtol=2e-3, atol=3e-4 * float(b.abs().max()) + 1e-12), k
    assert np.isclose(sum(ret[r]["loss"] for r in range(world)), float(ep["loss_epoch"]), rtol=1e-5)
    assert np.isclose(sum(ret[r]["loss_pc"] for r in range(world)), float(ep["loss_point_cloud_epoch"]), rtol=1e-5)


@pytest.mark.parametrize("identity_phase", [False, True])
def test_eight_ranks_times_one_sample_equal_one_batch_of_eight(identity_phase):
    """BASELINE configs[2] in miniature: 8 ranks x B=1 (gloo) == 1 process x B=8, in both training phases -- poses, the (Bg - j)/Bg
    weights of the global sample index, the identity loss of the LAST rank's sample only, the averaged gradients, the Adam update,
    and the epoch metrics after the all-reduce (the loss terms are sums over ranks; `visible_pixels` is the last sample's count,
    contributed by the last rank alone)."""
    from delora_amd.deploy.trainer import Trainer
    world, per_rank = 8, 1
    cfg, sd = _small_cfg(16, 128, world * per_rank, unsupervised_at_start=not identity_phase)
    tr = Trainer(cfg, dataset=util.ListDataset([]), geometry_backend=util.OracleStepGeometry())
    tr.raw_model.load_state_dict(sd)
    ep = tr.new_epoch_losses()
    tr.optimizer.zero_grad()
    ep, T = tr.step(preprocessed_dicts=_synthetic_samples(world * per_rank), epoch_losses=ep)
    single = {k: p.grad.clone() for k, p in tr.raw_model.named_parameters()}
    single_after = {k: v.clone() for k, v in tr.raw_model.state_dict().items()}
    ret = mp.Manager().dict()
    mp.spawn(_ddp_worker_n, args=(world, per_rank, 33500 + (os.getpid() % 2000), identity_phase, ret), nprocs=world, join=True)
    for r in range(world):
        assert torch.allclose(ret[r]["T"], T[r:r + 1].detach(), atol=2e-6)
        for k, b in single.items():
            assert torch.allclose(ret[r]["grads"][k], b, rtol=2e-3, atol=3e-4 * float(b.abs().max()) + 1e-12), k
        for k, b in single_after.items():
            # Adam's first step moves every weight by +-lr: only weights whose gradient is ~0 may move differently
            off = (ret[r]["after"][k] - b).abs() > 2e-7
            assert float(off.float().mean()) < 0.01, k
        # every rank holds the same reduced metrics, and they are the single process' epoch sums
        for key in ("loss_epoch", "loss_point_cloud_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch", "visible_pixels_epoch"):
            assert np.isclose(ret[r]["reduced"][key], float(ep[key]), rtol=1e-5, atol=1e-9), (r, key)
    if identity_phase:                                  # only the last rank's sample is fitted to the identity (deployer.py:324-338)
        assert all(ret[r]["loss"] == 0.0 for r in range(world - 1)) and ret[world - 1]["loss"] > 0.0
    assert np.isclose(sum(ret[r]["loss"] for r in range(world)), float(ep["loss_epoch"]), rtol=1e-5)
    assert ret[world - 1]["visible_local"] == float(ep["visible_pixels_epoch"]) > 0


def test_rotation_head_gradient_does_not_depend_on_which_rows_share_the_batch_norm():
    """model.py:114 divides the raw quaternions by ONE norm over the whole batch; under data parallelism each rank takes it over
    its local rows.  Every row is re-normalised inside quaternion -> R, so the loss is homogeneous of degree 0 in each row and
    the gradient that reaches the rotation head is the same whichever rows share that scalar."""
    from delora_amd.models.model_parts import GeometryHandler
    g = torch.Generator().manual_seed(11)
    raw0 = torch.randn((6, 4), generator=g, dtype=torch.float64)
    target = torch.randn((6, 3, 3), generator=g, dtype=torch.float64)

    def grad(groups):
        raw = raw0.clone().requires_grad_(True)
        loss = 0.0
        for rows in groups:                       # one "rank" per group: its own whole-batch norm
            rot = raw[rows] / torch.norm(raw[rows])
            R = GeometryHandler.quaternion_to_rot_matrix(rot)
            loss = loss + ((R - target[rows]) ** 2).sum()
        loss.backward()
        return raw.grad, float(loss)

    g1, l1 = grad([slice(0, 6)])
    g2, l2 = grad([slice(0, 3), slice(3, 6)])
    g3, l3 = grad([slice(0, 1), slice(1, 6)])
    assert abs(l1 - l2) < 1e-12 * abs(l1) and abs(l1 - l3) < 1e-12 * abs(l1)
    assert float((g1 - g2).abs().max()) < 1e-12 * float(g1.abs().max())
    assert float((g1 - g3).abs().max()) < 1e-12 * float(g1.abs().max())


def test_loss_weight_matrix_and_epoch_accumulation():
    """Deployer._loss_weights = the reference's in-loop weighting written as a matrix ((Bg - j)/Bg on sample j of the global
    batch, lambda on the point-to-plane column); Deployer._accumulate adds step values into the epoch dict without aliasing
    the step's own tensors."""
    from delora_amd.deploy.deployer import Deployer

    class Stub:
        rank = 1
    stub = Stub()
    W, col = Deployer._loss_weights(stub, 2, 6, 0.5, torch.device("cpu"), torch.float32)
    assert torch.allclose(W, torch.tensor([[4 / 6, 2 / 6, 4 / 6], [3 / 6, 1.5 / 6, 3 / 6]]))           # j_global = 2, 3
    assert torch.allclose(col, torch.tensor([1 / 6, 0.5 / 6, 1 / 6]))
    assert Deployer._loss_weights(stub, 2, 6, 0.5, torch.device("cpu"), torch.float32)[0] is W      # cached
    ep = {"a": 0.0, "b": 0.0, "untouched": 0.0}
    v1 = [torch.tensor(1.5), torch.tensor(2, dtype=torch.int32)]
    Deployer._accumulate(ep, ["a", "b"], v1)
    assert float(ep["a"]) == 1.5 and float(ep["b"]) == 2.0 and ep["untouched"] == 0.0
    shared = torch.tensor(0.25)
    Deployer._accumulate(ep, ["a", "b"], [shared, shared])                                         # the same tensor twice
    assert float(ep["a"]) == 1.75 and float(ep["b"]) == 2.25 and float(shared) == 0.25 and float(v1[0]) == 1.5
    ep2 = {"a": 3.0, "b": 0.0}                                                                       # a non-zero python start
    Deployer._accumulate(ep2, ["a", "b"], [torch.tensor(1.0), torch.tensor(1.0)])
    assert float(ep2["a"]) == 4.0 and float(ep2["b"]) == 1.0


def test_hip_trunk_shape_gate_decides_the_weight_layout():
    """`ResNetModified.hip_path_takes` is the shape test the Deployer uses before it stores the trunk's weights channels-last
    (the layout the HIP kernels read in place).  Since round 4 the image size is free (tiles hang over the edges of feature maps
    that do not divide): BASELINE.json's sizes AND the reference's shipped 64 x 720 / 64 x 512 (config/config_datasets.yaml:21, :47)
    take the HIP path; only a width that does not halve twice (the stem's two stride-2 stages) and narrow networks do not."""
    from delora_amd.models.model import OdometryModel
    cfg = util.repo_config(64, 2048)
    model = OdometryModel(cfg)
    takes = model.resnet.hip_path_takes
    assert takes(64, 2048) and takes(64, 1024) and takes(128, 2048)
    assert takes(64, 720) and takes(64, 512) and takes(16, 100) and not takes(64, 722) and not takes(64, 721)
    narrow = OdometryModel(util.repo_config(64, 720, factor_fewer_resnet_channels=8, resnet_outputs=64))
    assert not narrow.resnet.hip_path_takes(64, 720)
    cfg_m = util.repo_config(64, 2048)
    cfg_m["cnn_impl"] = "modules"
    assert not OdometryModel(cfg_m).resnet.hip_path_takes(64, 2048)


@pytest.mark.parametrize("blocker", ["random_point_cloud_rotations", "normalization_scaling", "world_size", "mixed"])
def test_graphed_step_runs_the_eager_step_when_capture_is_not_possible(blocker):
    """`hip_graph: true` together with augmentation / range normalisation / several ranks / a mixed-sensor batch: every call must
    run the eager step on the caller's own list of dicts (advisor r03: it packed the batch and Deployer.step raised ValueError)."""
    from delora_amd.deploy import step_geometry
    from delora_amd.deploy.graph_step import GraphedStep

    class FakeTrainer:
        world_size, batch_size = 1, 2
        config = {"normalization_scaling": False, "random_point_cloud_rotations": False}
        seen = []

        class optimizer:
            param_groups, state = [], {}

            @staticmethod
            def zero_grad(set_to_none=True):
                pass

        @staticmethod
        def new_epoch_losses():
            return {}

        def step(self, preprocessed_dicts, epoch_losses):
            self.seen.append(preprocessed_dicts)
            return epoch_losses, None

    tr = FakeTrainer()
    tr.config = dict(tr.config)
    batch = [{"dataset": "kitti", "scan_1": torch.zeros(1, 3, 10), "scan_2": torch.zeros(1, 3, 12), "normal_list_1": None, "normal_list_2": None}
             for _ in range(2)]
    if blocker == "world_size":
        tr.world_size = 2
    elif blocker == "mixed":
        batch[1]["dataset"] = "darpa"
    else:
        tr.config[blocker] = True
    g = GraphedStep(tr, batch)
    assert not g.eligible and not g.captured and tr.seen == []
    g(batch)
    g(batch)
    assert g.fallback_steps == 2 and len(tr.seen) == 2
    assert all(isinstance(b, list) and not isinstance(b, step_geometry.PackedBatch) for b in tr.seen)
    with pytest.raises(ValueError):
        g(None)


def test_slab_plan_of_the_merged_weight_gradient_launches():
    """dl_wgrad_batch_plan (host only): the slab counts the merged weight-gradient launches use.  For the layer groups of
    the network at 64x2048, batch 8: every layer gets between 1 and `chunks` slabs, the simulated makespan stays within 10 % of the even
    share, far fewer slabs than one launch per layer needs, and the plan is reproducible (memoised)."""
    import ctypes
    from delora_amd import _lib
    lib = _lib.load()

    def plan(tiles, chunks, slots, cost):
        n = len(tiles)
        t, c, ns = (ctypes.c_int32 * n)(*tiles), (ctypes.c_int32 * n)(*chunks), (ctypes.c_int32 * n)()
        mk = lib.dl_wgrad_batch_plan(ctypes.cast(t, ctypes.c_void_p), ctypes.cast(c, ctypes.c_void_p), n, slots, cost, ctypes.cast(ns, ctypes.c_void_p))
        assert mk > 0, lib.dl_last_error()
        return list(ns), mk

    groups = {  # name: (tiles, chunks, slots, partial cost)
        "fp32 Winograd-domain, layer2-4": ([4] * 3 + [16] * 3 + [64] * 3, [4096] * 3 + [2048] * 3 + [512] * 3, 256, 10),
        "bf16 stride-1 >= 128 channels": ([2] * 3 + [8] * 3 + [32] * 3, [1024] * 3 + [512] * 3 + [128] * 3, 256, 10),
        "fp32 direct, layer1": ([1] * 4, [8192] * 4, 512, 20),
        "fp32 direct, stride (1,2)": ([2, 8], [8192, 4096], 512, 20),
    }
    for name, (tiles, chunks, slots, cost) in groups.items():
        ns, mk = plan(tiles, chunks, slots, cost)
        share = sum(t * c for t, c in zip(tiles, chunks)) / slots
        assert all(1 <= a <= c for a, c in zip(ns, chunks)), (name, ns)
        assert mk <= 1.10 * share, (name, ns, mk, share)
        alone = [max(1, min(c, -(-slots // t))) for t, c in zip(tiles, chunks)]       # slabs of a launch per layer
        assert sum(t * a for t, a in zip(tiles, ns)) <= 0.6 * sum(t * a for t, a in zip(tiles, alone)), (name, ns, alone)
        assert plan(tiles, chunks, slots, cost) == (ns, mk)
    assert plan([4] * 3 + [16] * 3 + [64] * 3, [4096] * 3 + [2048] * 3 + [512] * 3, 256, 10)[0] == [4, 4, 4, 2, 2, 2, 1, 1, 1]
    n1 = (ctypes.c_int32 * 1)(0)
    assert lib.dl_wgrad_batch_plan(ctypes.cast(n1, ctypes.c_void_p), ctypes.cast(n1, ctypes.c_void_p), 1, 256, 10, ctypes.cast(n1, ctypes.c_void_p)) < 0


def test_merged_weight_gradients_are_routed_like_the_single_launches(monkeypatch):
    """ring_conv.wgrad_batch (host logic, no GPU): stride-1 3x3 layers go to the Winograd-domain batch (from 64 input channels on: the rule
    of wgrad_nhwc is 128, a merged launch lowers it), everything else to the direct batch, results come back in the order of the items, and with
    USE_WINOGRAD_WGRAD off everything takes the direct kernel."""
    import torch
    from delora_amd.models import ring_conv as rc
    calls = []

    def fake(items, size_fn, run_fn, what, *extra):
        calls.append((run_fn, [tuple(x.shape) + (ks,) + tuple(st) for x, g, ks, st in items]))
        return [("dw", run_fn, tuple(x.shape), ks, tuple(st)) for x, g, ks, st in items]

    monkeypatch.setattr(rc, "_wgrad_batch_call", fake)
    shapes = [(2, 8, 64, 128, 128, 3, (1, 1)), (2, 8, 64, 64, 64, 3, (1, 1)), (2, 8, 64, 128, 256, 3, (1, 2)), (2, 8, 64, 256, 256, 3, (1, 1)),
              (2, 8, 64, 128, 256, 1, (1, 2)), (2, 8, 64, 64, 128, 3, (1, 1))]
    items = [(torch.empty((N, H, W, C)), torch.empty((N, rc.out_size(H, st[0]), rc.out_size(W, st[1]), K)), ks, st) for (N, H, W, C, K, ks, st) in shapes]
    out = rc.wgrad_batch(items)
    by_fn = dict(calls)
    assert rc.WINO_WGRAD_MIN_C_BATCHED == 64        # (inside a merged launch the 64-channel layers take the Winograd-domain kernel too)
    assert [s[3] for s in by_fn["dl_wino_wgrad3x3_batch_nhwc_f32"]] == [128, 64, 256, 64]               # input channels of the Winograd layers
    assert len(by_fn["dl_conv2d_wgrad_batch_nhwc_f32"]) == 2
    for (N, H, W, C, K, ks, st), o in zip(shapes, out):
        wino = ks == 3 and st == (1, 1) and C >= 64
        assert o == ("dw", "dl_wino_wgrad3x3_batch_nhwc_f32" if wino else "dl_conv2d_wgrad_batch_nhwc_f32", (N, H, W, C), ks, st)
    calls.clear()
    monkeypatch.setattr(rc, "USE_WINOGRAD_WGRAD", False)
    rc.wgrad_batch(items)
    assert [fn for fn, _ in calls] == ["dl_conv2d_wgrad_batch_nhwc_f32"] and len(calls[0][1]) == 6


def test_relative_pose_errors_known_answers():
    """utility/poses.relative_pose_errors (KITTI-benchmark-style relative error; the metric of the convergence evidence): a 10 % scale
    error of a straight trajectory is a 10 % translation error and no rotation error; a constant yaw-rate error of 0.01 rad per metre is
    that rotation error; identical trajectories have none; segments longer than the trajectory are skipped."""
    from delora_amd.utility import poses as P

    def chain(steps):
        out = [np.eye(4)]
        for T in steps:
            out.append(out[-1] @ T)
        return np.stack(out)

    def step(dz, yaw=0.0):
        T = np.eye(4)
        T[2, 3] = dz
        c, s_ = np.cos(yaw), np.sin(yaw)
        T[:3, :3] = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])          # yaw about the camera frame's y axis
        return T

    gt = chain([step(1.0)] * 60)
    r = P.relative_pose_errors(chain([step(1.1)] * 60), gt, lengths_m=(5.0, 10.0, 1000.0), step=1)
    assert abs(r["translation"] - 0.1) < 1e-12 and r["rotation_rad_per_m"] < 1e-7 and set(r["per_length"]) == {5.0, 10.0}
    assert r["segments"] == (61 - 5) + (61 - 10)
    r = P.relative_pose_errors(chain([step(1.0, 0.01)] * 60), gt, lengths_m=(5.0,), step=1)
    assert abs(r["rotation_rad_per_m"] - 0.01) < 1e-9 and abs(r["rotation_deg_per_100m"] - np.degrees(1.0)) < 1e-6
    r = P.relative_pose_errors(gt, gt, lengths_m=(5.0,))
    assert r["translation"] < 1e-15 and r["segments"] == 6
    assert P.relative_pose_errors(gt, gt, lengths_m=(100.0,))["segments"] == 0
    with pytest.raises(ValueError):
        P.relative_pose_errors(gt[:-1], gt)


def test_graph_policy_reads_the_config_key():
    """Trainer.graph_policy: `hip_graph` absent = eager (round 6: a training trajectory must not depend on wall-clock measurements;
    "auto" = measure and decide is opt-in), true / false force it, and a configuration a capture cannot serve (augmentation, range normalisation, several ranks, fp16 loss scaling) or a CPU run is always eager."""
    from delora_amd.deploy.trainer import Trainer

    class T(Trainer):
        def __init__(self, config, device="cuda", world=1, scaler=None):            # noqa: super().__init__ not called on purpose
            self.config, self.device, self.world_size, self.grad_scaler = config, torch.device(device), world, scaler

    base = {"normalization_scaling": False, "random_point_cloud_rotations": False, "use_jit": False}
    assert T(dict(base)).graph_policy() == "off"
    for v, want in ((True, "on"), ("true", "on"), (False, "off"), ("off", "off"), ("auto", "auto"), ("Auto", "auto")):
        assert T(dict(base, hip_graph=v)).graph_policy() == want, v
    assert T(dict(base, hip_graph=True), device="cpu").graph_policy() == "off"
    assert T(dict(base, hip_graph=True), world=2).graph_policy() == "off"
    assert T(dict(base, hip_graph=True, normalization_scaling=True)).graph_policy() == "off"
    assert T(dict(base, hip_graph=True, random_point_cloud_rotations=True)).graph_policy() == "off"
    assert T(dict(base, hip_graph=True), scaler=object()).graph_policy() == "off"


def test_packed_feed_of_eight_ranks_with_two_workers_each_on_one_host():
    """BASELINE configs[2]'s feed in miniature (tools/feed_ranks.py): 8 rank processes x (1 consumer + 2 workers) = 24 processes on
    this host, each rank reading its DistributedSampler shard of one tree through its own PackedFeed.  Every rank must get every
    batch of its shard in every epoch; the aggregate rate is recorded (the full-size run is tools/feed_ranks.py on the GPU box)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))              # (importable by name: the rank processes are spawned, not forked)
    try:
        import feed_ranks as mod
        out = mod.run(ranks=8, workers=2, batch=1, scans=17, epochs=3, rings=16, cells=200)
    finally:
        sys.path.pop(0)
    assert out["processes"] == 24 and out["batches_per_rank"] == 3 * (16 // 8) and out["pairs_per_s_per_rank_min"] > 0
    print(out)
