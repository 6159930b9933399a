"""The full training step (projection -> CNN -> T -> correspondences -> losses -> backward -> Adam) on the GPU against
the vectors recorded from the reference's own ``Trainer.step`` (tests/golden/step_b{1,2}.npz), and the mirrors of
the reference's module API against golden vectors / the oracle."""
import os

import numpy as np
import pytest
import torch

from tests import util
from tests.util import orc

pytestmark = pytest.mark.gpu
REL = 1e-4


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _small_model_cfg(g, H, W, **over):
    return util.repo_config(H, W, device="cuda:0", factor_fewer_resnet_channels=int(g["cfg::factor_fewer_resnet_channels"]),
                            resnet_outputs=int(g["cfg::resnet_outputs"]), unsupervised_at_start=True, inference_only=False, **over)


def _state_dict(g, dev):
    return {k[4:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("sd::")}


def test_model_forward_matches_reference():
    from delora_amd.models.model import OdometryModel
    dev = _dev()
    g = util.load_golden("model_small")
    cfg = _small_model_cfg(g, int(g["H"]), int(g["W"]))
    m = OdometryModel(cfg).to(dev)
    assert set(m.state_dict().keys()) == set(_state_dict(g, dev).keys())
    m.load_state_dict(_state_dict(g, dev))
    i1, i2 = torch.from_numpy(g["image_1"]).to(dev).unsqueeze(0), torch.from_numpy(g["image_2"]).to(dev).unsqueeze(0)
    with torch.no_grad():
        t, q = m(i1, i2)
        t2, q2 = m(torch.cat((i1, i2), dim=1))
    assert torch.equal(t, t2) and torch.equal(q, q2)
    assert np.allclose(t.cpu().numpy(), g["translation"], rtol=REL, atol=1e-6)
    assert np.allclose(q.cpu().numpy(), g["quaternion"], rtol=REL, atol=1e-6)


def test_full_model_has_reference_parameter_layout():
    from delora_amd.models.model import OdometryModel
    cfg = util.repo_config(64, 2048)
    m = OdometryModel(cfg)
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 11876019
    assert len(sd) == 30 and tuple(sd["resnet.conv1.weight"].shape) == (64, 8, 3, 3)
    assert tuple(sd["fully_connected_rotation.3.weight"].shape) == (4, 100)


@pytest.mark.parametrize("name", ["b1", "b2"])
def test_training_step_matches_reference(name):
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()
    g = util.load_golden("step_" + name)
    gm = util.load_golden("model_small")
    B = len(g["picks"])
    cfg = _small_model_cfg(gm, int(g["H"]), int(g["W"]), batch_size=B)
    samples = []
    for j in range(B):
        s = {k: torch.from_numpy(g[f"s{j}::{k}"]).to(dev) for k in ("scan_1", "scan_2", "normal_list_1", "normal_list_2")}
        s["dataset"] = "kitti"
        samples.append(s)
    trainer = Trainer(cfg, dataset=util.ListDataset(samples))
    trainer.raw_model.load_state_dict(_state_dict(gm, dev))
    before = {k: v.detach().clone() for k, v in trainer.raw_model.state_dict().items()}
    ep = trainer.new_epoch_losses()
    trainer.optimizer.zero_grad()
    ep, T = trainer.step(preprocessed_dicts=[dict(s) for s in samples], epoch_losses=ep)
    torch.cuda.synchronize()
    T_ref = g["T"]
    assert np.allclose(T.detach().cpu().numpy(), T_ref, rtol=REL, atol=REL * np.abs(T_ref).max()), "poses"
    for key in ("loss_epoch", "loss_point_cloud_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch", "loss_po2po_epoch"):
        assert np.isclose(float(ep[key]), g["ep::" + key], rtol=REL, atol=1e-7), key
    assert int(ep["visible_pixels_epoch"]) == int(g["ep::visible_pixels_epoch"])
    worst = 0.0
    for k, p in trainer.raw_model.named_parameters():
        ref = float(g["gradnorm::" + k])
        worst = max(worst, abs(float(p.grad.double().norm()) - ref) / max(ref, 1e-9))
    util.measured(f"step_{name}: worst relative error of the per-parameter gradient norms vs the reference", worst, bound=1e-5)     # measured 1.1e-6
    after = trainer.raw_model.state_dict()
    for k in before:                                        # Adam's first step moves every weight by ~lr*sign(grad)
        got = float((after[k].double() - before[k].double()).sum())
        n = before[k].numel()
        assert abs(got - float(g["delta::" + k])) <= 1e-5 * max(2.0, 0.005 * n), f"Adam update {k}"


def test_identity_pretraining_loss_uses_last_sample_only():
    """deployer.py:324-338: with unsupervised_at_start False the loss is MSE(T_last, I)/B."""
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()
    g, gm = util.load_golden("step_b2"), util.load_golden("model_small")
    cfg = _small_model_cfg(gm, int(g["H"]), int(g["W"]), batch_size=2)
    cfg["unsupervised_at_start"] = False
    samples = [{**{k: torch.from_numpy(g[f"s{j}::{k}"]).to(dev) for k in ("scan_1", "scan_2", "normal_list_1", "normal_list_2")},
                "dataset": "kitti"} for j in range(2)]
    trainer = Trainer(cfg, dataset=util.ListDataset(samples))
    trainer.raw_model.load_state_dict(_state_dict(gm, dev))
    ep, T = trainer.step(preprocessed_dicts=[dict(s) for s in samples], epoch_losses=trainer.new_epoch_losses())
    expect = float(((T[1].detach() - torch.eye(4, device=dev)) ** 2).mean() / 2)
    assert np.isclose(float(ep["loss_epoch"]), expect, rtol=1e-5)
    assert np.isclose(float(ep["loss_point_cloud_epoch"]), g["ep::loss_point_cloud_epoch"], rtol=REL)   # still computed (quirk f)


def test_online_normals_step_runs_and_learns():
    """No normal lists in the samples: normals come from the projected images; a few steps reduce the loss."""
    from delora_amd.data.dataset import SyntheticPairDataset
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()
    cfg = util.repo_config(32, 256, device="cuda:0", factor_fewer_resnet_channels=4, resnet_outputs=128,
                           unsupervised_at_start=True, inference_only=False, batch_size=2, learning_rate=1e-4)
    ds = SyntheticPairDataset(cfg, "kitti", 2, rings=32, azimuth_steps=300)
    trainer = Trainer(cfg, dataset=ds)
    torch.manual_seed(0)
    batch = trainer.to_device([ds[0], ds[1]])
    first = None
    for it in range(12):
        ep = trainer.new_epoch_losses()
        trainer.optimizer.zero_grad()
        ep, T = trainer.step(preprocessed_dicts=[dict(b) for b in batch], epoch_losses=ep)
        val = float(ep["loss_epoch"])
        assert np.isfinite(val)
        first = val if first is None else first
    assert val < first


def test_checkpoint_layout_roundtrip(tmp_path):
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()
    gm = util.load_golden("model_small")
    cfg = _small_model_cfg(gm, 16, 128, batch_size=1)
    tr = Trainer(cfg, dataset=util.ListDataset([]))
    path = str(tmp_path / "ckpt.pth")
    tr.save_checkpoint(path, epoch=3, loss=0.5)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss", "parameters"}   # trainer.py:155-161
    cfg2 = _small_model_cfg(gm, 16, 128, batch_size=1)
    cfg2["checkpoint"] = path
    cfg2["unsupervised_at_start"] = False
    tr2 = Trainer(cfg2, dataset=util.ListDataset([]))
    assert cfg2["unsupervised_at_start"] is True
    for k, v in tr.raw_model.state_dict().items():
        assert torch.equal(v, tr2.raw_model.state_dict()[k])


# ------------------------------------------------------------------------------------ reference module API mirrors
def test_image_projection_layer_returns_reference_tuple():
    from delora_amd.utility.projection import ImageProjectionLayer
    dev = _dev()
    for name in ("small", "small_c6"):
        g = util.load_golden("proj_" + name)
        cfg = util.repo_config(int(g["H"]), int(g["W"]), device="cuda:0")
        layer = ImageProjectionLayer(cfg)
        scan = g["scan"]
        x = torch.from_numpy(scan).to(dev).view(1, scan.shape[0], -1)
        image, u, v, idx, pix = layer(input=x, dataset="kitti")
        o_sensor = util.oracle_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
        clean = ~util.tainted_pixels(scan, o_sensor)
        assert image.shape == g["image"].shape and u.shape == g["u"].shape and pix.shape[0] == 1 and pix.shape[2] == 2
        assert np.array_equal(image.cpu().numpy()[0][:, clean], g["image"][0][:, clean])
        assert np.allclose(u.cpu().numpy(), g["u"], atol=2e-3) and np.allclose(v.cpu().numpy(), g["v"], atol=2e-3)
        amb = set(np.nonzero(util.ambiguity_mask(scan, o_sensor))[0].tolist())
        got_idx = idx.cpu().numpy()
        assert set(got_idx.tolist()) ^ set(g["idx"].tolist()) <= amb
        if np.array_equal(np.sort(got_idx), np.sort(g["idx"])):
            assert np.array_equal(got_idx, g["idx"])                      # ascending range, the reference's order
            assert np.array_equal(pix.cpu().numpy(), g["pix"])
        # the kept points, gathered the way Deployer.step does, sit at the returned pixels
        p = pix[0].cpu().numpy()
        assert np.array_equal(image.cpu().numpy()[0, :3, p[:, 0], p[:, 1]].T if False else image.cpu().numpy()[0][:3][:, p[:, 0], p[:, 1]],
                              scan[:3][:, got_idx])


def test_normals_computer_returns_reference_triple():
    from delora_amd.preprocessing.normal_computation import NormalsComputer
    dev = _dev()
    g = util.load_golden("normals_small")
    cfg = util.repo_config(int(g["H"]), int(g["W"]), device="cuda:0")
    nc = NormalsComputer(config=cfg, dataset_name="kitti")
    normals, has, pts = nc.compute_normal_vectors(image=torch.from_numpy(g["image"]).to(dev))
    assert np.array_equal(pts.cpu().numpy(), g["points"])
    assert (has.cpu().numpy() != g["has"]).mean() <= 2e-3
    both = has.cpu().numpy() & g["has"]
    a, b = normals.cpu().numpy()[both].astype(np.float64), g["normals"][both].astype(np.float64)
    ang = np.arctan2(np.linalg.norm(np.cross(a, b), axis=1), np.sum(a * b, axis=1))
    # the conditioning-aware contract of tests/test_gpu_geometry.check_normals: where the two smallest eigenvalues are
    # separated (relative gap > 1e-3) the angle obeys 2e-4 + 50 eps32 / gap; nearly degenerate pixels are free
    lam = g["eigenvalues"][both].astype(np.float64)
    gap = (lam[:, 1] - lam[:, 0]) / np.maximum(lam[:, 2], 1e-30)
    well = gap > 1e-3
    outside = np.mean(ang[well] > 2e-4 + 50 * 6e-8 / gap[well])
    util.measured("NormalsComputer: median angle to the reference [rad]", float(np.median(ang)), bound=1e-5)
    util.measured("NormalsComputer: fraction of well-conditioned normals outside the conditioning bound", float(outside), bound=1e-3)
    util.measured("NormalsComputer: fraction of all normals further than 5e-3 rad from the reference (ill-conditioned pixels)",
                  float(np.mean(ang > 5e-3)), bound=5e-3)


@pytest.mark.parametrize("mode,p2p", [("squared", False), ("linear", True)])
def test_icp_losses_module_on_lists(mode, p2p):
    from delora_amd.losses.icp_losses import ICPLosses
    from delora_amd.models.model_parts import GeometryHandler
    dev = _dev()
    g = util.load_golden("loss_pair")
    cfg = util.repo_config(16, 128, device="cuda:0", normal_loss=mode, point_to_point_loss=p2p)
    mod = ICPLosses(cfg)
    tgt, tgt_n = (torch.from_numpy(g[k]).to(dev).view(1, 3, -1) for k in ("tgt", "tgt_n"))
    src, src_n = (torch.from_numpy(g[k]).to(dev).view(1, 3, -1) for k in ("src", "src_n"))
    for qname in ("identity", "true", "random"):
        key = f"{mode}_{'p2p' if p2p else 'nop2p'}_{qname}"
        t = torch.from_numpy(g[key + "_t"]).to(dev).requires_grad_(True)
        q = torch.from_numpy(g[key + "_q"]).to(dev).requires_grad_(True)
        T = GeometryHandler.get_transformation_matrix_quaternion(translation=t, quaternion=q, device=dev)
        assert np.allclose(T.detach().cpu().numpy(), g[key + "_T"], atol=2e-6)
        T.retain_grad()
        s_t = T[:, :3, :3].matmul(src) + T[:, :3, 3].view(-1, 3, 1)
        n_t = T[:, :3, :3].matmul(src_n)
        losses, plotting = mod(source_point_cloud_transformed=s_t, source_normal_list_transformed=n_t,
                               target_point_cloud=tgt, target_normal_list=tgt_n, compute_pointwise_loss_bool=False)
        (losses["loss_po2po"] + 2.0 * losses["loss_po2pl"] + 0.5 * losses["loss_pl2pl"]).sum().backward()
        got = np.array([float(losses[k]) for k in ("loss_po2po", "loss_po2pl", "loss_pl2pl")])
        assert np.allclose(got, g[key + "_losses"], rtol=REL, atol=1e-8), key
        ref_g = g[key + "_gradT"]
        assert np.allclose(T.grad.cpu().numpy(), ref_g, rtol=1e-3, atol=1e-4 * np.abs(ref_g).max()), key
        assert plotting["scan_2_transformed"].shape[2] == int(g[key + "_pairs"])


def test_normalized_step_matches_reference_on_gpu():
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()
    g, gm = util.load_golden("step_b1_norm"), util.load_golden("model_small")
    cfg = _small_model_cfg(gm, int(g["H"]), int(g["W"]), batch_size=1, normalization_scaling=True,
                           lambda_po2pl=float(g["lambda_po2pl"]))
    sample = {**{k: torch.from_numpy(g[f"s0::{k}"]).to(dev) for k in ("scan_1", "scan_2", "normal_list_1", "normal_list_2")},
              "dataset": "kitti"}
    tr = Trainer(cfg, dataset=util.ListDataset([sample]))
    tr.raw_model.load_state_dict(_state_dict(gm, dev))
    ep, T = tr.step(preprocessed_dicts=[dict(sample)], epoch_losses=tr.new_epoch_losses())
    assert np.allclose(T.detach().cpu().numpy(), g["T"], rtol=REL, atol=REL * np.abs(g["T"]).max())
    for key in ("loss_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch"):
        assert np.isclose(float(ep[key]), g["ep::" + key], rtol=REL), key
    for k, p in tr.raw_model.named_parameters():
        assert np.isclose(float(p.grad.double().norm()), float(g["gradnorm::" + k]), rtol=2e-3, atol=1e-9), k


@pytest.mark.parametrize("name", ["tower_relu", "single_mlp"])
def test_model_architecture_switches_on_gpu(name):
    from delora_amd.models.model import OdometryModel
    dev = _dev()
    g = util.load_golden("model_" + name)
    over = {k[5:]: v for k, v in g.items() if k.startswith("cfg::")}
    over = {k: (str(v) if k == "activation_fct" else (bool(v) if k in ("pre_feature_extraction", "use_single_mlp_at_output") else int(v)))
            for k, v in over.items()}
    cfg = util.repo_config(16, 128, device="cuda:0", **over)
    m = OdometryModel(cfg).to(dev)
    m.load_state_dict({k[4:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("sd::")})
    with torch.no_grad():
        t, q = m(torch.from_numpy(g["image_1"]).to(dev), torch.from_numpy(g["image_2"]).to(dev))
    assert np.allclose(t.cpu().numpy(), g["translation"], rtol=REL, atol=1e-5)
    assert np.allclose(q.cpu().numpy(), g["quaternion"], rtol=REL, atol=1e-5)


def test_train_loop_with_prefetcher_and_checkpoints(tmp_path):
    """Trainer.train(): DataLoader -> pinned async H2D prefetch -> steps -> per-epoch metrics -> checkpoint files with the
    reference's layout; identity pre-training first (unsupervised_at_start False), as the reference's default."""
    from delora_amd.data.dataset import SyntheticPairDataset
    from delora_amd.deploy.trainer import Trainer
    _dev()
    cfg = util.repo_config(16, 128, device="cuda:0", factor_fewer_resnet_channels=8, resnet_outputs=64, batch_size=2,
                           unsupervised_at_start=False, inference_only=False, checkpoint_dir=str(tmp_path), learning_rate=1e-3)
    ds = SyntheticPairDataset(cfg, "kitti", 4, rings=16, azimuth_steps=160)
    tr = Trainer(cfg, dataset=ds)
    tr.train(max_epochs=2)
    latest = tmp_path / "test_latest_checkpoint.pth"
    assert latest.exists() and (tmp_path / "test_checkpoint_epoch_0.pth").exists()
    ck = torch.load(str(latest), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss", "parameters"}
    assert np.isfinite(ck["loss"])


@pytest.mark.parametrize("with_normals", [False, True])
def test_training_from_the_on_disk_format_through_the_packed_feed(tmp_path, with_normals):
    """`Trainer.train` on a tree in the reference's on-disk format (src/preprocessing/preprocesser.py:64-68) with worker
    processes: the batches come through data/feed.py: PackedFeed (page-locked shared slots, one H2D copy per batch) as PackedBatch
    objects, and the epoch's metrics equal those of the same run fed by the plain DataLoader (same sample order, same
    kernels -- the packed buffer is just the concatenation `HipStepGeometry.prepare` would build on the device)."""
    from delora_amd.data import feed, synthetic
    from delora_amd.data.dataset import PreprocessedPointCloudDataset
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()
    scans, _ = synthetic.make_sequence(11, 7, rings=16, azimuth_steps=160)
    normals = None
    if with_normals:
        g = np.random.default_rng(2)
        normals = []
        for sc in scans:
            n = g.normal(size=sc.shape).astype(np.float32)
            normals.append(n / np.linalg.norm(n, axis=0, keepdims=True))
    synthetic.write_tree(str(tmp_path / "data"), scans, sequence=0, normals=normals)
    results = []
    # the plain DataLoader (config packed_feed: false), the packed feed without worker processes (num_dataloader_workers: 0, the YAML's
    # default: the consumer decodes in-process), with two worker processes, and over a dataset held in RAM
    for workers, packed, ram in ((0, False, False), (0, True, False), (2, True, False), (0, True, True)):
        cfg = util.repo_config(16, 128, device="cuda:0", factor_fewer_resnet_channels=8, resnet_outputs=64, batch_size=2,
                               unsupervised_at_start=True, inference_only=False, checkpoint_dir=str(tmp_path), learning_rate=1e-4,
                               num_dataloader_workers=workers, store_dataset_in_RAM=ram, shuffle_training_data=False, packed_feed=packed,
                               hip_graph=False)
        cfg["kitti"]["preprocessed_path"] = str(tmp_path / "data")
        cfg["kitti"]["data_identifiers"] = cfg["kitti"]["training_identifiers"] = [0]
        torch.manual_seed(7)
        tr = Trainer(cfg, dataset=PreprocessedPointCloudDataset(cfg))
        loader, sampler = tr.make_dataloader()
        assert isinstance(loader, feed.PackedFeed) == packed
        if packed:
            assert loader.pinned, "the slots of the feed must be page-locked (hipHostRegister) on a GPU box"
            assert len(loader.procs) == workers
        tr.steps_per_epoch_effective = len(loader)
        metrics = tr._reduce_metrics(tr.train_epoch(epoch=0, dataloader=loader))
        results.append((metrics, [p.detach().clone() for p in tr.raw_model.parameters()]))
        if packed:
            loader.close()
    m0, p0 = results[0]
    assert len(tr.dataset) == 6 and np.isfinite(m0["loss_epoch"]) and m0["loss_epoch"] > 0
    for name, (m1, p1) in zip(("in-process", "2 workers", "in-process, dataset in RAM"), results[1:]):
        for k in m0:
            assert np.isclose(m0[k], m1[k], rtol=1e-6, atol=1e-9), (name, k, m0[k], m1[k])
        util.measured(f"packed feed ({name}) vs DataLoader ({'stored normals' if with_normals else 'xyz only'}): largest parameter difference after one epoch",
                      max(float((a - b).abs().max()) for a, b in zip(p0, p1)), bound=2.5e-7)     # (0.3 - 1.0e-7 from run to run: 1 % of ONE Adam step at this rate)


@pytest.mark.parametrize("feed_kind", ["dataloader", "packed-in-process", "packed-2-workers", "packed-in-process-reverted"])
def test_product_loop_switches_to_graph_replay_and_keeps_the_trajectory(tmp_path, feed_kind):
    """`hip_graph: auto` (opt-in since round 6) in the product loop, on the reference's default batch size 1 with the full-width network on the HIP
    path: `Trainer.train_epoch` times its first eager steps, decides (forced here by `hip_graph_auto_threshold: 0`), captures the step
    in the MIDDLE of an epoch -- the capture's warm-up steps are rolled back -- and replays every later batch, from the DataLoader
    (`packed_feed: false`: lists of dicts packed scan by scan) and from the packed feed (in-process and with two workers: a
    PackedBatch, one device copy into the static buffers).  Epoch metrics and final weights must equal those of the eager run
    (`hip_graph: false`) on the same samples."""
    workers = 2 if feed_kind == "packed-2-workers" else 0
    # "reverted": the replays are judged not worth keeping (`hip_graph_min_gain` no replay can reach), so the loop goes eager -> capture ->
    # a few replays -> eager again, all inside one run: every hand-over must leave weights, Adam state and the epoch sums intact
    reverted = feed_kind.endswith("reverted")
    from delora_amd.data import feed, synthetic
    from delora_amd.data.dataset import PreprocessedPointCloudDataset
    from delora_amd.deploy.trainer import Trainer
    _dev()
    scans, _ = synthetic.make_sequence(21, 13, rings=16, azimuth_steps=160)
    g = np.random.default_rng(5)
    normals = []
    for sc in scans:
        n = g.normal(size=sc.shape).astype(np.float32)
        normals.append(n / np.linalg.norm(n, axis=0, keepdims=True))
    synthetic.write_tree(str(tmp_path / "data"), scans, sequence=0, normals=normals)
    runs = {}
    for mode in (False, "auto"):
        cfg = util.repo_config(16, 128, device="cuda:0", batch_size=1, unsupervised_at_start=True, inference_only=False,
                               checkpoint_dir=str(tmp_path), learning_rate=1e-5, num_dataloader_workers=workers, store_dataset_in_RAM=False,
                               shuffle_training_data=False, hip_graph=mode, hip_graph_auto_threshold=0.0, hip_graph_keep_if_slower=not reverted,
                               hip_graph_min_gain=0.999 if reverted else 0.03, packed_feed=feed_kind != "dataloader")
        cfg["kitti"]["preprocessed_path"] = str(tmp_path / "data")
        cfg["kitti"]["data_identifiers"] = cfg["kitti"]["training_identifiers"] = [0]
        torch.manual_seed(7)
        tr = Trainer(cfg, dataset=PreprocessedPointCloudDataset(cfg))
        assert tr.graph_policy() == ("off" if mode is False else "auto")
        loader, _ = tr.make_dataloader()
        assert isinstance(loader, feed.PackedFeed) == (feed_kind != "dataloader")
        tr.steps_per_epoch_effective = len(loader)
        metrics = [tr._reduce_metrics(tr.train_epoch(epoch=e, dataloader=loader)) for e in range(3)]
        torch.cuda.synchronize()
        runs[mode] = (metrics, [p.detach().clone() for p in tr.raw_model.parameters()], tr)
        if hasattr(loader, "close"):
            loader.close()
    tr = runs["auto"][2]
    probe = tr.PROBE_SKIP + tr.PROBE_STEPS
    if reverted:
        assert tr.graph_probe_result[True]["decision"].startswith("eager (replay measured") and tr._graphed is None
        # PROBE_STEPS step-to-step periods are measured before the verdict; the gap between two epochs is not one (round 6), so a trial
        # that crosses an epoch boundary (12 steps per epoch here) takes one replay more
        assert tr.graph_steps in (tr.PROBE_STEPS + 1, tr.PROBE_STEPS + 2), tr.graph_steps
    else:
        assert tr.graph_probe_result[True]["decision"] == "graph" and tr._graphed.captured
        assert tr.graph_steps == 3 * 12 - probe and tr._graphed.fallback_steps == 0, (tr.graph_steps, tr._graphed.fallback_steps)
    assert runs[False][2].graph_steps == 0
    worst, where = 0.0, None
    for e, (m_e, m_g) in enumerate(zip(runs[False][0], runs["auto"][0])):
        for k in m_e:
            d = abs(m_e[k] - m_g[k]) / max(abs(m_e[k]), 1e-12)
            if d > worst:
                worst, where = d, (e, k, m_e[k], m_g[k])
    if worst > 1e-5:
        print("graph replay vs eager: worst metric", where, "largest parameter difference",
              max(float((a - b).abs().max()) for a, b in zip(runs[False][1], runs["auto"][1])), "graph steps", tr.graph_steps,
              "probe", tr.graph_probe_result)
    util.measured(f"product loop, graph replay vs eager ({feed_kind}): worst relative difference of an epoch metric "
                  "over three epochs", worst, bound=1e-5)
    util.measured(f"product loop, graph replay vs eager ({feed_kind}): largest parameter difference after 36 steps",
                  max(float((a - b).abs().max()) for a, b in zip(runs[False][1], runs["auto"][1])), bound=2e-6)


def test_device_prefetcher_on_gpu():
    from delora_amd.data.feed import DevicePrefetcher
    dev = _dev()
    batches = [[{"scan_1": torch.randn(1, 3, 1000) + i, "dataset": "kitti"}] for i in range(5)]
    ref = [b[0]["scan_1"].clone() for b in batches]
    got = list(DevicePrefetcher(batches, dev))
    torch.cuda.synchronize()
    assert all(g[0]["scan_1"].is_cuda and torch.equal(g[0]["scan_1"].cpu(), r) for g, r in zip(got, ref))


def _ddp_gpu_worker(rank, world, port, ret):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from delora_amd.deploy.trainer import Trainer
        dev = torch.device("cuda:0")
        g, gm = util.load_golden("step_b2"), util.load_golden("model_small")
        cfg = _small_model_cfg(gm, int(g["H"]), int(g["W"]), batch_size=1)
        sample = {**{k: torch.from_numpy(g[f"s{rank}::{k}"]).to(dev) for k in ("scan_1", "scan_2", "normal_list_1", "normal_list_2")},
                  "dataset": "kitti"}
        tr = Trainer(cfg, dataset=util.ListDataset([sample]))
        tr.raw_model.load_state_dict(_state_dict(gm, dev))
        assert tr.world_size == 2 and isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
        ep = tr.new_epoch_losses()
        tr.optimizer.zero_grad()
        ep, T = tr.step(preprocessed_dicts=[dict(sample)], epoch_losses=ep)
        torch.cuda.synchronize()
        ret[rank] = {"T": T.detach().cpu(), "loss": float(ep["loss_epoch"]),
                     "gn": {k: float(p.grad.double().norm()) for k, p in tr.raw_model.named_parameters()}}
    finally:
        torch.distributed.destroy_process_group()


DDP_GRAD_BOUND = 2e-5          # measured 1.2e-6 (round 6)


def test_data_parallel_step_on_gpu_matches_reference_batch():
    """Two DDP ranks (one GPU, gloo transport; RCCL itself needs >1 GPU) x B=1 through the HIP path reproduce the
    reference's single-process B=2 step: poses, loss, averaged gradient norms."""
    import torch.multiprocessing as mp
    _dev()
    g = util.load_golden("step_b2")
    ret = mp.Manager().dict()
    mp.spawn(_ddp_gpu_worker, args=(2, 29650 + (os.getpid() % 300), ret), nprocs=2, join=True)
    worst, name = 0.0, ""
    for r in (0, 1):
        assert np.allclose(ret[r]["T"][0].numpy(), g["T"][r], rtol=REL, atol=REL * np.abs(g["T"]).max())
        for k, v in ret[r]["gn"].items():
            e = abs(v - float(g["gradnorm::" + k])) / max(float(g["gradnorm::" + k]), 1e-12)
            if e > worst:
                worst, name = e, k
    # (review, round 5: this bound was 3e-3; it is now ten times the largest deviation ever measured, like every other gradient bound.  The
    # narrow network of this fixture runs its convolutions on the library (MIOpen picks an algorithm per BATCH SIZE: one sample per rank here,
    # two in the reference's step), which is where the deviation comes from -- the full-width HIP trunk is held to 1e-4 in
    # test_step_matches_the_references_own_trainer_step.)
    util.measured(f"two DDP ranks x B=1 on the GPU vs the reference's B=2 step: worst relative gradient-norm deviation ({name})", worst, bound=DDP_GRAD_BOUND)
    assert np.isclose(ret[0]["loss"] + ret[1]["loss"], g["ep::loss_epoch"], rtol=REL)


def _rccl_worker(rank, world, port, ret):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.distributed.init_process_group(backend="nccl", device_id=dev)      # exactly bench.py's call
    try:
        from delora_amd.models import model as model_module
        gm = util.load_golden("model_small")
        cfg = _small_model_cfg(gm, 16, 128)
        net = model_module.OdometryModel(config=cfg).to(dev)
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], gradient_as_bucket_view=True, bucket_cap_mb=25)
        x = torch.randn(2, 8, 16, 128, device=dev)
        t, q = ddp(x)
        (t.square().sum() + q.square().sum()).backward()
        v = torch.ones(4, device=dev)
        torch.distributed.all_reduce(v)
        torch.distributed.barrier()
        torch.cuda.synchronize()
        ret[rank] = {"sum": float(v.sum()), "grads": all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())}
    finally:
        torch.distributed.destroy_process_group()


def test_rccl_backend_initialises_and_reduces_on_one_rank():
    """backend="nccl" (RCCL) with the init call and the DDP arguments bench.py / Trainer use, on the one GPU a test box
    has: communicator creation, DDP's bucketed all-reduce hooks in backward, an explicit all_reduce and a barrier."""
    import torch.multiprocessing as mp
    _dev()
    ret = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, 29350 + (os.getpid() % 300), ret), nprocs=1, join=True)
    assert ret[0]["sum"] == 4.0 and ret[0]["grads"]


def test_graphed_step_equals_eager_step():
    """The captured HIP graph of the whole training step replays to the same losses and weights as eager execution."""
    from delora_amd.data.dataset import SyntheticPairDataset
    from delora_amd.deploy.graph_step import GraphedStep
    from delora_amd.deploy.trainer import Trainer
    dev = _dev()

    def make():
        cfg = util.repo_config(32, 256, device="cuda:0", factor_fewer_resnet_channels=4, resnet_outputs=128,
                               unsupervised_at_start=True, inference_only=False, batch_size=2, learning_rate=1e-4)
        ds = SyntheticPairDataset(cfg, "kitti", 2, rings=32, azimuth_steps=300)
        torch.manual_seed(7)
        tr = Trainer(cfg, dataset=ds)
        return tr, tr.to_device([ds[0], ds[1]])

    tr_e, batch = make()
    eager = []
    for _ in range(4):
        tr_e.optimizer.zero_grad(set_to_none=True)
        ep, _ = tr_e.step(preprocessed_dicts=[dict(b) for b in batch], epoch_losses=tr_e.new_epoch_losses())
        eager.append(float(ep["loss_epoch"]))
    tr_g, batch_g = make()
    gs = GraphedStep(tr_g, batch_g, warmup=3)               # 3 eager warm-up steps whose updates are rolled back + 1 capture pass (which does not execute)
    assert gs.captured
    got = []
    for _ in range(4):
        ep, _ = gs()
        got.append(float(ep["loss_epoch"]))
    # two independent runs: MIOpen's split-K weight-gradient kernels accumulate with atomics, so trajectories agree to
    # fp32 noise, not bitwise
    assert np.allclose(got, eager, rtol=1e-3), (got, eager)
    for (k, a), (_, b) in zip(tr_g.raw_model.state_dict().items(), tr_e.raw_model.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-2, atol=2e-4), k


def test_graphed_step_equals_eager_step_on_the_hip_trunk():
    """The same comparison where it can be tight (review, round 5): the FULL-width network on a 16x512 pair runs on this library's own
    kernels -- fixed-order partial sums, no atomics -- so four replayed steps must reproduce four eager steps to fp32 rounding of the
    capturable Adam update (its bias corrections are computed on the device: an ulp per step), not to 1e-3."""
    from delora_amd.data.dataset import SyntheticPairDataset
    from delora_amd.deploy.graph_step import GraphedStep
    from delora_amd.deploy.trainer import Trainer
    _dev()

    def make():
        cfg = util.repo_config(16, 512, device="cuda:0", unsupervised_at_start=True, inference_only=False, batch_size=2, learning_rate=1e-5)
        ds = SyntheticPairDataset(cfg, "kitti", 2, rings=16, azimuth_steps=600)
        torch.manual_seed(7)
        tr = Trainer(cfg, dataset=ds)
        assert tr.raw_model.resnet.hip_path_takes(16, 512, batch=2)
        return tr, tr.to_device([ds[0], ds[1]])

    tr_e, batch = make()
    eager = []
    for _ in range(4):
        tr_e.optimizer.zero_grad(set_to_none=True)
        ep, _ = tr_e.step(preprocessed_dicts=[dict(b) for b in batch], epoch_losses=tr_e.new_epoch_losses())
        eager.append(float(ep["loss_epoch"]))
    tr_g, batch_g = make()
    gs = GraphedStep(tr_g, batch_g, warmup=3)
    assert gs.captured
    got = []
    for _ in range(4):
        ep, _ = gs()
        got.append(float(ep["loss_epoch"]))
    util.measured("graph replay vs eager on the HIP trunk (16x512, full width): worst relative loss difference over four steps",
                  float(np.max(np.abs(np.array(got) - np.array(eager)) / np.abs(np.array(eager)))), bound=2e-6)
    worst = max(float((a - b).abs().max()) for a, b in zip(tr_g.raw_model.state_dict().values(), tr_e.raw_model.state_dict().values()))
    util.measured("graph replay vs eager on the HIP trunk: largest weight difference after four steps (lr 1e-5: a step moves a weight by <= 1e-5)",
                  worst, bound=2e-7)


def test_mixed_sensor_batch_on_gpu():
    """Mixed kitti 16x128 / darpa 16x96 batch through the HIP path == the two samples run on their own."""
    from tests.test_host_logic import _mixed_setup, _terms_of
    _dev()
    cfg, sd, samples = _mixed_setup("cuda:0")
    terms, T, losses = _terms_of(cfg, sd, samples, None, device="cuda:0")
    for j, smp in enumerate(samples):
        t1, T1, _ = _terms_of(cfg, sd, [smp], None, device="cuda:0")
        assert torch.allclose(terms[j], t1[0], rtol=1e-5, atol=1e-7) and torch.allclose(T[j], T1[0], atol=1e-6)
    assert np.isfinite(float(losses["loss_pc"]))


def test_offline_preprocessing_writes_the_training_format(tmp_path):
    """Raw KITTI .bin scans -> Preprocesser (dl_project + dl_normals) -> scans/normals npy lists that (i) match the oracle's
    projection + normals of the same scan and (ii) load through PreprocessedPointCloudDataset."""
    from delora_amd.data import synthetic
    from delora_amd.data.dataset import PreprocessedPointCloudDataset
    from delora_amd.preprocessing.preprocesser import Preprocesser
    _dev()
    cfg = util.repo_config(16, 128, device="cuda:0")
    cfg["kitti"].update(horizontal_cells_preprocessing=160, data_path=str(tmp_path / "raw"), preprocessed_path=str(tmp_path / "pre"),
                        data_identifiers=[4])
    os.makedirs(tmp_path / "raw" / "04" / "velodyne")
    scans = [synthetic.make_pair(900 + i, rings=16, azimuth_steps=200)[0] for i in range(3)]
    for i, sc in enumerate(scans):
        np.concatenate([sc.T, np.ones((sc.shape[1], 1), np.float32)], axis=1).astype(np.float32).tofile(
            str(tmp_path / "raw" / "04" / "velodyne" / f"{i:06d}.bin"))
    Preprocesser(cfg).preprocess_data()
    assert cfg["kitti"]["horizontal_cells"] == 160
    o_sensor = util.oracle_sensor(16, 160, cfg["kitti"]["vertical_field_of_view"], cfg["horizontal_field_of_view"])
    for i, sc in enumerate(scans):
        pts = np.load(str(tmp_path / "pre" / "04" / "scans" / f"{i:06d}.npy"))
        nrm = np.load(str(tmp_path / "pre" / "04" / "normals" / f"{i:06d}.npy"))
        img, _, _, _, _ = orc.project_to_img(torch.from_numpy(sc).view(1, 3, -1), o_sensor)
        on, oh, op = orc.compute_normal_vectors(img.clone(), o_sensor)
        if util.ambiguity_mask(sc, o_sensor).sum() == 0 or pts.shape == tuple(op.shape):
            assert pts.shape == tuple(op.shape) and np.array_equal(pts, op.numpy())
            both = (np.abs(nrm).sum(1) > 0) & oh.numpy()
            a, b = nrm[both].astype(np.float64), on.numpy()[both].astype(np.float64)
            ang = np.arctan2(np.linalg.norm(np.cross(a, b), axis=1), np.sum(a * b, axis=1))
            assert np.median(ang) < 1e-5 and np.mean(ang > 5e-3) < 5e-3
    cfg["kitti"]["horizontal_cells"] = 128
    ds = PreprocessedPointCloudDataset(cfg)
    assert len(ds) == 2 and ds[0]["scan_1"].shape[1] == 3 and ds[0]["normal_list_2"].shape == ds[0]["scan_2"].shape


def test_scan_to_scan_odometry_on_gpu_matches_the_cpu_core():
    """The inference core (delora_amd/ros_utils/odometry.py) with the HIP projection on the GPU against the same core on
    the CPU with the oracle's projection injected: same weights, same three scans, same poses."""
    from delora_amd.data import synthetic
    from delora_amd.models import model as model_module
    from delora_amd.ros_utils import odometry
    dev = _dev()
    gm = util.load_golden("model_small")
    cfg_gpu = _small_model_cfg(gm, 16, 128)
    cfg_cpu = dict(cfg_gpu, device=torch.device("cpu"))
    for c in (cfg_gpu, cfg_cpu):
        c["integrate_odometry"] = True
    net_gpu = model_module.OdometryModel(config=cfg_gpu).to(dev)
    net_gpu.load_state_dict(_state_dict(gm, dev))
    net_cpu = model_module.OdometryModel(config=cfg_cpu)
    net_cpu.load_state_dict(_state_dict(gm, torch.device("cpu")))
    o_sensor = util.oracle_sensor(16, 128, cfg_cpu["kitti"]["vertical_field_of_view"], cfg_cpu["horizontal_field_of_view"])

    def oracle_pair(previous, current, sensor):
        return torch.cat([orc.project_to_img(c.cpu(), o_sensor)[0][:, :4] for c in (previous, current)], dim=0)

    gpu = odometry.ScanToScanOdometry(cfg_gpu, model=net_gpu)
    cpu = odometry.ScanToScanOdometry(cfg_cpu, model=net_cpu, project_pair=oracle_pair)
    scans = [synthetic.make_pair(300 + i, rings=16, azimuth_steps=140)[i % 2][None] for i in range(3)]
    assert gpu.push(scans[0]) is None and cpu.push(scans[0]) is None
    for k in (1, 2):
        a, b = gpu.push(scans[k]), cpu.push(scans[k])
        assert np.allclose(a["transformation"], b["transformation"], rtol=REL, atol=REL)
        assert np.allclose(a["T_0_t"], b["T_0_t"], rtol=REL, atol=REL)


def test_tester_on_gpu_path_matches_oracle_backend_and_pose_integration(tmp_path):
    """Reference src/deploy/tester.py:38-109 + src/utility/poses.py:11-74 on the HIP path: checkpoint -> Tester.test() over a
    3-scan synthetic sequence on the GPU (projection / normals / search / loss kernels, batch size 1, losses evaluated) ->
    transformations equal to the same Tester run on the CPU with the oracle geometry backend; pose files = compute_poses."""
    from delora_amd.data import synthetic
    from delora_amd.deploy.tester import Tester
    from delora_amd.deploy.trainer import Trainer
    from delora_amd.utility import poses
    dev = _dev()
    gm = util.load_golden("model_small")
    H, W = 16, 128
    cfg = _small_model_cfg(gm, H, W, batch_size=1)
    tr = Trainer(cfg, dataset=util.ListDataset([]))
    tr.raw_model.load_state_dict(_state_dict(gm, dev))
    ck = str(tmp_path / "m.pth")
    tr.save_checkpoint(ck, 0, 0.0)
    scans = [synthetic.make_pair(700 + i, rings=16, azimuth_steps=160)[0] for i in range(3)]
    samples = []
    for j in range(2):                                          # pairs (scan j, scan j+1) of one sequence
        samples.append({"index": j, "index_dataset": 0, "index_sequence": 0, "index_scan": j, "dataset": "kitti",
                        "scan_1": torch.from_numpy(scans[j]).unsqueeze(0), "scan_2": torch.from_numpy(scans[j + 1]).unsqueeze(0),
                        "normal_list_1": None, "normal_list_2": None})
    results = {}
    for name, device, backend in (("gpu", "cuda:0", None), ("cpu", "cpu", util.OracleStepGeometry())):
        c = _small_model_cfg(gm, H, W, batch_size=1)
        c.update(device=torch.device(device), checkpoint=ck, inference_only=False, output_dir=str(tmp_path / name), run_name="t",
                 mode="testing")
        os.makedirs(str(tmp_path / name), exist_ok=True)
        c["kitti"]["data_identifiers"] = [9]
        te = Tester(c, dataset=util.ListDataset([dict(s) for s in samples]), geometry_backend=backend)
        summary = te.test()
        assert len(te.written) == 1
        results[name] = (np.load(te.written[0]["transformations"]), np.load(te.written[0]["poses"]),
                         open(te.written[0]["poses_text"]).read(), summary)
    Tg, Pg, txt, sg = results["gpu"]
    Tc, Pc, _, sc = results["cpu"]
    assert Tg.shape == (2, 1, 4, 4)
    util.measured("Tester: max |T_gpu - T_cpu(oracle backend)|", float(np.abs(Tg - Tc).max()), bound=REL)
    assert np.array_equal(Pg, poses.compute_poses(list(Tg)))
    assert len(txt.strip().splitlines()) == 3
    for k in ("loss_point_cloud_epoch", "loss_po2pl_epoch", "loss_pl2pl_epoch"):
        assert np.isclose(sg[k], sc[k], rtol=5e-3), (k, sg[k], sc[k])    # images differ by the atan2-ambiguous pixels only
