"""Parity of the HIP geometry kernels (through the C ABI) against the oracle and the golden vectors.

Bars: pixel indices / correspondences bit-exact (projection: outside the documented atan2 ambiguity mask),
fp32 values exact where the same arithmetic is specified, loss terms and gradients within 1e-4 relative.
"""
import numpy as np
import pytest
import torch

from tests import util
from tests.util import orc

pytestmark = pytest.mark.gpu

REL = 1e-4       # tolerance of the north star for floating-point results


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _geo():
    from delora_amd import geometry
    return geometry


def gpu_sensor(H, W, vfov, hfov):
    return _geo().Sensor(int(H), int(W), [float(v) for v in vfov], [float(h) for h in hfov])


def run_project(scans, sensor, want_uv=True):
    """scans: list of [C,N] numpy arrays -> dict of CPU numpy outputs (batched in one launch)."""
    dev = _dev()
    C = scans[0].shape[0]
    pts = torch.from_numpy(np.concatenate(scans, axis=1)).to(dev)
    lens = [s.shape[1] for s in scans]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    if pts.shape[1] == 0:
        pts = torch.zeros((C, 1), device=dev)[:, :0]
    out = _geo().project(pts, offs, max(lens), sensor, want_uv=want_uv)
    torch.cuda.synchronize()
    return out


def check_projection(scan, out, s, o_sensor, expect_pix2pt=None, expect_image=None):
    """The projection contract for scan ``s`` of a batched result."""
    ref_pix, ref_u, ref_v = util.reference_pixels(scan, o_sensor)
    n0 = 0 if s == 0 else None
    H, W = o_sensor.H, o_sensor.W
    pix2pt = out["pix2pt"][s].cpu().numpy()
    image = out["image4"][s].cpu().numpy()
    N = scan.shape[1]
    if out["uv"] is not None and n0 is not None and N:
        uv = out["uv"].cpu().numpy()[:, :N]
        assert np.nanmax(np.abs(uv[0] - ref_u)) < 2e-3 and np.nanmax(np.abs(uv[1] - ref_v)) < 2e-3
        ru, rv = np.rint(uv[0]), np.rint(uv[1])
        inside = (ru <= W - 1) & (ru >= 0) & (rv <= H - 1) & (rv >= 0)
        gpu_pix = np.where(inside, rv.astype(np.int64) * W + ru.astype(np.int64), -1)
        differ = gpu_pix != ref_pix
        amb = util.ambiguity_mask(scan, o_sensor)
        assert not np.any(differ & ~amb), "pixel index differs outside the ambiguity mask"
        # how many points actually moved (SURVEY.md's probe: ~5e-5 of the points; the mask itself covers ~4e-3)
        util.measured(f"projection: points in another pixel than the torch-CPU reference (N={N}, H={H}, W={W})",
                      int(differ.sum()), bound=max(2, int(np.ceil(2e-4 * N))))
    else:
        differ = np.zeros(N, dtype=bool)
        gpu_pix = ref_pix
    # expected map from the reference semantics: nearest range per pixel, ties to the lower index
    rng = torch.norm(torch.from_numpy(np.ascontiguousarray(scan[:3])).view(1, 3, -1), dim=1)[0].numpy()
    exp = -np.ones(H * W, dtype=np.int64)
    order = np.lexsort((np.arange(N), rng))
    keep = ref_pix[order] >= 0
    po, oo = ref_pix[order][keep], order[keep]
    _, first = np.unique(po, return_index=True)
    exp[po[first]] = oo[first]
    touched = np.zeros(H * W, dtype=bool)
    for arr in (gpu_pix[differ], ref_pix[differ]):
        touched[arr[arr >= 0]] = True
    ok = ~touched
    assert np.array_equal(pix2pt.reshape(-1)[ok], exp[ok]), "pixel -> point map mismatch"
    if expect_pix2pt is not None:
        # the golden map was produced on another CPU (different Sleef path): compare away from ambiguous points
        clean = ~util.tainted_pixels(scan, o_sensor)
        exp2 = np.asarray(expect_pix2pt)
        bad = np.argwhere((pix2pt != exp2) & clean)
        for r, c in bad:
            # the reference's argsort is not stable: two points of one pixel with bit-equal range are a legitimate tie
            a, b = int(pix2pt[r, c]), int(exp2[r, c])
            assert a >= 0 and b >= 0 and rng[a] == rng[b] and a < b, f"pixel ({r},{c}): {a} vs {b}"
        assert len(bad) <= 4
    # image content: xyz are copies, range follows torch.norm's rounding
    occ = exp >= 0
    sel = ok & occ
    for c in range(3):
        assert np.array_equal(image[c].reshape(-1)[sel], scan[c][exp[sel]])
    assert np.array_equal(image[3].reshape(-1)[sel], rng[exp[sel]])
    assert np.all(image.reshape(4, -1)[:, ok & ~occ] == 0)
    assert int(out["kept"][s].item()) == int((pix2pt >= 0).sum())
    if expect_image is not None:
        clean = ~util.tainted_pixels(scan, o_sensor)
        assert np.array_equal(image[:, clean], expect_image[:, clean])
    return differ


@pytest.mark.parametrize("name", ["small", "small_c6", "mid", "edge", "all_outside"])
def test_projection_golden(name):
    g = util.load_golden("proj_" + name)
    o_sensor = util.oracle_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
    sensor = gpu_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
    scan = g["scan"]
    out = run_project([scan], sensor)
    H, W = o_sensor.H, o_sensor.W
    exp_map = -np.ones((H, W), dtype=np.int64)
    pix = g["pix"][0]
    exp_map[pix[:, 0], pix[:, 1]] = g["idx"]
    C = scan.shape[0]
    ref_img = g["image"][0]                                   # [C+1,H,W], last = range
    exp_img4 = np.concatenate([ref_img[:3], ref_img[C:C + 1]], axis=0)
    differ = check_projection(scan, out, 0, o_sensor, exp_map, exp_img4)
    if C > 3:
        clean = ~util.tainted_pixels(scan, o_sensor)
        assert np.array_equal(out["aux"][0].cpu().numpy()[:, clean], ref_img[3:C][:, clean])


def test_projection_batched_ragged_and_deterministic():
    """Several scans of different length (one empty) in one launch; results independent of batching and run."""
    from delora_amd.data import synthetic
    vf, hf = util.kitti_fov()
    o_sensor, sensor = util.oracle_sensor(32, 256, vf, hf), gpu_sensor(32, 256, vf, hf)
    scans = [synthetic.make_pair(100 + i, rings=32, azimuth_steps=300 + 17 * i)[0] for i in range(3)]
    scans.insert(1, np.zeros((3, 0), dtype=np.float32))
    out = run_project(scans, sensor, want_uv=False)
    for s, scan in enumerate(scans):
        check_projection(scan, out, s, o_sensor)
    again = run_project(scans, sensor, want_uv=False)
    for k in ("image4", "pix2pt", "kept"):
        assert torch.equal(out[k], again[k])
    solo = run_project([scans[2]], sensor, want_uv=False)
    assert torch.equal(solo["pix2pt"][0], out["pix2pt"][2]) and torch.equal(solo["image4"][0], out["image4"][2])


def test_projection_full_size_digest():
    """64x2048 KITTI-shaped scan: pixel->point map equals the reference's (committed as int32 digest)."""
    from delora_amd.data import synthetic
    g = util.load_golden("proj_full_digest")
    import hashlib
    s1 = synthetic.portable_cloud(int(g["seed"]), int(g["N"]))
    assert hashlib.sha256(s1.tobytes()).hexdigest() == str(g["scan_sha"]), "portable_cloud is not bit-reproducible here"
    o_sensor = util.oracle_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
    sensor = gpu_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
    out = run_project([s1], sensor)
    differ = check_projection(s1, out, 0, o_sensor, g["pix2pt"].astype(np.int64))
    assert set(np.nonzero(differ)[0]).issubset(set(g["ambiguous_idx"].tolist()))
    if True:
        assert abs(float(out["image4"][0, 3].double().sum().item()) - float(g["range_sum"])) < 1e-4 * float(g["range_sum"])


def test_projection_idempotent_full_size():
    """Projecting the kept points again reproduces the image (size-independent property, batch of 8)."""
    from delora_amd.data import synthetic
    vf, hf = util.kitti_fov()
    sensor = gpu_sensor(64, 2048, vf, hf)
    scans = [synthetic.make_pair(300 + i)[i % 2] for i in range(8)]
    out = run_project(scans, sensor, want_uv=False)
    relists = []
    for s in range(8):
        img = out["image4"][s].cpu().numpy().reshape(4, -1)
        occ = out["pix2pt"][s].cpu().numpy().reshape(-1) >= 0
        relists.append(np.ascontiguousarray(img[:3, occ]))
    out2 = run_project(relists, sensor, want_uv=False)
    assert torch.equal(out["image4"], out2["image4"])
    assert torch.equal(out["kept"], out2["kept"])


def test_projection_fast_path_gives_the_pixels_of_the_exact_path():
    """dl_project takes atan2f + a distance-to-rounding-boundary test per point and falls back to the fp64 evaluation near
    k + 1/2; with want_uv it evaluates every point in fp64.  Both must give the same image -- on 4 M points spread over
    magnitudes from 1e-3 to 1e3 m, on points planted within a few ulp of pixel boundaries, and on the coordinate axes."""
    rng = np.random.default_rng(77)
    H, W = 64, 2048
    vf, hf = util.kitti_fov()
    sensor = gpu_sensor(H, W, vf, hf)
    n = 1_000_000
    scans = []
    for scale in (1e-3, 1.0, 30.0, 1e3):
        p = rng.normal(size=(3, n)).astype(np.float32) * np.float32(scale)
        p[2] *= np.float32(0.15)                       # keep most of them inside the vertical field of view
        scans.append(p)
    # planted: azimuths exactly on / next to pixel boundaries (k + 1/2) * 2 pi / (W - 1), all elevations likewise
    k = np.arange(-2, W + 2, dtype=np.float64) + 0.5
    az = hf[0] + k * (hf[1] - hf[0]) / (W - 1)
    j = np.arange(-2, H + 2, dtype=np.float64) + 0.5
    el = vf[0] + j * (vf[1] - vf[0]) / (H - 1)
    azg, elg = np.meshgrid(az, el)
    planted = []
    for r in (0.7, 9.3, 61.0):
        for d in (-2e-7, 0.0, 2e-7):
            a_, e_ = azg.ravel() * (1 + d), elg.ravel() * (1 + d)
            planted.append(np.stack([r * np.cos(e_) * np.cos(a_), r * np.cos(e_) * np.sin(a_), r * np.sin(e_)]).astype(np.float32))
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0], [1e-30, 1e-30, 0], [-5, 1e-38, 0.1],
                     [-5, -1e-38, 0.1]], dtype=np.float32).T
    scans.append(np.concatenate(planted + [axes], axis=1))
    fast = run_project(scans, sensor, want_uv=False)
    exact = run_project(scans, sensor, want_uv=True)
    for key in ("image4", "pix2pt", "kept"):
        assert torch.equal(fast[key], exact[key]), key
    util.measured(f"projection fast path: pixels that differ from the all-fp64 path ({sum(s.shape[1] for s in scans)} points)",
                  int((fast["pix2pt"] != exact["pix2pt"]).sum()), bound=0)


# ------------------------------------------------------------------------------------------ normals
def _angle(a, b):
    # atan2(|a x b|, a.b): arccos of the dot product has a noise floor of sqrt(2 eps32) ~ 3e-4 rad on fp32 unit vectors
    return np.arctan2(np.linalg.norm(np.cross(a, b), axis=1), np.sum(a * b, axis=1))


def check_normals(image, got, ref_normals, ref_has, ref_v, ref_u, eigenvalues, pts):
    """Compare normals at the reference's valid pixels, conditioning-aware."""
    g = got[:, ref_v, ref_u].T                                  # [M,3]
    got_has = np.any(g != 0, axis=1)
    mism = got_has != ref_has
    util.measured(f"normals: has-normal mask differs from the reference (pixels of {len(mism)})", int(mism.sum()),
                  bound=max(2, int(np.ceil(5e-4 * len(mism)))))
    both = got_has & ref_has
    lam = eigenvalues[both].astype(np.float64)
    gap = (lam[:, 1] - lam[:, 0]) / np.maximum(lam[:, 2], 1e-30)
    p = pts[both].astype(np.float64)
    graze = np.abs(np.sum(ref_normals[both] * p, axis=1)) / np.linalg.norm(p, axis=1) < 1e-3
    a, b = g[both].astype(np.float64), ref_normals[both].astype(np.float64)
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5)
    ang = _angle(a, b)
    ang_unsigned = np.minimum(ang, np.pi - ang)
    ang = np.where(graze, ang_unsigned, ang)
    well = gap > 1e-3
    # fp32 LAPACK error of the reference scales like eps32 * lambda_max / gap
    bound = 2e-4 + 50 * 6e-8 / np.maximum(gap, 1e-12)
    frac_bad = np.mean(ang[well] > bound[well])
    util.measured(f"normals: fraction of well-conditioned normals outside the conditioning bound (M={len(mism)})", frac_bad, bound=1e-3)
    util.measured(f"normals: median angle to the reference [rad] (M={len(mism)})", float(np.median(ang)), bound=1e-5)
    # pixels that are not valid in the reference carry no normal
    mask = np.ones(got.shape[1:], dtype=bool)
    mask[ref_v, ref_u] = False
    assert np.all(got[:, mask] == 0)
    return ang


@pytest.mark.parametrize("name", ["small", "mid"])
def test_normals_golden(name):
    g = util.load_golden("normals_" + name)
    dev = _dev()
    image = torch.from_numpy(g["image"]).to(dev)                # [1,4,H,W]
    a, b = int(g["side"][0] / 2), int(g["side"][1] / 2)
    got = _geo().normals(image, a, b, float(g["epsilon_range"]), int(g["min_neighbors"]))[0].cpu().numpy()
    check_normals(g["image"], got, g["normals"], g["has"], g["v"], g["u"], g["eigenvalues"], g["points"])


def test_normals_vs_oracle_batched_128_rings():
    """Ouster-style 128-row images, batch of 3 (BASELINE config 4 shape class, reduced width for oracle time)."""
    from delora_amd.data import synthetic
    dev = _dev()
    vf = (-22.5 * np.pi / 180.0, 22.5 * np.pi / 180.0)
    hf = util.kitti_fov()[1]
    sensor, o_sensor = gpu_sensor(128, 256, vf, hf), util.oracle_sensor(128, 256, vf, hf)
    scans = [synthetic.make_pair(500 + i, rings=128, azimuth_steps=290, vfov_deg=(-22.5, 22.5))[0] for i in range(3)]
    out = run_project(scans, sensor, want_uv=False)
    got = _geo().normals(out["image4"]).cpu().numpy()
    for s in range(3):
        img = out["image4"][s:s + 1].cpu()
        n, has, pts, aux = orc.compute_normal_vectors(img.clone(), o_sensor, return_aux=True)
        check_normals(None, got[s], n.numpy(), has.numpy(), aux["v"].numpy(), aux["u"].numpy(),
                      aux["eigenvalues"].numpy(), pts.numpy())


# ------------------------------------------------------------------------------------ correspondences
def _pair_images(seed, H, W, rings, az, vfov_deg=(-24.5, 2.0), online_normals=True):
    from delora_amd.data import synthetic
    vf = (vfov_deg[0] * np.pi / 180.0, vfov_deg[1] * np.pi / 180.0)
    hf = util.kitti_fov()[1]
    sensor = gpu_sensor(H, W, vf, hf)
    s1, s2, T = synthetic.make_pair(seed, rings=rings, azimuth_steps=az, vfov_deg=vfov_deg)
    out = run_project([s1, s2], sensor, want_uv=False)
    nrm = _geo().normals(out["image4"])
    return sensor, out["image4"], nrm, T


def _random_T(rng, scale_t=1.0):
    q = torch.tensor(rng.normal(size=(1, 4)), dtype=torch.float32)
    t = torch.tensor(rng.normal(0, scale_t, size=(1, 3)), dtype=torch.float32)
    return orc.transformation_matrix(t, q)


def _q_slack(q):
    """Two candidates whose distances differ by less than the fp32 rounding of the transformed point q = R p + t
    (a few ulp of |q|; the GPU forms q with an fmaf chain, torch's CPU bmm with its own order) are a tie."""
    return 8 * 6e-8 * q.norm(dim=0) + 1e-12


def oracle_nn_pixels(tgt_img, src_img, src_nrm, T, need_wo):
    """Oracle correspondences expressed as target pixel ids per source pixel (-1 where none is requested)."""
    tp, _, tpix = util.lists_from_images(tgt_img, torch.zeros(3, *tgt_img.shape[1:]))
    sp, sn, spix = util.lists_from_images(src_img, src_nrm)
    has = (sn[0] != 0).any(dim=0)
    sel = torch.ones_like(has) if need_wo else has
    q = orc.transform_points(T, sp[:, :, sel])
    nn = orc.nearest_target_indices(tp, q)
    out = -torch.ones(src_img.shape[1] * src_img.shape[2], dtype=torch.long)
    out[spix[sel]] = tpix[nn]
    return out, q, tp, tpix


@pytest.mark.parametrize("case", ["identity", "true", "small_error", "random", "random_far"])
@pytest.mark.parametrize("shape", [(16, 128, 16, 160), (64, 512, 64, 600), (64, 2048, 64, 2250)])
def test_nn_matches_kdtree(case, shape):
    """Exact k=1 correspondences against cKDTree in five pose regimes, up to the FULL image size (64x2048: 131 k target points,
    where the hard regimes -- random rotation, 30 m translation -- run the seed scan and the whole-image tile walk)."""
    H, W, rings, az = shape
    sensor, img, nrm, T_true = _pair_images(77, H, W, rings, az)
    rng = np.random.default_rng(3)
    if case == "identity":
        T = torch.eye(4).view(1, 4, 4)
    elif case == "true":
        T = torch.from_numpy(T_true).view(1, 4, 4)
    elif case == "small_error":
        T = torch.from_numpy(T_true).view(1, 4, 4).clone()
        T[0, :3, 3] += torch.tensor([0.15, -0.1, 0.05])
    elif case == "random":
        T = _random_T(rng)
    else:
        T = _random_T(rng, scale_t=30.0)
    for need_wo in (False, True):
        nn, vis, _ = _geo().nn_correspond(img[1:2], nrm[1:2], _geo().pack_image(img[0:1]), None, T.to(img.device), sensor, need_without_normals=need_wo)
        torch.cuda.synchronize()
        exp, q, tp, tpix = oracle_nn_pixels(img[0].cpu(), img[1].cpu(), nrm[1].cpu(), T, need_wo)
        got = nn[0].reshape(-1).cpu().long()
        if not torch.equal(got, exp):
            # only exact distance ties may differ: compare the fp64 distances of the two answers
            bad = torch.nonzero(got != exp).reshape(-1)
            assert ((got[bad] >= 0) & (exp[bad] >= 0)).all()
            tflat = img[0, :3].reshape(3, -1).cpu().double()
            sflat = img[1, :3].reshape(3, -1).cpu()
            qq = orc.transform_points(T, sflat[:, bad].view(1, 3, -1))[0].double()
            d_got = (qq - tflat[:, got[bad]]).norm(dim=0)
            d_exp = (qq - tflat[:, exp[bad]]).norm(dim=0)
            assert torch.all(d_got <= d_exp + _q_slack(qq)), f"{len(bad)} non-tie mismatches"
            assert len(bad) <= 1e-4 * len(got) + 1
        sp, _, _ = util.lists_from_images(img[1].cpu(), nrm[1].cpu())
        assert int(vis[0].item()) == orc.visible_pixels(orc.transform_points(T, sp), util.oracle_sensor(H, W, sensor.vfov, sensor.hfov))


def test_nn_full_size_against_bruteforce_kernel():
    """64x2048, batch 2: the windowed search equals the exhaustive kernel (and the KD-tree on sample 0)."""
    sensor, img, nrm, T_true = _pair_images(2024, 64, 2048, 64, 2250)
    dev = img.device
    imgs_t = torch.stack([img[0], img[0]])
    imgs_s = torch.stack([img[1], img[1]])
    nrm_s = torch.stack([nrm[1], nrm[1]])
    T = torch.stack([torch.from_numpy(T_true), torch.eye(4)]).to(dev)
    nn, _, _ = _geo().nn_correspond(imgs_s, nrm_s, _geo().pack_image(imgs_t), None, T, sensor, need_without_normals=True)
    for b in range(2):
        tp, _, tpix = util.lists_from_images(imgs_t[b].cpu(), torch.zeros(3, 64, 2048))
        sp, _, spix = util.lists_from_images(imgs_s[b].cpu(), nrm_s[b].cpu())
        q = orc.transform_points(T[b:b + 1].cpu(), sp)
        bf = _geo().nn_bruteforce(q[0].to(dev), tp[0].to(dev)).cpu().long()
        got = nn[b].reshape(-1).cpu().long()[spix]
        exp = tpix[bf]
        if not torch.equal(got, exp):
            bad = torch.nonzero(got != exp).reshape(-1)
            tflat = imgs_t[b, :3].reshape(3, -1).cpu().double()
            d_got = (q[0][:, bad].double() - tflat[:, got[bad]]).norm(dim=0)
            d_exp = (q[0][:, bad].double() - tflat[:, exp[bad]]).norm(dim=0)
            assert torch.all(d_got <= d_exp + _q_slack(q[0][:, bad].double())), f"{len(bad)} non-tie mismatches"
            assert len(bad) <= 1e-4 * len(got) + 1
        if b == 0:
            kd = orc.nearest_target_indices(tp, q)
            assert (tpix[kd] != got).sum() <= 2          # exact ties only


def test_nn_packet_walk_batched_random_poses():
    """64x2048, batch 4, poses without a usable bound (random rotations, a 30 m translation, a tilted pose): most source tiles go through
    the PACKET walk of pass B (k_nn_packets: the 64 queries of a tile walk the pyramid together); indices against the exhaustive
    kernel, the matched points / normals planes against a gather of the target images."""
    sensor, img, nrm, T_true = _pair_images(4242, 64, 2048, 64, 2250)
    dev = img.device
    rng = np.random.default_rng(11)
    tilt = torch.from_numpy(T_true).view(1, 4, 4).clone().float()
    tilt[0, :3, :3] = orc.transformation_matrix(torch.zeros(1, 3), torch.tensor([[0.03, -0.04, 0.01, 1.0]]))[0, :3, :3] @ tilt[0, :3, :3]
    T = torch.cat([_random_T(rng), _random_T(rng, scale_t=30.0), _random_T(rng, scale_t=3.0), tilt]).float().to(dev)
    B = T.shape[0]
    imgs_t = img[0:1].expand(B, -1, -1, -1).contiguous()
    imgs_s = img[1:2].expand(B, -1, -1, -1).contiguous()
    nrm_s = nrm[1:2].expand(B, -1, -1, -1).contiguous()
    tpk, tnpk = _geo().pack_image(imgs_t), _geo().pack_image(nrm[0:1].expand(B, -1, -1, -1).contiguous())
    nn, _, match = _geo().nn_correspond(imgs_s, nrm_s, tpk, tnpk, T, sensor, need_without_normals=False)
    torch.cuda.synchronize()
    tflat = imgs_t[0, :3].reshape(3, -1)
    nflat = nrm[0].reshape(3, -1)
    for b in range(B):
        tp, _, tpix = util.lists_from_images(imgs_t[b].cpu(), torch.zeros(3, 64, 2048))
        sp, sn, spix = util.lists_from_images(imgs_s[b].cpu(), nrm_s[b].cpu())
        has = (sn[0] != 0).any(dim=0)
        q = orc.transform_points(T[b:b + 1].cpu(), sp[:, :, has])
        bf = _geo().nn_bruteforce(q[0].to(dev), tp[0].to(dev)).cpu().long()
        got = nn[b].reshape(-1).cpu().long()
        exp = -torch.ones_like(got)
        exp[spix[has]] = tpix[bf]
        if not torch.equal(got, exp):
            bad = torch.nonzero(got != exp).reshape(-1)
            assert ((got[bad] >= 0) & (exp[bad] >= 0)).all()
            qq = orc.transform_points(T[b:b + 1].cpu(), imgs_s[b, :3].reshape(3, -1).cpu()[:, bad].view(1, 3, -1))[0].double()
            d_got = (qq - tflat.cpu().double()[:, got[bad]]).norm(dim=0)
            d_exp = (qq - tflat.cpu().double()[:, exp[bad]]).norm(dim=0)
            assert torch.all(d_got <= d_exp + _q_slack(qq)), f"sample {b}: {len(bad)} non-tie mismatches"
            assert len(bad) <= 1e-4 * len(got) + 1
        idx = nn[b].reshape(-1).long()
        ok = idx >= 0
        m = match[b].reshape(6, -1)
        assert torch.equal(m[:3, ok], tflat[:, idx[ok]]) and torch.equal(m[3:, ok], nflat[:, idx[ok]])
        assert (m[:, ~ok] == 0).all()


def test_nn_windowed_packets_batch_of_tilted_poses(monkeypatch):
    """64x2048, batch 8, the true motion composed with a few degrees of roll / pitch error (a network in mid-training): thousands of
    source tiles whose bound windows are large -- pass B takes them as WINDOWED packets (k_nn_pass_b's first workgroup range) and skips
    their records in the lists.  Indices against the exhaustive kernel; the test reads the list counters to make sure the regime was met."""
    sensor, img, nrm, T_true = _pair_images(999, 64, 2048, 64, 2250)
    dev = img.device
    B = 8
    rng = np.random.default_rng(21)
    Ts = []
    for b in range(B):
        e = rng.normal(0, 0.04, size=3)
        q = torch.tensor([[e[0], e[1], 0.2 * e[2], 1.0]], dtype=torch.float32)
        dT = orc.transformation_matrix(torch.tensor(rng.normal(0, 0.2, size=(1, 3)), dtype=torch.float32), q)
        Ts.append(dT[0] @ torch.from_numpy(T_true).float())
    T = torch.stack(Ts).to(dev)
    imgs_t = img[0:1].expand(B, -1, -1, -1).contiguous()
    imgs_s = img[1:2].expand(B, -1, -1, -1).contiguous()
    nrm_s = nrm[1:2].expand(B, -1, -1, -1).contiguous()
    kept = {}
    real_empty = torch.empty

    def spy(*a, **k):
        t = real_empty(*a, **k)
        if k.get("dtype") == torch.int64:
            kept["ws"] = t
        return t
    monkeypatch.setattr(torch, "empty", spy)
    nn, _, _ = _geo().nn_correspond(imgs_s, nrm_s, _geo().pack_image(imgs_t), None, T, sensor, need_without_normals=True)
    monkeypatch.undo()
    torch.cuda.synchronize()
    counters = kept["ws"].view(torch.int32)[:6].tolist()
    assert counters[5] >= 4096, f"the poses of this test no longer produce windowed packets: counters {counters}"
    tflat = imgs_t[0, :3].reshape(3, -1).cpu().double()
    tp, _, tpix = util.lists_from_images(imgs_t[0].cpu(), torch.zeros(3, 64, 2048))
    sp, _, spix = util.lists_from_images(imgs_s[0].cpu(), nrm_s[0].cpu())
    for b in range(B):
        q = orc.transform_points(T[b:b + 1].cpu(), sp)
        bf = _geo().nn_bruteforce(q[0].to(dev), tp[0].to(dev)).cpu().long()
        got = nn[b].reshape(-1).cpu().long()[spix]
        exp = tpix[bf]
        if not torch.equal(got, exp):
            bad = torch.nonzero(got != exp).reshape(-1)
            d_got = (q[0][:, bad].double() - tflat[:, got[bad]]).norm(dim=0)
            d_exp = (q[0][:, bad].double() - tflat[:, exp[bad]]).norm(dim=0)
            assert torch.all(d_got <= d_exp + _q_slack(q[0][:, bad].double())), f"sample {b}: {len(bad)} non-tie mismatches"
            assert len(bad) <= 1e-4 * len(got) + 1


@pytest.mark.parametrize("case", ["random", "true"])
def test_nn_ragged_image_large_batch(case):
    """30x500 (neither a multiple of the 4x16 tiles), batch 48: the tile-shaped waves of pass A hang over the image's edges, and with
    random poses the batch has enough source tiles for the packet walk; indices against the exhaustive kernel."""
    H, W = 30, 500
    sensor, img, nrm, T_true = _pair_images(31, H, W, 30, 550)
    dev = img.device
    B = 48
    rng = np.random.default_rng(5)
    T = torch.cat([_random_T(rng) if case == "random" else torch.from_numpy(T_true).view(1, 4, 4).float() for _ in range(B)]).float().to(dev)
    imgs_t = img[0:1].expand(B, -1, -1, -1).contiguous()
    imgs_s = img[1:2].expand(B, -1, -1, -1).contiguous()
    nrm_s = nrm[1:2].expand(B, -1, -1, -1).contiguous()
    nn, _, _ = _geo().nn_correspond(imgs_s, nrm_s, _geo().pack_image(imgs_t), None, T, sensor, need_without_normals=True)
    torch.cuda.synchronize()
    tflat = imgs_t[0, :3].reshape(3, -1).cpu().double()
    tp, _, tpix = util.lists_from_images(imgs_t[0].cpu(), torch.zeros(3, H, W))
    sp, _, spix = util.lists_from_images(imgs_s[0].cpu(), nrm_s[0].cpu())
    for b in range(0, B, 7):
        q = orc.transform_points(T[b:b + 1].cpu(), sp)
        bf = _geo().nn_bruteforce(q[0].to(dev), tp[0].to(dev)).cpu().long()
        got = nn[b].reshape(-1).cpu().long()[spix]
        exp = tpix[bf]
        if not torch.equal(got, exp):
            bad = torch.nonzero(got != exp).reshape(-1)
            d_got = (q[0][:, bad].double() - tflat[:, got[bad]]).norm(dim=0)
            d_exp = (q[0][:, bad].double() - tflat[:, exp[bad]]).norm(dim=0)
            assert torch.all(d_got <= d_exp + _q_slack(q[0][:, bad].double())), f"sample {b}: {len(bad)} non-tie mismatches"
            assert len(bad) <= 1e-4 * len(got) + 1


def test_nn_empty_target_and_empty_source():
    vf, hf = util.kitti_fov()
    sensor = gpu_sensor(16, 128, vf, hf)
    dev = _dev()
    z = torch.zeros((1, 4, 16, 128), device=dev)
    _, img, nrm, _ = _pair_images(5, 16, 128, 16, 160)
    T = torch.eye(4, device=dev).view(1, 4, 4)
    nn, vis, _ = _geo().nn_correspond(img[1:2], nrm[1:2], _geo().pack_image(z), None, T, sensor)
    assert (nn == -1).all()
    nn, vis, _ = _geo().nn_correspond(z, z[:, :3], _geo().pack_image(img[0:1]), None, T, sensor)
    assert (nn == -1).all() and int(vis[0].item()) == 0


# ------------------------------------------------------------------------------------------- losses
def _flags(mode, p2p):
    G = _geo()
    f = G.LOSS_POINT_TO_PLANE | G.LOSS_PLANE_TO_PLANE
    if p2p:
        f |= G.LOSS_POINT_TO_POINT
    if mode == "linear":
        f |= G.LOSS_NORMAL_LINEAR
    return f


def _close(a, b, rel=REL, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b).max(), 1e-30)
    assert np.all(np.abs(a - b) <= rel * scale + 1e-9), f"{what}: {a} vs {b}"


@pytest.mark.parametrize("mode", ["squared", "linear"])
@pytest.mark.parametrize("p2p", [False, True])
def test_loss_golden_from_reference(mode, p2p):
    """Raw preprocessed lists -> projection with normals as extra channels -> correspondences -> loss terms and
    dL/dT, against the numbers the reference's ICPLosses + autograd produced on the same lists."""
    g = util.load_golden("loss_pair")
    G, dev = _geo(), _dev()
    vf, hf = util.kitti_fov()
    sensor = gpu_sensor(g["H"], g["W"], vf, hf)
    scans = [np.ascontiguousarray(np.concatenate([g["raw_tgt"].T, g["raw_tgt_n"].T], axis=0), dtype=np.float32),
             np.ascontiguousarray(np.concatenate([g["raw_src"].T, g["raw_src_n"].T], axis=0), dtype=np.float32)]
    out = run_project(scans, sensor, want_uv=False)
    img, nrm = out["image4"], out["aux"]
    # the kept sets must be the reference's filtered lists (deployer.py:258-261)
    assert int(out["kept"][0]) == g["tgt"].shape[1] and int(out["kept"][1]) == g["src"].shape[1]
    for qname in ("identity", "true", "random"):
        key = f"{mode}_{'p2p' if p2p else 'nop2p'}_{qname}"
        T = torch.from_numpy(g[key + "_T"]).to(dev).requires_grad_(True)
        nn, _, match = G.nn_correspond(img[1:2], nrm[1:2], out["packed"][0:1], out["packed_aux"][0:1], T, sensor, need_without_normals=p2p)
        terms, counts = G.icp_loss(T, img[1:2], nrm[1:2], match, nn, _flags(mode, p2p))
        total = terms[0, 0] + 2.0 * terms[0, 1] + 0.5 * terms[0, 2]
        total.backward()
        assert int(counts[0, 0]) == int(g[key + "_pairs"])
        _close(terms[0].detach().cpu().numpy(), g[key + "_losses"], what=key + " losses")
        _close(T.grad[0, :3].cpu().numpy(), g[key + "_gradT"][0, :3], what=key + " dL/dT")
        assert torch.all(T.grad[0, 3] == 0)


def test_loss_po2po_alone_golden_from_reference():
    """po2po_alone (src/losses/icp_losses.py:36-45): every source point against its nearest target, point-to-point
    only -- loss, pair count, correspondences and dL/dT against the reference's ICPLosses + autograd; the combination with a
    normal-based term (which the reference cannot evaluate) is rejected by the C ABI."""
    g = util.load_golden("loss_pair_alone")
    G, dev = _geo(), _dev()
    vf, hf = util.kitti_fov()
    sensor = gpu_sensor(g["H"], g["W"], vf, hf)
    scans = [np.ascontiguousarray(np.concatenate([g["raw_tgt"].T, g["raw_tgt_n"].T], axis=0), dtype=np.float32),
             np.ascontiguousarray(np.concatenate([g["raw_src"].T, g["raw_src_n"].T], axis=0), dtype=np.float32)]
    out = run_project(scans, sensor, want_uv=False)
    img, nrm = out["image4"], out["aux"]
    flags = G.LOSS_POINT_TO_POINT | G.LOSS_PO2PO_ALONE
    assert G.loss_flags({"point_to_point_loss": True, "point_to_plane_loss": False, "plane_to_plane_loss": False,
                         "normal_loss": "squared", "po2po_alone": True}) == flags
    for qname in ("identity", "true", "random"):
        T = torch.from_numpy(g[qname + "_T"]).to(dev).requires_grad_(True)
        nn, _, match = G.nn_correspond(img[1:2], nrm[1:2], out["packed"][0:1], out["packed_aux"][0:1], T, sensor,
                                       need_without_normals=True)
        terms, counts = G.icp_loss(T, img[1:2], nrm[1:2], match, nn, flags)
        terms[0, 0].backward()
        assert int(counts[0, 1]) == int(g[qname + "_pairs"]) and int(counts[0, 0]) == 0
        _close([float(terms[0, 0])], [float(g[qname + "_loss_po2po"])], what=qname + " po2po")
        assert float(terms[0, 1]) == 0.0 and float(terms[0, 2]) == 0.0
        _close(T.grad[0, :3].cpu().numpy(), g[qname + "_gradT"][0, :3], what=qname + " dL/dT")
        # correspondences of ALL source points: source point (bit pattern) -> matched target point, compared as maps
        occ = (nn[0].reshape(-1) >= 0).cpu().numpy()
        sp = img[1, :3].reshape(3, -1).cpu().numpy()[:, occ]
        mp = match[0, :3].reshape(3, -1).cpu().numpy()[:, occ]
        got = {sp[:, i].tobytes(): mp[:, i].tobytes() for i in range(sp.shape[1])}
        exp_t = g["tgt"][:, g[qname + "_nn"]]
        exp = {np.ascontiguousarray(g["src"][:, i]).tobytes(): np.ascontiguousarray(exp_t[:, i]).tobytes() for i in range(exp_t.shape[1])}
        assert got.keys() == exp.keys()
        differ = sum(1 for k in exp if got[k] != exp[k])
        print(f"po2po_alone {qname}: {differ} of {len(exp)} correspondences differ from the reference's KD-tree")
        assert differ <= 1e-3 * len(exp)
    with pytest.raises(Exception):
        G.icp_loss(T, img[1:2], nrm[1:2], match, nn, flags | G.LOSS_POINT_TO_PLANE)
    with pytest.raises(Exception):
        G.loss_flags({"point_to_point_loss": True, "point_to_plane_loss": True, "plane_to_plane_loss": False,
                      "normal_loss": "squared", "po2po_alone": True})


@pytest.mark.parametrize("mode,p2p", [("squared", False), ("linear", True)])
def test_loss_vs_oracle_batched(mode, p2p):
    """Batch of 4 different pairs (64x512) in one launch against the oracle per sample, values and gradients."""
    G, dev = _geo(), _dev()
    imgs, nrms, Ts = [], [], []
    rng = np.random.default_rng(8)
    for i in range(4):
        sensor, img, nrm, T_true = _pair_images(900 + i, 64, 512, 64, 600)
        imgs.append(img)
        nrms.append(nrm)
        T = torch.from_numpy(T_true).clone()
        T[:3, 3] += torch.tensor(rng.normal(0, 0.05, 3), dtype=torch.float32)
        Ts.append(T if i != 3 else _random_T(rng)[0])
    tgt = torch.stack([im[0] for im in imgs]); src = torch.stack([im[1] for im in imgs])
    tgt_n = torch.stack([n[0] for n in nrms]); src_n = torch.stack([n[1] for n in nrms])
    T = torch.stack(Ts).to(dev).requires_grad_(True)
    nn, _, match = G.nn_correspond(src, src_n, G.pack_image(tgt), G.pack_image(tgt_n), T, sensor, need_without_normals=p2p)
    terms, counts = G.icp_loss(T, src, src_n, match, nn, _flags(mode, p2p))
    w = torch.tensor([[1.0, 2.0, 0.5]], device=dev) * torch.arange(1, 5, device=dev).view(4, 1)
    (terms * w).sum().backward()
    for b in range(4):
        tp, tn, _ = util.lists_from_images(tgt[b].cpu(), tgt_n[b].cpu())
        sp, sn, _ = util.lists_from_images(src[b].cpu(), src_n[b].cpu())
        Tb = Ts[b].view(1, 4, 4).clone().requires_grad_(True)
        l, aux = orc.icp_losses(orc.transform_points(Tb, sp), orc.rotate_points(Tb, sn), tp, tn, normal_loss=mode,
                                point_to_point=p2p, return_aux=True)
        exp = torch.stack([l["loss_po2po"].reshape(()), l["loss_po2pl"].reshape(()), l["loss_pl2pl"].reshape(())])
        (exp * w[b].cpu()).sum().backward()
        assert int(counts[b, 0]) == aux["pairs"]
        _close(terms[b].detach().cpu().numpy(), exp.detach().numpy(), what=f"sample {b} losses")
        _close(T.grad[b, :3].cpu().numpy(), Tb.grad[0, :3].numpy(), what=f"sample {b} dL/dT")


def test_loss_identity_on_same_image_is_zero_and_deterministic():
    """Size-independent properties at the full 64x2048 size, batch 8: with T = I and source == target every
    point is its own neighbour and all terms vanish; two runs are bit-identical."""
    G, dev = _geo(), _dev()
    sensor, img, nrm, _ = _pair_images(4242, 64, 2048, 64, 2250)
    tgt = img[0:1].expand(8, -1, -1, -1).contiguous()
    tn = nrm[0:1].expand(8, -1, -1, -1).contiguous()
    T = torch.eye(4, device=dev).repeat(8, 1, 1)
    tgt_pk, tn_pk = G.pack_image(tgt), G.pack_image(tn)
    nn, vis, match = G.nn_correspond(tgt, tn, tgt_pk, tn_pk, T, sensor, need_without_normals=True)
    own = torch.arange(64 * 2048, device=dev, dtype=torch.int32).view(1, 64, 2048).expand(8, -1, -1)
    occ = ~((tgt[:, 0] == 0) & (tgt[:, 1] == 0) & (tgt[:, 2] == 0))
    assert torch.equal(nn[occ], own[occ]) and (nn[~occ] == -1).all()
    terms, counts = G.icp_loss(T, tgt, tn, match, nn, _flags("squared", True))
    occ6 = occ.unsqueeze(1).expand(-1, 3, -1, -1)
    assert torch.equal(match[:, :3][occ6], tgt[:, :3][occ6]) and torch.equal(match[:, 3:][occ6], tn[occ6])
    assert torch.all(terms[:, 1:] == 0)
    has = (tn != 0).any(dim=1)
    assert int(counts[0, 0]) == int(has[0].sum())
    # determinism with a non-trivial transform
    T2 = T.clone(); T2[:, :3, 3] = torch.tensor([0.3, -0.2, 0.05], device=dev)
    r = []
    for _ in range(2):
        nn2, _, m2 = G.nn_correspond(tgt, tn, tgt_pk, tn_pk, T2, sensor)
        t2, c2 = G.icp_loss(T2, tgt, tn, m2, nn2, _flags("squared", False))
        r.append((nn2.clone(), t2.clone()))
    assert torch.equal(r[0][0], r[1][0]) and torch.equal(r[0][1], r[1][1])
    assert torch.allclose(r[0][1][0], r[0][1][7])


def test_full_geometry_path_128_rings_2048_columns():
    """BASELINE config 4 shape (Ouster-style 128x2048, vFoV +-22.5 deg): projection -> online normals -> correspondences ->
    loss and dL/dT for one pair against the oracle (KD-tree on ~240k points)."""
    G, dev = _geo(), _dev()
    sensor, img, nrm, T_true = _pair_images(77128, 128, 2048, 128, 2250, vfov_deg=(-22.5, 22.5))
    T0 = torch.from_numpy(T_true).clone()
    T0[:3, 3] += torch.tensor([0.1, -0.05, 0.02])
    T = T0.view(1, 4, 4).to(dev).requires_grad_(True)
    tgt_pk, tgt_n_pk = G.pack_image(img[0:1]), G.pack_image(nrm[0:1])
    nn, vis, match = G.nn_correspond(img[1:2], nrm[1:2], tgt_pk, tgt_n_pk, T, sensor)
    terms, counts = G.icp_loss(T, img[1:2], nrm[1:2], match, nn, _flags("squared", False))
    (terms[0, 1] + terms[0, 2]).backward()
    tp, tn, tpix = util.lists_from_images(img[0].cpu(), nrm[0].cpu())
    sp, sn, spix = util.lists_from_images(img[1].cpu(), nrm[1].cpu())
    Tb = T0.view(1, 4, 4).clone().requires_grad_(True)
    l, aux = orc.icp_losses(orc.transform_points(Tb, sp), orc.rotate_points(Tb, sn), tp, tn, return_aux=True)
    (l["loss_po2pl"] + l["loss_pl2pl"]).sum().backward()
    assert abs(int(counts[0, 0]) - aux["pairs"]) <= 2                      # fp32 ties of q only
    _close(terms[0, 1:].detach().cpu().numpy(), [float(l["loss_po2pl"]), float(l["loss_pl2pl"])], what="128x2048 losses")
    _close(T.grad[0, :3].cpu().numpy(), Tb.grad[0, :3].numpy(), rel=2e-4, what="128x2048 dL/dT")
    got = nn[0].reshape(-1).cpu().long()[spix[aux["src_index_with_normals"]]]
    exp = tpix[aux["nn_with_normals"]]
    assert (got != exp).sum() <= 1e-4 * len(exp) + 2


def test_projection_random_small_clouds_with_duplicates_and_border_points():
    """20 random clouds (0..400 points, duplicated points = exact range+pixel ties, points just outside the vertical FoV)
    projected in ONE ragged batch; every scan obeys the projection contract."""
    from tests.test_oracle_properties import SENSOR, _cloud
    sensor = gpu_sensor(SENSOR.H, SENSOR.W, SENSOR.vfov, SENSOR.hfov)
    rng = np.random.default_rng(0)
    scans = [_cloud(int(rng.integers(0, 400)), int(rng.integers(0, 10_000))) for _ in range(20)]
    out = run_project(scans, sensor, want_uv=False)
    for s, scan in enumerate(scans):
        check_projection(scan, out, s, SENSOR)


def test_normals_known_answer_plane():
    """Wall x = 10 m: all normals are (-1,0,0), i.e. they face the sensor (viewpoint flip, normal_computation.py:78-81)."""
    G, dev = _geo(), _dev()
    vf, hf = util.kitti_fov()
    vv, uu = np.meshgrid(np.arange(16), np.arange(64), indexing="ij")
    el = vf[0] + vv / 15.0 * (vf[1] - vf[0])
    az = -0.6 + uu / 63.0 * 1.2
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    img = torch.zeros(1, 4, 16, 64, device=dev)
    img[0, :3] = torch.from_numpy((d * (10.0 / d[0])).astype(np.float32)).to(dev)
    n = G.normals(img)[0].cpu().numpy().reshape(3, -1).T
    has = np.abs(n).sum(1) > 0
    assert has.mean() > 0.9 and np.allclose(n[has], [-1.0, 0.0, 0.0], atol=2e-3)


def _torch_loss_terms(T, src, src_n, tgt, tgt_n, nn, mode, p2p):
    """The three loss modules as plain fp32 torch ops on the device (icp_losses.py:102-121,168-240 with the gathers the
    reference performs), differentiable with respect to T -- the full-size counterpart of the CPU oracle."""
    rows = []
    for b in range(src.shape[0]):
        idx = nn[b].reshape(-1).long()
        valid = idx >= 0
        idx = idx.clamp(min=0)
        p, n = src[b, :3].reshape(3, -1), src_n[b].reshape(3, -1)
        pt, nt = tgt[b, :3].reshape(3, -1)[:, idx], tgt_n[b].reshape(3, -1)[:, idx]
        R, t = T[b, :3, :3], T[b, :3, 3:4]
        q, rn = R @ p + t, R @ n
        has_s, has_t = (n != 0).any(dim=0), (nt != 0).any(dim=0)
        m = valid & has_s & has_t
        K = m.sum()
        r = ((q - pt) * nt).sum(dim=0)
        po2pl = (r[m] ** 2).sum() / K
        if mode == "linear":
            pl2pl = ((1.0 - (rn * nt).sum(dim=0))[m] ** 2).sum() / K
        else:
            pl2pl = ((rn - nt)[:, m] ** 2).sum() / K
        po2po = torch.zeros((), device=src.device)
        K2 = torch.zeros((), device=src.device)
        if p2p:
            m2 = valid & ~has_s & ~has_t
            K2 = m2.sum()
            po2po = ((q - pt)[:, m2] ** 2).sum() / (3 * K2)
        rows.append((torch.stack((po2po, po2pl, pl2pl)), int(K), int(K2)))
    return torch.stack([r[0] for r in rows]), [r[1] for r in rows], [r[2] for r in rows]


@pytest.mark.parametrize("mode,p2p", [("squared", False), ("linear", True)])
def test_loss_full_size_against_torch_ops(mode, p2p):
    """BASELINE size (64x2048, ~128k pairs per sample): loss terms, pair counts and dL/dT of the fused kernel against
    the same formulas written as plain fp32 torch ops (gather + transform + residuals + mean, autograd) on the device."""
    G, dev = _geo(), _dev()
    sensor, img, nrm, T_true = _pair_images(4243, 64, 2048, 64, 2250)
    src, src_n = torch.stack((img[1], img[0])), torch.stack((nrm[1], nrm[0]))
    tgt, tgt_n = torch.stack((img[0], img[1])), torch.stack((nrm[0], nrm[1]))
    T0 = torch.from_numpy(T_true).float()
    T0[:3, 3] += torch.tensor([0.05, -0.03, 0.02])
    T = torch.stack((T0, torch.linalg.inv(T0))).to(dev).requires_grad_(True)
    nn, _, match = G.nn_correspond(src, src_n, G.pack_image(tgt), G.pack_image(tgt_n), T, sensor, need_without_normals=p2p)
    terms, counts = G.icp_loss(T, src, src_n, match, nn, _flags(mode, p2p))
    w = torch.tensor([[1.0, 2.0, 0.5], [3.0, 1.5, 2.5]], device=dev)
    if not p2p:
        w[:, 0] = 0.0
    (terms * w).sum().backward()
    Tr = T.detach().clone().requires_grad_(True)
    exp, K, K2 = _torch_loss_terms(Tr, src, src_n, tgt, tgt_n, nn, mode, p2p)
    (exp * w).sum().backward()
    assert [int(c) for c in counts[:, 0]] == K and min(K) > 100000
    if p2p:
        assert [int(c) for c in counts[:, 1]] == K2 and min(K2) > 0
    for b in range(2):
        _close(terms[b].detach().cpu().numpy(), exp[b].detach().cpu().numpy(), what=f"sample {b} terms")
        _close(T.grad[b, :3].cpu().numpy(), Tr.grad[b, :3].cpu().numpy(), what=f"sample {b} dL/dT")


@pytest.mark.parametrize("shape", [(16, 130), (15, 131)])
def test_loss_ragged_image_sizes_against_torch_ops(shape):
    """Image sizes that are not a multiple of the 256-pixel chunk (16x130: whole chunks + a ragged tail) or not even a
    multiple of 4 pixels (15x131: no 16-byte loads at all) take the scalar path of the loss kernel; same check as at
    full size, with point-to-point enabled so that all 38 accumulators are exercised."""
    G, dev = _geo(), _dev()
    H, W = shape
    sensor, img, nrm, T_true = _pair_images(4300 + H, H, W, 16, 150)
    src, src_n = torch.stack((img[1], img[0])), torch.stack((nrm[1], nrm[0]))
    tgt, tgt_n = torch.stack((img[0], img[1])), torch.stack((nrm[0], nrm[1]))
    T0 = torch.from_numpy(T_true).float()
    T0[:3, 3] += torch.tensor([0.05, -0.03, 0.02])
    T = torch.stack((T0, torch.linalg.inv(T0))).to(dev).requires_grad_(True)
    nn, _, match = G.nn_correspond(src, src_n, G.pack_image(tgt), G.pack_image(tgt_n), T, sensor, need_without_normals=True)
    for mode in ("squared", "linear"):
        T.grad = None
        terms, counts = G.icp_loss(T, src, src_n, match, nn, _flags(mode, True))
        w = torch.tensor([[1.0, 2.0, 0.5], [3.0, 1.5, 2.5]], device=dev)
        Tr = T.detach().clone().requires_grad_(True)
        exp, K, K2 = _torch_loss_terms(Tr, src, src_n, tgt, tgt_n, nn, mode, True)
        use = torch.isfinite(exp)                       # an empty point-to-point set is 0/0 on both sides
        assert torch.equal(torch.isfinite(terms), use)
        (torch.where(use, terms, torch.zeros_like(terms)) * w).sum().backward()
        (torch.where(use, exp, torch.zeros_like(exp)) * w).sum().backward()
        assert [int(c) for c in counts[:, 0]] == K and [int(c) for c in counts[:, 1]] == K2 and min(K) > 500
        for b in range(2):
            _close(terms[b][use[b]].detach().cpu().numpy(), exp[b][use[b]].detach().cpu().numpy(), what=f"{mode} sample {b} terms")
            _close(T.grad[b, :3].cpu().numpy(), Tr.grad[b, :3].cpu().numpy(), what=f"{mode} sample {b} dL/dT")


def test_search_and_loss_survive_non_finite_and_huge_poses():
    """A diverged network can emit NaN / inf / 1e30 poses: the search must neither fault nor return indices outside the
    image, and the sample with a sane pose in the same batch must be unaffected."""
    from delora_amd.data import synthetic
    dev = _dev()
    H, W, B = 64, 512, 4
    vf, hf = util.kitti_fov()
    sensor = gpu_sensor(H, W, vf, hf)
    s1, s2, _ = synthetic.make_pair(41, rings=64, azimuth_steps=560)
    prj = run_project([s1, s2] * B, sensor, want_uv=False)
    nrm, nrm_pk = _geo().normals(prj["image4"], want_packed=True)
    img, nr = prj["image4"].view(B, 2, 4, H, W), nrm.view(B, 2, 3, H, W)
    tpk, tn = prj["packed"].view(B, 2, H, W, 4)[:, 0], nrm_pk.view(B, 2, H, W, 4)[:, 0]
    flags = _geo().LOSS_POINT_TO_PLANE | _geo().LOSS_PLANE_TO_PLANE
    ref = None
    for name, val in (("nan", float("nan")), ("inf", float("inf")), ("huge", 1e30), ("-huge", -1e30)):
        T = torch.eye(4, device=dev).repeat(B, 1, 1)
        T[0, 0, 3] = val
        T[1, 1, 1] = val
        T[2, :3, :3] = val
        nn, vis, match = _geo().nn_correspond(img[:, 1], nr[:, 1], tpk, tn, T, sensor)
        terms, counts = _geo().icp_loss(T, img[:, 1], nr[:, 1], match, nn, flags)
        torch.cuda.synchronize()
        assert int(nn.max()) < H * W and int(nn.min()) >= -1, name
        sane = terms[3].cpu().numpy()
        assert np.all(np.isfinite(sane)) and int(counts[3, 0]) > 1000, name
        ref = sane if ref is None else ref
        assert np.array_equal(sane, ref), name                     # the sample with T = I does not see its neighbours' poses


def test_quaternion_to_transform_kernel_against_the_torch_formulation():
    """dl_quat_to_T_fwd / _bwd (one kernel each way on the GPU) against the element-wise torch formulation of
    GeometryHandler (what CPU tensors take) and torch autograd through it: unnormalised, tiny and huge quaternions."""
    from delora_amd.models.model_parts import GeometryHandler
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    q = torch.randn((10, 4), generator=g)
    q[1] *= 1e-3
    q[2] *= 1e4
    q[3] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    q[4] = torch.tensor([1e-20, 0.0, 0.0, 0.0])          # below the normalisation eps: clamped branch
    t = torch.randn((10, 3), generator=g)
    G = torch.randn((10, 4, 4), generator=g)
    qc, tc = q.clone().requires_grad_(True), t.clone().requires_grad_(True)
    T_ref = GeometryHandler.get_transformation_matrix_quaternion(tc, qc, torch.device("cpu"))
    (T_ref * G).sum().backward()
    qg, tg = q.to(dev).requires_grad_(True), t.to(dev).requires_grad_(True)
    T = GeometryHandler.get_transformation_matrix_quaternion(tg, qg, dev)
    (T * G.to(dev)).sum().backward()
    util.measured("quaternion -> T kernel vs torch formulation (absolute)", float((T.detach().cpu() - T_ref.detach()).abs().max()), bound=1e-6)
    scale = qc.grad.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    util.measured("quaternion -> T kernel: dL/dq vs torch autograd (relative to the row's largest component)",
                  float(((qg.grad.cpu() - qc.grad).abs() / scale).max()), bound=2e-5)
    assert torch.equal(tg.grad.cpu(), tc.grad)
