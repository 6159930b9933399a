"""Property tests of the oracle itself (CPU): independent restatements of small pieces, on random inputs with the edge
cases the golden vectors cannot enumerate (collisions, ties, empty clouds, points on the field-of-view border)."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from tests import util
from tests.util import orc

SENSOR = util.oracle_sensor(8, 24, *util.kitti_fov())


def _cloud(draw_n, seed):
    rng = np.random.default_rng(seed)
    r = rng.uniform(2.0, 30.0, draw_n)
    az = rng.uniform(-np.pi, np.pi, draw_n)
    el = rng.uniform(SENSOR.vfov[0] - 0.05, SENSOR.vfov[1] + 0.05, draw_n)
    pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)]).astype(np.float32)
    if draw_n > 3:                       # duplicates: exact ties in range and pixel
        pts[:, -1] = pts[:, 0]
    return pts


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 400), st.integers(0, 10_000))
def test_projection_equals_sequential_first_wins(n, seed):
    """project_to_img == range-sort, then walk the points and let the first one claim each pixel (the reference's loop,
    src/utility/projection.py:34-43), written out literally."""
    scan = _cloud(n, seed)
    image, u, v, idx, pix = orc.project_to_img(torch.from_numpy(scan).view(1, 3, -1), SENSOR)
    H, W = SENSOR.H, SENSOR.W
    rng = torch.norm(torch.from_numpy(scan).view(1, 3, -1), dim=1)[0]
    order = torch.argsort(rng.view(1, -1), dim=1)[0].numpy()
    taken = np.zeros((H, W), dtype=bool)
    exp_idx, exp_pix = [], []
    ref_pix, _, _ = util.reference_pixels(scan, SENSOR)
    for k in order:
        p = ref_pix[k]
        if p < 0:
            continue
        vv, uu = divmod(int(p), W)
        if not taken[vv, uu]:
            taken[vv, uu] = True
            exp_idx.append(k)
            exp_pix.append((vv, uu))
    assert idx.numpy().tolist() == [int(k) for k in exp_idx]
    assert pix.numpy().reshape(-1, 2).tolist() == [list(p) for p in exp_pix]
    img = image.numpy()[0]
    assert int((img[3] != 0).sum()) == len(exp_idx)
    for k, (vv, uu) in zip(exp_idx, exp_pix):
        assert np.array_equal(img[:3, vv, uu], scan[:, k]) and img[3, vv, uu] == rng[k].item()


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 200), st.integers(1, 200), st.integers(0, 10_000))
def test_nearest_neighbour_equals_brute_force(ns, nt, seed):
    rng = np.random.default_rng(seed)
    s = rng.normal(0, 10, (3, ns)).astype(np.float32)
    t = rng.normal(0, 10, (3, nt)).astype(np.float32)
    nn = orc.nearest_target_indices(torch.from_numpy(t).view(1, 3, -1), torch.from_numpy(s).view(1, 3, -1)).numpy()
    d = ((s.astype(np.float64)[:, :, None] - t.astype(np.float64)[:, None, :]) ** 2).sum(0)
    assert np.allclose(d[np.arange(ns), nn], d.min(axis=1), rtol=0, atol=0)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10_000), st.sampled_from(["squared", "linear"]), st.booleans())
def test_loss_terms_equal_their_formulas(seed, mode, p2p):
    """po2pl = mean (n_t.(s-t))^2, pl2pl = mean |n_s-n_t|^2 or mean (1-n_s.n_t)^2, po2po = mean over 3K' components
    (src/losses/icp_losses.py:168-240), evaluated with plain numpy over the oracle's own correspondences."""
    rng = np.random.default_rng(seed)
    m = 60
    tgt = rng.normal(0, 5, (3, m)).astype(np.float32)
    src = (tgt + rng.normal(0, 0.2, (3, m))).astype(np.float32)
    def normals(k):
        n = rng.normal(size=(3, k)).astype(np.float32)
        n /= np.linalg.norm(n, axis=0, keepdims=True)
        n[:, rng.uniform(size=k) < 0.3] = 0.0                      # some points have no normal
        return n
    tn, sn = normals(m), normals(m)
    T = lambda a: torch.from_numpy(a).view(1, 3, -1)
    l, aux = orc.icp_losses(T(src), T(sn), T(tgt), T(tn), normal_loss=mode, point_to_point=p2p, return_aux=True)
    has_s, has_t = (sn != 0).any(0), (tn != 0).any(0)
    d = ((src.astype(np.float64)[:, :, None] - tgt.astype(np.float64)[:, None, :]) ** 2).sum(0)
    nn = d.argmin(1)
    keep = has_s & has_t[nn]
    assert aux["pairs"] == int(keep.sum())
    if keep.any():
        r = ((src[:, keep] - tgt[:, nn[keep]]) * tn[:, nn[keep]]).sum(0)
        assert np.isclose(float(l["loss_po2pl"]), float((r.astype(np.float64) ** 2).mean()), rtol=1e-5)
        if mode == "squared":
            e = ((sn[:, keep] - tn[:, nn[keep]]).astype(np.float64) ** 2).sum(0).mean()
        else:
            e = ((1 - (sn[:, keep] * tn[:, nn[keep]]).sum(0).astype(np.float64)) ** 2).mean()
        assert np.isclose(float(l["loss_pl2pl"]), float(e), rtol=1e-5)
    if p2p:
        k2 = (~has_s) & (~has_t[nn])
        if k2.any():
            e = ((src[:, k2] - tgt[:, nn[k2]]).astype(np.float64) ** 2).mean()
            assert np.isclose(float(l["loss_po2po"]), float(e), rtol=1e-5)


def test_normals_of_a_plane_point_back_at_the_sensor():
    """A synthetic wall x = 10 seen by the sensor: every normal the oracle returns is (-1,0,0) and faces the sensor."""
    vf, hf = util.kitti_fov()
    sen = util.oracle_sensor(16, 64, vf, (-0.6, 0.6))
    vv, uu = np.meshgrid(np.arange(16), np.arange(64), indexing="ij")
    el = vf[0] + vv / 15.0 * (vf[1] - vf[0])
    az = -0.6 + uu / 63.0 * 1.2
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    pts = d * (10.0 / d[0])
    img = torch.zeros(1, 4, 16, 64)
    img[0, :3] = torch.from_numpy(pts.astype(np.float32))
    n, has, p = orc.compute_normal_vectors(img, sen)
    assert has.float().mean() > 0.9
    nn = n[has].numpy()
    assert np.allclose(nn, np.array([-1.0, 0, 0]), atol=2e-3)
    assert ((nn * p[has].numpy()).sum(1) <= 0).all()
