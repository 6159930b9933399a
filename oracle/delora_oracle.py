"""CPU oracle for the DeLORA per-scan-pair training step.

TEST INFRASTRUCTURE ONLY.  This module restates, on the CPU, the algorithm of the
reference hot path (leggedrobotics/delora) so that the HIP kernels can be checked
against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; nothing under ``delora_amd/`` does, and the
product path raises when the HIP library is missing instead of falling back here.

Parity pin: every function below is checked against outputs of the reference itself
(imported in the build container with stubs for absent third-party modules) by
``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/`` and replayed by ``tests/test_oracle_golden.py``.  One boundary is
NOT pinned by the reference: ``quaternion_to_rotation_matrix`` is kornia 0.3.0
arithmetic (third-party, absent from /root/reference) restated from its published
formula -- "parity unpinned" there, see DESIGN.md.

Everything is written with torch CPU ops in the same operation order as the reference
so that fp32 results are bit-identical to the reference's CPU path on the same torch
build; scipy's cKDTree supplies the exact nearest neighbour exactly as the reference
does.  Citations are ``file:line`` into /root/reference.
"""
import math

import numpy as np
import scipy.spatial
import torch


# --------------------------------------------------------------------------------------
# Sensor description
# --------------------------------------------------------------------------------------
class Sensor:
    """Resolved per-dataset projection parameters (radians, python floats = float64).

    Mirrors what the reference reads from its flat config at projection time:
    ``config[dataset]["horizontal_cells"|"vertical_cells"|"vertical_field_of_view"]`` and
    ``config["horizontal_field_of_view"]`` (src/utility/projection.py:16,50-52).
    """

    def __init__(self, height, width, vfov, hfov=(-179.9 * math.pi / 180.0, 179.9 * math.pi / 180.0)):
        self.H = int(height)
        self.W = int(width)
        self.vfov = (float(vfov[0]), float(vfov[1]))
        self.hfov = (float(hfov[0]), float(hfov[1]))

    @staticmethod
    def kitti(width=2048, height=64):
        # config/config_datasets.yaml:19 (degrees there; bin/run_training.py:63-67 converts in place)
        return Sensor(height, width, (-24.5 * (np.pi / 180.0), 2.0 * (np.pi / 180.0)),
                      (-179.9 * (np.pi / 180.0), 179.9 * (np.pi / 180.0)))


# --------------------------------------------------------------------------------------
# a1-a3: spherical projection
# --------------------------------------------------------------------------------------
def compute_2d_coordinates(pc, sensor):
    """u,v image coordinates of a ``[1,C,N]`` cloud (src/utility/projection.py:21-31).

    Same op order: (atan2 - f0) / (f1 - f0) * (cells - 1), the field-of-view difference
    formed in float64 python arithmetic before it meets the fp32 tensor.
    """
    hf, vf = sensor.hfov, sensor.vfov
    u = ((torch.atan2(pc[:, 1, :], pc[:, 0, :]) - hf[0]) / (hf[1] - hf[0]) * (sensor.W - 1))
    v = ((torch.atan2(pc[:, 2, :], torch.norm(pc[:, :2, :], dim=1)) - vf[0]) / (vf[1] - vf[0]) * (sensor.H - 1))
    return u, v


def project_to_img(point_cloud, sensor):
    """Range-image projection of one scan (src/utility/projection.py:48-106).

    point_cloud: ``[1,C,N]`` fp32 (first three channels x,y,z).
    Returns the reference 5-tuple:
      image ``[1,C+1,H,W]`` (last channel = range, empty pixels 0),
      u, v ``[1,N]`` fp32 for ALL range-sorted points (unfiltered),
      point_cloud_indices ``[M]`` int64 into the input, ascending range,
      image_to_pointcloud_indices ``[1,M,2]`` int64 (v,u) of the kept points.
    The sequential first-wins loop of the reference (projection.py:34-43) is restated as a
    first-occurrence selection over the range-sorted order, which is the same function.
    """
    H, W = sensor.H, sensor.W
    B, C, N = point_cloud.shape
    with_range = torch.zeros((B, C + 1, N))
    with_range[:, :C, :] = point_cloud
    with_range[:, -1, :] = torch.norm(with_range[:, :3, :], dim=1).detach()        # :55-60
    order = torch.argsort(with_range[:, C, :], dim=1)                                # :63-64
    with_range = with_range[:, :, order[0]]                                          # :67
    u, v = compute_2d_coordinates(with_range, sensor)                                # :69-72
    ru, rv = torch.round(u), torch.round(v)                                          # half-to-even
    inside = (ru <= W - 1) & (ru >= 0) & (rv <= H - 1) & (rv >= 0)                   # :74-75
    uf = ru[inside].long().numpy()
    vf = rv[inside].long().numpy()
    with_range = with_range[:, :, inside[0]]
    lin = vf * W + uf
    first = np.zeros(len(lin), dtype=bool)
    if len(lin):
        _, first_idx = np.unique(lin, return_index=True)                             # first occurrence wins
        first[first_idx] = True
    keep = torch.from_numpy(first)
    uf_t = torch.from_numpy(uf)[keep]
    vf_t = torch.from_numpy(vf)[keep]
    with_range = with_range[:, :, keep]
    image = torch.zeros((B, C + 1, H, W))
    image[:, :, vf_t, uf_t] = with_range                                             # :98-103
    pix = torch.stack((vf_t, uf_t), dim=1).view(1, -1, 2)
    return image, u, v, order[inside][keep], pix


# --------------------------------------------------------------------------------------
# a4-a6: normals from a projected image
# --------------------------------------------------------------------------------------
def masked_covariance(nbrs):
    """Per-pixel covariance over present neighbours (src/utility/linalg.py:33-56).

    nbrs: ``[M,3,K]``; a neighbour is present iff any component != 0.  Mean = sum/n,
    centred differences of absent neighbours are zeroed, cov = D D^T / (n-1).
    """
    present = (nbrs[:, 0, :] != 0) | (nbrs[:, 1, :] != 0) | (nbrs[:, 2, :] != 0)
    n = torch.sum(present, dim=1)
    factor = torch.ones(1) / (n - 1)
    mean = torch.mean(nbrs, dim=2, keepdim=True) * nbrs.shape[2] / n.view(-1, 1, 1)
    diff = nbrs - mean
    diff.permute(0, 2, 1)[~present] = 0.0
    cov = diff.matmul(diff.permute(0, 2, 1))
    return factor.view(-1, 1, 1) * cov, n


def compute_normal_vectors(image, sensor, side=(7, 11), epsilon_range=0.5, min_neighbors=10,
                           return_aux=False):
    """Normals of every valid pixel of ``image[1,>=3,H,W]``
    (src/preprocessing/normal_computation.py:30-41, 53-87, 89-122).

    Valid pixel: x!=0 & y!=0 & z!=0 (AND), raster order.  Window (2a+1)x(2b+1) with
    a=int(side[0]/2) rows, b=int(side[1]/2) columns and CLAMPED coordinates (edge pixels
    are duplicated, no horizontal wrap).  Neighbours whose range deviates by more than
    epsilon_range from the centre range are zeroed; >= min_neighbors present neighbours are
    needed; the normal is the eigenvector of the smallest eigenvalue (fp32 symmetric
    eigensolver on the upper triangle, as torch.symeig did), flipped so that n.p <= 0.
    Returns (normals[M,3] with zeros where none, has_normal[M] bool, points[M,3]).
    """
    H, W = sensor.H, sensor.W
    img = image[0, :3]
    flat = img.reshape(3, H * W).transpose(0, 1)
    valid = (flat[:, 0] != 0) & (flat[:, 1] != 0) & (flat[:, 2] != 0)                # :35
    vv, uu = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    v0 = vv.reshape(-1)[valid]
    u0 = uu.reshape(-1)[valid]
    a, b = int(side[0] / 2), int(side[1] / 2)
    cols = []
    for dv in range(-a, a + 1):                                                      # :99-117
        for du in range(-b, b + 1):
            vn = torch.clamp(v0 + dv, 0, H - 1)
            un = torch.clamp(u0 + du, 0, W - 1)
            cols.append(img[:, vn, un])
    nbrs = torch.stack(cols, dim=0)                                                  # [K,3,M]
    centre = img[:, v0, u0].view(1, 3, -1)
    deviates = torch.abs(torch.norm(nbrs, dim=1) - torch.norm(centre, dim=1)) > epsilon_range   # :55-57
    nbrs.permute(0, 2, 1)[deviates] = 0.0                                            # :59
    cov, n = masked_covariance(nbrs.permute(2, 1, 0))                                # :61-63
    enough = n >= min_neighbors                                                      # :67-69
    evals, evecs = torch.linalg.eigh(cov[enough], UPLO="U")                          # :70 (symeig, upper)
    normals_e = evecs[:, :, 0].clone()                                               # :76
    pts = centre[0].permute(1, 0)
    dots = normals_e.view(-1, 1, 3).matmul(pts[enough].view(-1, 3, 1)).reshape(-1)   # :79-80
    normals_e[dots > 0] *= -1                                                        # :81
    normals = torch.zeros_like(pts)
    normals[enough] = normals_e
    if return_aux:
        ev_full = torch.zeros((pts.shape[0], 3))
        ev_full[enough] = evals
        return normals, enough, pts, {"count": n, "eigenvalues": ev_full, "v": v0, "u": u0}
    return normals, enough, pts


# --------------------------------------------------------------------------------------
# a8: quaternion (x,y,z,w) -> T   [kornia 0.3.0 restated; parity unpinned by the reference]
# --------------------------------------------------------------------------------------
def quaternion_to_rotation_matrix(q):
    """kornia 0.3.0 ``quaternion_to_rotation_matrix`` restated (call sites
    src/models/model_parts.py:31; pinned version conda/DeLORA-py3.9.yml:53).
    L2-normalise (eps 1e-12), unpack x,y,z,w, build R.  Corroborated inside the reference
    only by the ROS node's own quat2mat (src/ros_utils/odometry_publisher.py:113-126)."""
    qn = torch.nn.functional.normalize(q, p=2.0, dim=-1, eps=1e-12)
    x, y, z, w = torch.chunk(qn, chunks=4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.tensor(1.0)
    m = torch.stack([one - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, one - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, one - (txx + tyy)], dim=-1)
    return m.view(-1, 3, 3)


def transformation_matrix(translation, quaternion):
    """``T[B,4,4] = [R t; 0 1]`` (src/models/model_parts.py:38-44)."""
    R = quaternion_to_rotation_matrix(quaternion)
    T = torch.zeros((R.shape[0], 4, 4))
    T[:, :3, :3] = R
    T[:, 3, 3] = 1
    T[:, :3, 3] = translation
    return T


def transform_points(T, pc):
    """``R @ p + t`` on ``[1,3,M]`` (src/deploy/deployer.py:181-189)."""
    out = T[:, :3, :3].matmul(pc[:, :3, :])
    out = out + T[:, :3, 3].view(-1, 3, 1)
    return out


def rotate_points(T, pc):
    """``R @ n`` (src/deploy/deployer.py:181-182)."""
    return T[:, :3, :3].matmul(pc[:, :3, :])


# --------------------------------------------------------------------------------------
# a10-a11: KD-tree correspondences and ICP losses
# --------------------------------------------------------------------------------------
def nearest_target_indices(target, source):
    """Exact 3-D nearest neighbour of each source point among the target points, as the
    reference gets it from scipy (src/losses/icp_losses.py:24-26,34): cKDTree built on the
    fp32 target cast to float64, k=1 Euclidean query.  target/source: ``[1,3,M]``."""
    tree = scipy.spatial.cKDTree(target[0].permute(1, 0).detach().cpu())
    q = source.permute(0, 2, 1).detach().cpu().numpy()[0]
    if len(q) == 0:
        return torch.zeros(0, dtype=torch.long)
    return torch.from_numpy(np.asarray(tree.query(q)[1])).long()


def icp_losses(src_t, src_n_t, tgt, tgt_n, normal_loss="squared", point_to_point=False,
               point_to_plane=True, plane_to_plane=True, return_aux=False, po2po_alone=False):
    """ICP loss terms of one pair (src/losses/icp_losses.py:28-158; :196-206 point-to-plane,
    :224-240 plane-to-plane, :168-179 point-to-point).

    ``po2po_alone`` (:36-45): EVERY source point is paired with its nearest target point and the
    only term is the point-to-point MSE over all of them (no normal masks).  The reference defines
    the pair lists of the two normal-based terms only in the other branch, so enabling
    point_to_plane / plane_to_plane together with po2po_alone raises UnboundLocalError there
    (:135-146); this restatement raises too.

    src_t, src_n_t: transformed source points / rotated normals ``[1,3,Ms]`` (may carry
    autograd history); tgt, tgt_n: ``[1,3,Mt]``.  A point "has a normal" iff any component
    of its normal != 0.  Pairs = source-with-normal -> NN target, kept iff that target has a
    normal.  po2pl = mean (n_t.(s-t))^2, pl2pl squared = mean ||n_s-n_t||^2 (via norm then
    MSE, as the reference), linear = mean (1-n_s.n_t)^2, po2po = MSE over the 3K'
    components of source-without-normal -> target-without-normal pairs.
    """
    if po2po_alone:
        if point_to_plane or plane_to_plane:
            raise UnboundLocalError("po2po_alone with point_to_plane/plane_to_plane: the reference has no pair lists "
                                    "for the normal-based terms in this mode (icp_losses.py:135-146)")
        zero = torch.zeros(1)
        nn_all = nearest_target_indices(tgt, src_t)
        loss_po2po = torch.nn.MSELoss()(src_t, tgt[:, :, nn_all]) if point_to_point else zero
        losses = {"loss_po2po": loss_po2po, "loss_po2pl": zero, "loss_pl2pl": zero}
        if return_aux:
            return losses, {"nn_all": nn_all, "pairs": int(src_t.shape[2])}
        return losses
    src_has = (src_n_t[:, 0, :] != 0) | (src_n_t[:, 1, :] != 0) | (src_n_t[:, 2, :] != 0)
    tgt_has = (tgt_n[:, 0, :] != 0) | (tgt_n[:, 1, :] != 0) | (tgt_n[:, 2, :] != 0)
    s_w = src_t[:, :, src_has[0]]
    sn_w = src_n_t[:, :, src_has[0]]
    s_wo = src_t[:, :, ~src_has[0]]
    nn_w = nearest_target_indices(tgt, s_w)
    zero = torch.zeros(1)
    loss_po2po, loss_po2pl, loss_pl2pl = zero, zero, zero
    mse = torch.nn.MSELoss()
    if point_to_point:
        nn_wo = nearest_target_indices(tgt, s_wo)
        t_has_wo = tgt_has[:, nn_wo]
        t_pts = tgt[:, :, nn_wo][:, :, ~t_has_wo[0]]
        s_pts = s_wo[:, :, ~t_has_wo[0]]
        loss_po2po = mse(s_pts, t_pts)
    keep = tgt_has[:, nn_w][0]
    s_k = s_w[:, :, keep]
    sn_k = sn_w[:, :, keep]
    t_k = tgt[:, :, nn_w][:, :, keep]
    tn_k = tgt_n[:, :, nn_w][:, :, keep]
    if point_to_plane:
        dvec = s_k - t_k
        nd = dvec.permute(2, 0, 1).matmul(tn_k.permute(2, 1, 0))
        loss_po2pl = mse(nd, torch.zeros(nd.shape))
    if plane_to_plane:
        sn_p = sn_k.permute(2, 0, 1)
        if normal_loss == "linear":
            dots = torch.matmul(sn_p, tn_k.permute(2, 1, 0))
            loss_pl2pl = mse(1 - dots, torch.zeros(dots.shape))
        elif normal_loss == "squared":
            dist = torch.norm(sn_p - tn_k.permute(2, 0, 1), dim=2, keepdim=True)
            loss_pl2pl = mse(dist, torch.zeros(dist.shape))
        else:
            raise Exception("The normal loss which is defined here is not admissible.")
    losses = {"loss_po2po": loss_po2po, "loss_po2pl": loss_po2pl, "loss_pl2pl": loss_pl2pl}
    if return_aux:
        src_idx = torch.nonzero(src_has[0]).reshape(-1)
        return losses, {"nn_with_normals": nn_w, "src_index_with_normals": src_idx,
                        "pair_mask": keep, "pairs": int(keep.sum())}
    return losses


# --------------------------------------------------------------------------------------
# a12: the per-batch step glue (losses only; the CNN is supplied by the caller)
# --------------------------------------------------------------------------------------
def filter_to_projected(sample, sensor):
    """Project both scans of a sample dict and keep only the projected points
    (src/deploy/deployer.py:252-267).  Returns (image_1[4,H,W], image_2[4,H,W], lists dict)."""
    img1, _, _, idx1, _ = project_to_img(sample["scan_1"], sensor)
    img2, _, _, idx2, _ = project_to_img(sample["scan_2"], sensor)
    lists = {
        "scan_1": sample["scan_1"][:, :, idx1], "normal_list_1": sample["normal_list_1"][:, :, idx1],
        "scan_2": sample["scan_2"][:, :, idx2], "normal_list_2": sample["normal_list_2"][:, :, idx2],
    }
    return img1[0], img2[0], lists


def step_losses(lists_batch, T, lambda_po2pl=1.0, normal_loss="squared", point_to_point=False,
                point_to_plane=True, plane_to_plane=True, batch_offset=0, global_batch=None, po2po_alone=False):
    """Batch loss accumulation of ``Deployer.step`` (src/deploy/deployer.py:290-332).

    Reproduces the accumulation order of the reference: the running sums of the three terms are
    added to ``loss_pc`` inside the sample loop (:309-312), so sample j (0-based) carries weight
    (B-j)/B in ``loss_pc``, the loss that is back-propagated (:338).  ``batch_offset`` /
    ``global_batch`` let a data-parallel rank evaluate its slice of a global batch.
    """
    Bl = len(lists_batch)
    Bg = global_batch if global_batch is not None else Bl
    out = {k: torch.zeros(1) for k in ("loss_pc", "loss_po2po", "loss_po2pl", "loss_pl2pl")}
    per_sample = []
    for j in range(Bl):
        Tj = T[j:j + 1]
        L = lists_batch[j]
        s_t = transform_points(Tj, L["scan_2"])
        n_t = rotate_points(Tj, L["normal_list_2"])
        l = icp_losses(s_t, n_t, L["scan_1"], L["normal_list_1"], normal_loss=normal_loss,
                       point_to_point=point_to_point, point_to_plane=point_to_plane,
                       plane_to_plane=plane_to_plane, po2po_alone=po2po_alone)
        per_sample.append(l)
        out["loss_po2po"] = out["loss_po2po"] + l["loss_po2po"]
        out["loss_po2pl"] = out["loss_po2pl"] + lambda_po2pl * l["loss_po2pl"]
        out["loss_pl2pl"] = out["loss_pl2pl"] + l["loss_pl2pl"]
        if global_batch is None:
            # the reference's own order: running sums added inside the loop (:312)
            out["loss_pc"] = out["loss_pc"] + (out["loss_po2po"] + out["loss_po2pl"] + out["loss_pl2pl"])
        else:
            # same weights (Bg - global index), written per sample so that a rank can hold a slice
            c = l["loss_po2po"] + lambda_po2pl * l["loss_po2pl"] + l["loss_pl2pl"]
            out["loss_pc"] = out["loss_pc"] + (Bg - (batch_offset + j)) * c
    for k in out:
        out[k] = out[k] / Bg
    return out, per_sample


def visible_pixels(scan_transformed, sensor):
    """The ``visible_pixels`` metric of the reference step (src/deploy/deployer.py:349-352,365-367):
    number of points of the (last) transformed source scan with round(v) < H and v > 0."""
    _, _, v, _, _ = project_to_img(scan_transformed.detach(), sensor)
    return int(((torch.round(v) < sensor.H) & (v > 0)).sum())
