"""Deterministic synthetic LiDAR scans (no KITTI in the build/bench environment).

A rotating multi-ring sensor is ray-cast inside a box room with a ground plane and a few
box obstacles; the second scan of a pair re-observes the same scene from a pose moved by
a known small SE(3).  Only numpy (``default_rng(seed)``) is used, so the same seed gives
the same scans in the build container and on the GPU box.  The layout matches what the
reference reads from disk: a scan is ``[3,N]`` fp32, x forward / y left / z up, sensor at
the origin (src/data/dataset.py:94-102 loads ``[M,3]`` npy and permutes to ``[1,3,M]``).
"""
import math

import numpy as np


def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


class Scene:
    """Axis-aligned room [-ax,ax]x[-ay,ay], ground z=-h, ceiling z=+c, plus solid boxes."""

    def __init__(self, rng, half_x=22.0, half_y=9.0, sensor_height=1.73, ceiling=6.0, n_boxes=6, cross_walls=0, pillars=0):
        """cross_walls / pillars (round 6, default 0 = the scenes every committed fixture was generated with): thin walls ACROSS the
        direction of travel, reaching from a side wall to 2.5 m short of the centre line, on alternating sides, and full-height square
        pillars -- surfaces whose normals have a component along x, so that a point-to-plane loss can observe the forward translation
        (in a bare corridor walls, ground and ceiling are all parallel to the motion)."""
        self.half_x, self.half_y = half_x, half_y
        self.ground, self.ceiling = -sensor_height, ceiling
        boxes = []
        for k in range(int(cross_walls)):
            cx = -0.9 * half_x + (k + rng.uniform(0.2, 0.8)) * (1.8 * half_x / max(1, int(cross_walls)))
            side = 1.0 if k % 2 == 0 else -1.0
            y_in, y_out = side * rng.uniform(2.5, 4.0), side * half_y
            boxes.append((cx - 0.15, cx + 0.15, min(y_in, y_out), max(y_in, y_out), self.ground, self.ground + rng.uniform(2.0, 3.5)))
        for _ in range(int(pillars)):
            cx = rng.uniform(-0.9 * half_x, 0.9 * half_x)
            cy = rng.uniform(2.0, 0.9 * half_y) * (1.0 if rng.uniform() < 0.5 else -1.0)
            r = rng.uniform(0.25, 0.5)
            boxes.append((cx - r, cx + r, cy - r, cy + r, self.ground, ceiling))
        for _ in range(n_boxes):
            cx = rng.uniform(-0.8 * half_x, 0.8 * half_x)
            cy = rng.uniform(-0.8 * half_y, 0.8 * half_y)
            if abs(cx) < 3.0 and abs(cy) < 3.0:           # keep the sensor in free space
                cx += 6.0 * (1 if cx >= 0 else -1)
            sx, sy, sz = rng.uniform(0.6, 2.5), rng.uniform(0.6, 2.5), rng.uniform(0.8, 2.5)
            boxes.append((cx - sx, cx + sx, cy - sy, cy + sy, self.ground, self.ground + sz))
        self.boxes = boxes

    def cast(self, origin, dirs):
        """Distance along each unit ray (``dirs[K,3]``) to the first surface; inf if none."""
        o = origin.reshape(1, 3)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / dirs
        t_best = np.full(dirs.shape[0], np.inf)
        # room: we are inside, take the exit distance of the slab intersection
        lo = np.array([-self.half_x, -self.half_y, self.ground])
        hi = np.array([self.half_x, self.half_y, self.ceiling])
        t1 = (lo - o) * inv
        t2 = (hi - o) * inv
        t_exit = np.nanmin(np.maximum(t1, t2), axis=1)
        t_best = np.minimum(t_best, np.where(t_exit > 0, t_exit, np.inf))
        for (x0, x1, y0, y1, z0, z1) in self.boxes:
            lo = np.array([x0, y0, z0])
            hi = np.array([x1, y1, z1])
            t1 = (lo - o) * inv
            t2 = (hi - o) * inv
            t_in = np.nanmax(np.minimum(t1, t2), axis=1)
            t_out = np.nanmin(np.maximum(t1, t2), axis=1)
            hit = (t_in > 0) & (t_in <= t_out)
            t_best = np.where(hit & (t_in < t_best), t_in, t_best)
        return t_best


def scan_scene(scene, rng, rings=64, azimuth_steps=2250, vfov_deg=(-24.5, 2.0), pose=None,
               noise=0.01, min_range=1.0, max_range=80.0, dropout=0.02, point_order="shuffled"):
    """One revolution: ``rings`` x ``azimuth_steps`` rays, returns ``[3,N]`` fp32 in the sensor frame.  ``point_order``:
    "shuffled" (default: a random permutation -- the hardest order for the projection's vote, and what every committed fixture was
    generated with), "raster" (ring after ring, azimuth ascending: the order of the reference's stored point lists,
    src/preprocessing/preprocesser.py:60-67) or "firing" (azimuth step after azimuth step, all rings of a step together: the order
    a spinning sensor's driver delivers).  The same random numbers are drawn in every mode, so the SET of points is the same."""
    R, t = (np.eye(3), np.zeros(3)) if pose is None else pose
    el = np.deg2rad(np.linspace(vfov_deg[0], vfov_deg[1], rings))
    el = el + rng.normal(0, 2e-4, size=rings)
    az0 = rng.uniform(0, 2 * math.pi / azimuth_steps, size=rings)
    az = (np.arange(azimuth_steps) * (2 * math.pi / azimuth_steps))[None, :] + az0[:, None] - math.pi
    elg = np.broadcast_to(el[:, None], az.shape)
    d = np.stack([np.cos(elg) * np.cos(az), np.cos(elg) * np.sin(az), np.sin(elg)], axis=-1).reshape(-1, 3)
    dist = scene.cast(t, d @ R.T)
    dist = dist + rng.normal(0, noise, size=dist.shape)
    ok = np.isfinite(dist) & (dist > min_range) & (dist < max_range) & (rng.uniform(size=dist.shape) > dropout)
    pts = d[ok] * dist[ok, None]
    order = rng.permutation(pts.shape[0])        # drawn in every mode (keeps the random stream of later scans identical)
    if point_order == "raster":
        order = np.arange(pts.shape[0])
    elif point_order == "firing":
        ring, step = np.divmod(np.nonzero(ok)[0], azimuth_steps)
        order = np.lexsort((ring, step))
    elif point_order != "shuffled":
        raise ValueError(f"point_order {point_order!r}: shuffled, raster or firing")
    return np.ascontiguousarray(pts[order].T.astype(np.float32))


def make_pair(seed, rings=64, azimuth_steps=2250, vfov_deg=(-24.5, 2.0), max_shift=1.0, max_rot_deg=3.0, **kw):
    """(scan_1, scan_2, T_21) with T_21 mapping scan-2 coordinates into the scan-1 frame (what the
    network is asked to predict: target = scan 1, source = scan 2, src/deploy/deployer.py:294-307)."""
    rng = np.random.default_rng(seed)
    scene = Scene(rng)
    s1 = scan_scene(scene, rng, rings, azimuth_steps, vfov_deg, None, **kw)
    yaw, pitch, roll = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg)), np.deg2rad(rng.uniform(-0.3, 0.3)), np.deg2rad(rng.uniform(-0.3, 0.3))
    R = _rot_zyx(yaw, pitch, roll)
    t = np.array([rng.uniform(0.2, max_shift), rng.uniform(-0.1, 0.1), rng.uniform(-0.02, 0.02)])
    s2 = scan_scene(scene, rng, rings, azimuth_steps, vfov_deg, (R, t), **kw)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return s1, s2, T.astype(np.float32)


def pad_or_trim(scan, n, rng=None):
    """Fixed-size ``[3,n]`` variant (clean byte counts for the bench): trims, or repeats points."""
    N = scan.shape[1]
    if N >= n:
        return np.ascontiguousarray(scan[:, :n])
    rng = rng or np.random.default_rng(0)
    extra = scan[:, rng.integers(0, N, size=n - N)]
    return np.ascontiguousarray(np.concatenate([scan, extra], axis=1))


def portable_cloud(seed, n):
    """``[3,n]`` fp32 cloud built from the PCG64 bit stream with +,-,*,/ only (no libm calls), hence bit-identical
    on every machine; used where a committed digest must be reproduced from a seed.  Ground disc plus four walls,
    roughly the range/elevation statistics of a 64-ring scan."""
    rng = np.random.default_rng(seed)
    u = rng.random((6, n))
    ground = u[0] < 0.55
    gx, gy = (u[1] - 0.5) * 90.0, (u[2] - 0.5) * 90.0
    gz = -1.73 + (u[3] - 0.5) * 0.04
    side = u[4]
    wx = np.where(side < 0.25, 22.0, np.where(side < 0.5, -22.0, (u[1] - 0.5) * 44.0)) + (u[3] - 0.5) * 0.03
    wy = np.where(side < 0.5, (u[2] - 0.5) * 18.0, np.where(side < 0.75, 9.0, -9.0)) + (u[5] - 0.5) * 0.03
    wz = -1.73 + u[5] * 2.4
    pts = np.where(ground[None, :], np.stack([gx, gy, gz]), np.stack([wx, wy, wz]))
    return np.ascontiguousarray(pts.astype(np.float32))


def portable_pair(seed, n, yaw_t=0.012, shift=(0.6, -0.05, 0.01)):
    """A scan pair WITH normal lists, bit-identical on every machine (PCG64 stream, +,-,*,/ and the correctly rounded sqrt only):
    ``{"scan_1","scan_2","normal_list_1","normal_list_2"}`` as ``[3,n]`` fp32 -- the stored-list form the reference reads from disk
    (src/data/dataset.py:143-153).  Both scans sample the same surfaces (ground + four walls, ``portable_cloud``'s scene) with
    different points; scan 2 is seen from a pose moved by ``shift`` and rotated about z by 2 atan(yaw_t) (a rational rotation:
    c = (1-t^2)/(1+t^2), s = 2t/(1+t^2)).  Normals are the analytic surface normals, perturbed a little, normalised, pointing at the
    sensor.  Used where a committed reference result must be reproduced from a seed (tests/golden/step_full_*.npz)."""
    rng = np.random.default_rng(seed)
    c, s = (1.0 - yaw_t * yaw_t) / (1.0 + yaw_t * yaw_t), 2.0 * yaw_t / (1.0 + yaw_t * yaw_t)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    t = np.asarray(shift, dtype=np.float64)
    out = {}
    for name, moved in (("1", False), ("2", True)):
        u = rng.random((9, n))
        ground = u[0] < 0.55
        gx, gy = (u[1] - 0.5) * 90.0, (u[2] - 0.5) * 90.0
        gz = -1.73 + (u[3] - 0.5) * 0.04
        side = u[4]
        wx = np.where(side < 0.25, 22.0, np.where(side < 0.5, -22.0, (u[1] - 0.5) * 44.0)) + (u[3] - 0.5) * 0.03
        wy = np.where(side < 0.5, (u[2] - 0.5) * 18.0, np.where(side < 0.75, 9.0, -9.0)) + (u[5] - 0.5) * 0.03
        wz = -1.73 + u[5] * 2.4
        pts = np.where(ground[None, :], np.stack([gx, gy, gz]), np.stack([wx, wy, wz]))
        zero, one = np.zeros(n), np.ones(n)
        nw = np.where(side[None, :] < 0.25, np.stack([-one, zero, zero]), np.where(side[None, :] < 0.5, np.stack([one, zero, zero]),
                      np.where(side[None, :] < 0.75, np.stack([zero, -one, zero]), np.stack([zero, one, zero]))))
        nrm = np.where(ground[None, :], np.stack([zero, zero, one]), nw) + (u[6:9] - 0.5) * 0.06
        nrm = nrm / np.sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2])[None, :]
        if moved:                                   # sensor frame 2: p2 = R^T (p1 - t), n2 = R^T n1 (explicit sums: no BLAS, fixed order)
            d = pts - t[:, None]
            pts = np.stack([(R[0, i] * d[0] + R[1, i] * d[1]) + R[2, i] * d[2] for i in range(3)])
            nrm = np.stack([(R[0, i] * nrm[0] + R[1, i] * nrm[1]) + R[2, i] * nrm[2] for i in range(3)])
        out["scan_" + name] = np.ascontiguousarray(pts.astype(np.float32))
        out["normal_list_" + name] = np.ascontiguousarray(nrm.astype(np.float32))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    out["T_21"] = T.astype(np.float32)
    return out


def portable_state_dict(seed, shapes, head_scale=0.05):
    """A state_dict for ``OdometryModel`` from the PCG64 stream with +,-,*,/ only (bit-identical everywhere): ``shapes`` = ordered
    ``{name: shape}`` of the 30 tensors.  Convolutions: uniform with the standard deviation of the reference's initialisation
    (kaiming-normal, fan_out, tanh gain 5/3: src/models/resnet_modified.py:64-66); linear layers: uniform(+-1/sqrt(fan_in)); the last
    layer of each head scaled by ``head_scale`` with the rotation bias at (0,0,0,1), so that the network predicts a small motion."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in shapes.items():
        shape = tuple(int(d) for d in shape)
        count = 1
        for d in shape:
            count *= d
        u = rng.random(count).reshape(shape) - 0.5
        if len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            std = (5.0 / 3.0) / np.sqrt(float(fan_out))
            w = u * (std * 2.0 * np.sqrt(3.0))
        else:
            fan_in = shape[-1] if len(shape) == 2 else sd_fan_in(sd, name)
            w = u * (2.0 / np.sqrt(float(fan_in)))
        last = name.startswith("fully_connected_") and name.split(".")[1] == "3"
        if last:
            w = w * head_scale
            if name == "fully_connected_rotation.3.bias":
                w = w + np.array([0.0, 0.0, 0.0, 1.0])
        sd[name] = np.ascontiguousarray(w.astype(np.float32))
    return sd


def sd_fan_in(sd, bias_name):
    """fan_in of the linear layer a bias belongs to (its weight precedes it in the state_dict)."""
    return sd[bias_name[:-len("bias")] + "weight"].shape[1]


def digest(arrays):
    """sha256 over the raw bytes of a sequence of arrays (fixtures store it to prove that regenerated inputs are the committed ones)."""
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make_sequence(seed, n_scans, rings=64, azimuth_steps=2250, vfov_deg=(-24.5, 2.0), step=(0.45, 0.02, 0.005), max_yaw_deg=1.5, scene=None, **kw):
    """``n_scans`` consecutive scans of ONE scene along a smooth trajectory (sensor poses R_k, t_k in the scene frame; every step
    moves ~``step`` metres and turns by up to ``max_yaw_deg``): a synthetic sequence in the sense of the reference's dataset
    (src/data/dataset.py:124-154: sample k = the pair (scan k, scan k+1)).  Returns (list of ``[3,N_k]`` fp32 scans, list of 4x4 poses)."""
    rng = np.random.default_rng(seed)
    scene = Scene(rng, half_x=40.0, **(scene or {}))
    R, t = np.eye(3), np.array([-0.5 * step[0] * n_scans, 0.0, 0.0])
    scans, poses = [], []
    for _ in range(n_scans):
        scans.append(scan_scene(scene, rng, rings, azimuth_steps, vfov_deg, (R, t), **kw))
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        poses.append(T)
        R = R @ _rot_zyx(np.deg2rad(rng.uniform(-max_yaw_deg, max_yaw_deg)), np.deg2rad(rng.uniform(-0.2, 0.2)), np.deg2rad(rng.uniform(-0.2, 0.2)))
        t = t + R @ (np.asarray(step) * rng.uniform(0.6, 1.4))
    return scans, poses


def write_tree(path, scans, sequence=0, normals=None):
    """Write one sequence in the reference's on-disk training format (src/preprocessing/preprocesser.py:64-68):
    ``<path>/<sequence:02d>/scans/<idx:06d>.npy`` = ``[M,3]`` fp32 and, when ``normals`` is given, ``.../normals/<idx:06d>.npy``.
    Without the ``normals`` directory the tree is an xyz-only one: the step then estimates the normals online."""
    import os
    root = os.path.join(path, format(int(sequence), "02d"))
    os.makedirs(os.path.join(root, "scans"), exist_ok=True)
    if normals is not None:
        os.makedirs(os.path.join(root, "normals"), exist_ok=True)
    for i, s in enumerate(scans):
        np.save(os.path.join(root, "scans", format(i, "06d") + ".npy"), np.ascontiguousarray(np.asarray(s, dtype=np.float32).T))
        if normals is not None:
            np.save(os.path.join(root, "normals", format(i, "06d") + ".npy"), np.ascontiguousarray(np.asarray(normals[i], dtype=np.float32).T))
    return root
