"""Shared helpers of the test-suite (the only place besides bench.py's cpu_baseline leg and
__graft_entry__.smoke() that touches oracle/)."""
import os

import numpy as np
import torch

from oracle import delora_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def oracle_sensor(H, W, vfov, hfov):
    return orc.Sensor(int(H), int(W), [float(v) for v in vfov], [float(h) for h in hfov])


def kitti_fov():
    return ((-24.5 * (np.pi / 180.0), 2.0 * (np.pi / 180.0)), (-179.9 * (np.pi / 180.0), 179.9 * (np.pi / 180.0)))


def reference_pixels(xyz, sensor):
    """Per-point (pixel index or -1, u, v) as the reference computes them: fp32 torch ops of
    compute_2D_coordinates + round-half-even + FoV test (elementwise, so input order is irrelevant)."""
    pc = torch.from_numpy(np.ascontiguousarray(xyz[:3])).view(1, 3, -1)
    u, v = orc.compute_2d_coordinates(pc, sensor)
    ru, rv = torch.round(u[0]), torch.round(v[0])
    inside = (ru <= sensor.W - 1) & (ru >= 0) & (rv <= sensor.H - 1) & (rv >= 0)
    pix = torch.where(inside, rv.long() * sensor.W + ru.long(), torch.full_like(ru, -1).long())
    return pix.numpy(), u[0].numpy(), v[0].numpy()


def ambiguity_mask(xyz, sensor, tol=2e-3):
    """Points whose exact (fp64) image coordinate lies within tol px of a rounding boundary: the only points
    whose pixel may differ between two correct fp32 evaluations of atan2."""
    p = np.asarray(xyz[:3], dtype=np.float64)
    u = (np.arctan2(p[1], p[0]) - sensor.hfov[0]) / (sensor.hfov[1] - sensor.hfov[0]) * (sensor.W - 1)
    v = (np.arctan2(p[2], np.hypot(p[0], p[1])) - sensor.vfov[0]) / (sensor.vfov[1] - sensor.vfov[0]) * (sensor.H - 1)
    fu = np.abs(u - np.floor(u) - 0.5)
    fv = np.abs(v - np.floor(v) - 0.5)
    return (fu < tol) | (fv < tol)


def lists_from_images(image4, normals):
    """Raster-order point / normal lists ``[1,3,M]`` (CPU) of the occupied pixels of one image, plus the pixel ids."""
    img = image4.detach().cpu()
    occ = ~((img[0] == 0) & (img[1] == 0) & (img[2] == 0))
    pix = torch.nonzero(occ.reshape(-1)).reshape(-1)
    pts = img[:3].reshape(3, -1)[:, pix].contiguous().view(1, 3, -1)
    nrm = normals.detach().cpu().reshape(3, -1)[:, pix].contiguous().view(1, 3, -1)
    return pts, nrm, pix


def tainted_pixels(xyz, sensor, tol=2e-3):
    """Pixels an ambiguous point (see ambiguity_mask) may or may not land in: both rounding candidates in u and v.
    Two correct fp32 evaluations of the projection (different CPUs' Sleef paths, or the GPU's) can only differ there."""
    p = np.asarray(xyz[:3], dtype=np.float64)
    amb = ambiguity_mask(xyz, sensor, tol)
    u = (np.arctan2(p[1], p[0]) - sensor.hfov[0]) / (sensor.hfov[1] - sensor.hfov[0]) * (sensor.W - 1)
    v = (np.arctan2(p[2], np.hypot(p[0], p[1])) - sensor.vfov[0]) / (sensor.vfov[1] - sensor.vfov[0]) * (sensor.H - 1)
    t = np.zeros((sensor.H, sensor.W), dtype=bool)
    ua, va = u[amb], v[amb]
    for du in (-tol * 2, tol * 2):
        for dv in (-tol * 2, tol * 2):
            uu, vv = np.rint(ua + du).astype(np.int64), np.rint(va + dv).astype(np.int64)
            ok = (uu >= 0) & (uu < sensor.W) & (vv >= 0) & (vv < sensor.H)
            t[vv[ok], uu[ok]] = True
    return t


def repo_config(H, W, dataset="kitti", device="cpu", **over):
    """The flat run config built from THIS repo's config/*.yaml the way bin/run_training.py builds it."""
    import copy
    from delora_amd import config as cfgmod
    from tests.conftest import ROOT
    cfg = cfgmod.load_yaml_config(os.path.join(ROOT, "config"))
    cfg["datasets"] = [dataset]
    cfgmod.degrees_to_radians(cfg)
    cfg[dataset]["data_identifiers"] = cfg[dataset]["training_identifiers"]
    cfg[dataset]["vertical_cells"], cfg[dataset]["horizontal_cells"] = int(H), int(W)
    cfg["device"] = torch.device(device)
    cfg["checkpoint"] = None
    cfg["training_run_name"] = cfg["run_name"] = "test"
    cfg["mode"] = "training"
    cfg.update(over)
    return copy.deepcopy(cfg)


class ListDataset(torch.utils.data.Dataset):
    def __init__(self, samples):
        self.samples = samples

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return dict(self.samples[i])
