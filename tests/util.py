"""Shared helpers of the test-suite (the only place besides bench.py's cpu_baseline leg and
__graft_entry__.smoke() that touches oracle/)."""
import os

import numpy as np
import torch

from oracle import delora_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Measured deviations (mismatch counts, worst relative errors) recorded by the parity tests: printed as a table at the end
# of the session and written to gpurun_out/parity_measured.json (tests/conftest.py), so that a regression inside a
# tolerance is still visible as a number.
MEASURED = {}


def measured(name, value, bound=None):
    """Record (and print) a measured deviation; with ``bound`` also assert value <= bound."""
    MEASURED[name] = {"value": float(value), "bound": None if bound is None else float(bound)}
    print(f"[measured] {name} = {float(value):.6g}" + (f"  (bound {float(bound):.6g})" if bound is not None else ""))
    if bound is not None:
        assert float(value) <= float(bound), f"{name}: measured {float(value):.6g} exceeds the bound {float(bound):.6g}"


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


class OracleStepGeometry:
    """Test-only geometry backend with the interface of delora_amd.deploy.step_geometry.HipStepGeometry, evaluated by
    the CPU oracle.  It lets the host logic of the step (loss weighting, DDP, optimiser) be exercised without a GPU;
    the product never constructs it (Deployer defaults to the HIP backend and fails without the library)."""

    def prepare(self, samples, sensor, normal_params):
        o_sensor = orc.Sensor(sensor.H, sensor.W, sensor.vfov, sensor.hfov)
        stacked, lists = [], []
        a, b, eps, min_n = normal_params
        for s in samples:
            entry, imgs = {}, []
            for k in ("1", "2"):
                scan = s["scan_" + k].detach().cpu()
                img, _, _, idx, _ = orc.project_to_img(scan, o_sensor)
                imgs.append(img[0])
                if s.get("normal_list_" + k) is not None:
                    entry["scan_" + k] = scan[:, :, idx]
                    entry["normal_list_" + k] = s["normal_list_" + k].detach().cpu()[:, :, idx]
                else:
                    n, has, pts = orc.compute_normal_vectors(img.clone(), o_sensor, side=(2 * a + 1, 2 * b + 1),
                                                             epsilon_range=eps, min_neighbors=min_n)
                    entry["scan_" + k] = pts.t().contiguous().view(1, 3, -1)
                    entry["normal_list_" + k] = n.t().contiguous().view(1, 3, -1)
            stacked.append(torch.cat(imgs, dim=0))
            lists.append(entry)
        return {"stacked": torch.stack(stacked), "lists": lists, "sensor": o_sensor}

    def losses(self, T, prepared, flags, need_without_normals):
        rows, vis = [], []
        for j, L in enumerate(prepared["lists"]):
            Tj = T[j:j + 1]
            s_t = orc.transform_points(Tj, L["scan_2"])
            l = orc.icp_losses(s_t, orc.rotate_points(Tj, L["normal_list_2"]), L["scan_1"], L["normal_list_1"],
                               normal_loss="linear" if flags & 8 else "squared", point_to_point=bool(flags & 1),
                               point_to_plane=bool(flags & 2), plane_to_plane=bool(flags & 4), po2po_alone=bool(flags & 16))
            rows.append(torch.stack([l["loss_po2po"].reshape(()), l["loss_po2pl"].reshape(()), l["loss_pl2pl"].reshape(())]))
            vis.append(orc.visible_pixels(s_t, prepared["sensor"]))
        return torch.stack(rows), None, torch.tensor(vis)


def portable_step_inputs(g):
    """Regenerate the inputs of a `step_full_*` / `step_b8_small` fixture from its seeds (delora_amd.data.synthetic's portable generators)
    and prove by sha256 that they are the arrays the reference ran on.  Returns the list of sample dicts ([1,3,n] CPU tensors)."""
    from delora_amd.data import synthetic
    pairs = [synthetic.portable_pair(int(sd), int(g["n_points"])) for sd in g["seeds"]]
    got = synthetic.digest([p[k] for p in pairs for k in ("scan_1", "normal_list_1", "scan_2", "normal_list_2")])
    assert got == str(g["input_sha"]), "the portable generator does not reproduce the fixture's inputs on this machine"
    return [{**{k: torch.from_numpy(p[k]).unsqueeze(0) for k in ("scan_1", "scan_2", "normal_list_1", "normal_list_2")}, "dataset": "kitti"}
            for p in pairs]


def portable_full_state(g):
    """The full network's state_dict of a `step_full_*` fixture, regenerated and sha-checked."""
    from delora_amd.data import synthetic
    shapes = {k[len("shape::"):]: tuple(int(d) for d in v) for k, v in g.items() if k.startswith("shape::")}
    state = synthetic.portable_state_dict(int(g["state_seed"]), shapes)
    assert synthetic.digest([state[k] for k in shapes]) == str(g["state_sha"]), "portable weights differ from the fixture's"
    return {k: torch.from_numpy(v) for k, v in state.items()}
