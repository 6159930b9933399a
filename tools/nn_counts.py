"""List sizes of the exact search on the bench batch (B=8, 64x2048): how many queries pass A leaves to which list of pass B, how many
source tiles become packets.  usage: python tools/nn_counts.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools")
import bench
from delora_amd import geometry as G
import delora_amd.geometry as GG
from delora_amd.deploy.step_geometry import HipStepGeometry
class A: batch=8; height=64; width=2048; point_order="raster"
dev = torch.device("cuda:0")
cfg = bench.build_config(type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))(), dev)
batch = bench.make_batch(A(), 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
geo = HipStepGeometry()
prep = geo.prepare(batch, sensor, (3, 5, 0.5, 10))
img, nrm = prep["images"], prep["normals"]
tpk, tnpk = prep["packed"][:, 0], prep["normals_packed"][:, 0]
orig_empty = torch.empty
keep = {}
def spy(*a, **k):
    t = orig_empty(*a, **k)
    if k.get("dtype") == torch.int64: keep["ws"] = t
    return t
from delora_amd.models.model_parts import GeometryHandler
g = torch.Generator().manual_seed(5)
T_rand = GeometryHandler.get_transformation_matrix_quaternion(torch.randn((8, 3), generator=g), torch.randn((8, 4), generator=g), torch.device("cpu")).to(dev)
for shift in (0.0, 0.4, 1.0, "random"):
    T = torch.eye(4, device=dev).repeat(8, 1, 1)
    if shift == "random": T = T_rand
    else: T[:, 0, 3] = shift
    torch.empty = spy
    nn, vis, match = G.nn_correspond(img[:, 1], nrm[:, 1], tpk, tnpk, T, sensor)
    torch.empty = orig_empty
    torch.cuda.synchronize()
    c = keep["ws"].view(torch.int32)[:5].tolist()
    print(f"shift {shift}: queries {(nn >= 0).sum().item()} lists: wave walk {c[0]}, scans {c[1]}, 16-lane walk {c[2]}, packets {c[3]}")
