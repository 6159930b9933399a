import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import util
from tests.test_gpu_geometry import run_project, gpu_sensor
from delora_amd.data import synthetic
g = util.load_golden("proj_full_digest")
s1 = synthetic.portable_cloud(int(g["seed"]), int(g["N"]))
o_sensor = util.oracle_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
sensor = gpu_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
out = run_project([s1], sensor)
m = out["pix2pt"][0].cpu().numpy(); gm = g["pix2pt"]
t = util.tainted_pixels(s1, o_sensor)
d = np.argwhere((m != gm) & ~t)
print("clean diff pixels", len(d), " all diff", (m!=gm).sum())
P = s1.astype(np.float64)
u = (np.arctan2(P[1],P[0]) - o_sensor.hfov[0])/(o_sensor.hfov[1]-o_sensor.hfov[0])*(o_sensor.W-1)
v = (np.arctan2(P[2],np.hypot(P[0],P[1])) - o_sensor.vfov[0])/(o_sensor.vfov[1]-o_sensor.vfov[0])*(o_sensor.H-1)
uv = out["uv"].cpu().numpy()
rp, ru, rv = util.reference_pixels(s1, o_sensor)
for (r,c) in d[:10]:
    a,b = m[r,c], gm[r,c]
    for k in (a,b):
        if k>=0: print((r,c), "gpu" if k==a else "gold", k, "fp64 uv", u[k], v[k], "gpu uv", uv[0][k], uv[1][k], "cpu uv", ru[k], rv[k], "range", np.float32(np.linalg.norm(P[:,k])))
