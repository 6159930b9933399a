import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import util
from tests.util import orc
from delora_amd.data import synthetic
g = util.load_golden("proj_full_digest")
s1 = synthetic.portable_cloud(int(g["seed"]), int(g["N"]))
sensor = util.oracle_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
image, _, _, idx, pix = orc.project_to_img(torch.from_numpy(s1).view(1,3,-1), sensor)
m = -np.ones((sensor.H, sensor.W), dtype=np.int32); p = pix.numpy()[0]; m[p[:,0],p[:,1]] = idx.numpy().astype(np.int32)
gm = g["pix2pt"]
d = np.argwhere(m != gm)
t = util.tainted_pixels(s1, sensor)
print("diff pixels", len(d), "tainted among them", t[d[:,0],d[:,1]].sum())
P = s1.astype(np.float64)
u = (np.arctan2(P[1],P[0]) - sensor.hfov[0])/(sensor.hfov[1]-sensor.hfov[0])*(sensor.W-1)
v = (np.arctan2(P[2],np.hypot(P[0],P[1])) - sensor.vfov[0])/(sensor.vfov[1]-sensor.vfov[0])*(sensor.H-1)
for (r,c) in d[:12]:
    a,b = m[r,c], gm[r,c]
    print((r,c), "box", a, "golden", b, "tainted", t[r,c], [ (float(u[k]),float(v[k]), float(np.linalg.norm(P[:,k]))) for k in (a,b) if k>=0])
