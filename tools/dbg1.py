import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import util
from tests.test_gpu_geometry import run_project, gpu_sensor
g = util.load_golden("proj_small")
sensor = gpu_sensor(g["H"], g["W"], g["vfov"], g["hfov"])
scan = g["scan"]
out = run_project([scan], sensor)
img = out["image4"][0].cpu().numpy()
ref = g["image"][0]
d = img[3] != ref[3]
print("range mismatches vs golden:", d.sum(), "of", (ref[3]!=0).sum())
if d.sum():
    a=img[3][d]; b=ref[3][d]
    print(np.abs(a-b).max(), (np.abs(a.view(np.int32)-b.view(np.int32))).max())
rng = torch.norm(torch.from_numpy(scan[:3].copy()).view(1,3,-1), dim=1)[0].numpy()
x64=scan.astype(np.float64)
B = np.sqrt((((x64[0]*x64[0]).astype(np.float32).astype(np.float64) + x64[1]*x64[1]).astype(np.float32).astype(np.float64) + x64[2]*x64[2]).astype(np.float32))
A = np.sqrt((scan[0]*scan[0]+scan[1]*scan[1])+scan[2]*scan[2])
print("on-box torch.norm vs fma-chain:", (rng!=B).sum(), " vs no-fma:", (rng!=A).sum())
idx = out["pix2pt"][0].cpu().numpy().reshape(-1); occ = idx>=0
print("gpu vs fma-chain:", (img[3].reshape(-1)[occ]!=B[idx[occ]]).sum(), "gpu vs nofma", (img[3].reshape(-1)[occ]!=A[idx[occ]]).sum())
print(torch.__config__.show()[:600])
import subprocess; print(subprocess.run("lscpu | head -20", shell=True, capture_output=True, text=True).stdout)
