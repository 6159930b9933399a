import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
here = os.path.dirname(os.path.abspath(__file__))
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "exp_loss.hip"), "-o", "/tmp/exp_loss.so"], check=True)
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry
lib = ctypes.CDLL("/tmp/exp_loss.so")
dev = torch.device("cuda:0")
A = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))()
cfg = bench.build_config(A, dev); batch = bench.make_batch(A, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
T = torch.eye(4, device=dev).repeat(8, 1, 1); T[:, 0, 3] = 0.4
nn, _ = G.nn_correspond(prep["images"][:, 1], prep["normals"][:, 1], prep["packed"][:, 0], T, sensor)
out = torch.zeros(8 * 128 * 4, device=dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
sp = prep["images"][:, 1]; sn = prep["normals"][:, 1]; tp = prep["packed"][:, 0]; tn = prep["normals_packed"][:, 0]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in (0, 1, 2):
    for _ in range(30):
        lib.run_var(mode, vp(sp), vp(sn), vp(tp), vp(tn), vp(nn), 64 * 2048, 8, vp(out), st)
torch.cuda.synchronize()
print("done", float(out.sum()))
