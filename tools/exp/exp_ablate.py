import ctypes, os, subprocess, sys, torch, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "delora_amd", "csrc")
import bench
from delora_amd import geometry as G, _lib
from delora_amd.deploy.step_geometry import HipStepGeometry
dev = torch.device("cuda:0")
A = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))()
cfg = bench.build_config(A, dev); batch = bench.make_batch(A, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
T = torch.eye(4, device=dev).repeat(8, 1, 1); T[:, 0, 3] = 0.4
src, srcn, tpk, tnpk = prep["images"][:, 1], prep["normals"][:, 1], prep["packed"][:, 0], prep["normals_packed"][:, 0]
nn, _ = G.nn_correspond(src, srcn, tpk, T, sensor)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for ab in (1, 2):
    so = f"/tmp/abl{ab}.so"
    srcs = [os.path.join(csrc, f) for f in ("abi.hip", "loss.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", f"-DDL_ABLATE={ab}", "-shared", "-fPIC", *srcs, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lt = torch.zeros(8, 3, device=dev); pc = torch.zeros(8, 2, dtype=torch.int32, device=dev); gt = torch.zeros(8, 36, device=dev)
    ws = torch.zeros(4_000_000, device=dev)
    for _ in range(30):
        lib.dl_icp_loss_fwd(vp(src), ctypes.c_int64(src.stride(0)), vp(srcn), ctypes.c_int64(srcn.stride(0)), vp(tpk), ctypes.c_int64(tpk.stride(0)),
                            vp(tnpk), ctypes.c_int64(tnpk.stride(0)), vp(nn), vp(T), 8, 64, 2048, ctypes.c_uint32(6), vp(lt), vp(pc), vp(gt), vp(ws), st)
    torch.cuda.synchronize()
    # distinguish in the trace by a marker kernel count: ablation 1 runs first
flags = 6
for _ in range(30):
    G.icp_loss(T, src, srcn, tpk, tnpk, nn, flags)
torch.cuda.synchronize()
print("done")
