import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "delora_amd", "csrc")
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry
dev = torch.device("cuda:0")
A = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))()
cfg = bench.build_config(A, dev); batch = bench.make_batch(A, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
T = torch.eye(4, device=dev).repeat(8, 1, 1)
src, srcn, tpk, tnpk = prep["images"][:, 1], prep["normals"][:, 1], prep["packed"][:, 0], prep["normals_packed"][:, 0]
nn, _, match = G.nn_correspond(src, srcn, tpk, tnpk, T, sensor)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for wg, extra in ((128, []), (64, []), (32, []), (256, ["-DLOSS_PX8"])):
    so = f"/tmp/lwg{wg}.so"
    srcs = [os.path.join(csrc, f) for f in ("abi.hip", "loss.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", f"-DLOSS_WG_PER_SAMPLE={wg}", *extra, "-shared", "-fPIC", *srcs, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    ws = torch.zeros(4_000_000, device=dev)
    for _ in range(30):
        lib.dl_icp_loss_partial(vp(src), ctypes.c_int64(src.stride(0)), vp(srcn), ctypes.c_int64(srcn.stride(0)), vp(match), ctypes.c_int64(match.stride(0)),
                                vp(nn), vp(T), 8, 64, 2048, ctypes.c_uint32(6), vp(ws), st)
    torch.cuda.synchronize()
print("done")
