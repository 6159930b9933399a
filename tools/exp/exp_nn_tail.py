import ctypes, os, subprocess, sys, torch
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
T = torch.eye(4, device=dev).repeat(8, 1, 1)
src, srcn, tpk, tnpk = prep["images"][:, 1], prep["normals"][:, 1], prep["packed"][:, 0], prep["normals_packed"][:, 0]
B, H, W = 8, 64, 2048
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for extra in (["-DNN_PIXEL_ROUNDS=1"], ["-DNN_PIXEL_ROUNDS=2"], ["-DNN_PIXEL_ROUNDS=0"]):
    so = f"/tmp/nnt{len(extra)}{extra[0][-1] if extra else 0}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", *extra, "-shared", "-fPIC",
                    os.path.join(csrc, "abi.hip"), os.path.join(csrc, "nn.hip"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.dl_nn_workspace_bytes.restype = ctypes.c_size_t
    nn = torch.empty((B, H, W), dtype=torch.int32, device=dev); match = torch.empty((B, 6, H, W), device=dev)
    ws = torch.zeros(lib.dl_nn_workspace_bytes(B, H, W) // 4 + 4, dtype=torch.int32, device=dev)
    for _ in range(10):
        lib.dl_nn_correspond(vp(src), ctypes.c_int64(src.stride(0)), vp(srcn), ctypes.c_int64(srcn.stride(0)), vp(tpk), ctypes.c_int64(tpk.stride(0)),
                             vp(tnpk), ctypes.c_int64(tnpk.stride(0)), vp(T), B, ctypes.byref(sensor.struct), 0, vp(nn), vp(match), None, vp(ws), st)
    torch.cuda.synchronize()
print("done")
