import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from delora_amd import geometry as G, _lib
from delora_amd.deploy.step_geometry import HipStepGeometry
dev = torch.device("cuda:0")
A = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False))()
cfg = bench.build_config(A, dev); batch = bench.make_batch(A, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
prep = HipStepGeometry().prepare(batch, sensor, (3, 5, 0.5, 10))
T = torch.eye(4, device=dev).repeat(8, 1, 1)
lib = _lib.load()
src, srcn, tpk, tnpk = prep["images"][:, 1], prep["normals"][:, 1], prep["packed"][:, 0], prep["normals_packed"][:, 0]
B, H, W = 8, 64, 2048
nn = torch.empty((B, H, W), dtype=torch.int32, device=dev); match = torch.empty((B, 6, H, W), device=dev)
nbytes = lib.dl_nn_workspace_bytes(B, H, W)
ws = torch.zeros(nbytes // 4 + 4, dtype=torch.int32, device=dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
rc = lib.dl_nn_correspond(vp(src), src.stride(0), vp(srcn), srcn.stride(0), vp(tpk), tpk.stride(0), vp(tnpk), tnpk.stride(0), vp(T), B,
                          ctypes.byref(sensor.struct), 0, vp(nn), vp(match), None, vp(ws), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
cnt = int(ws[0])
active = int((nn >= 0).sum())
print("rc", rc, "hard", cnt, "of active", active, "=%.1f%%" % (100.0 * cnt / active))
rec = ws[64:64 + cnt * 8].view(cnt, 8).cpu().numpy()      # NNHard = 32 bytes = 8 dwords: d2(2) slot idx qx qy qz pad
d2 = rec[:, 0:2].copy().view(np.float64).reshape(-1)
q = rec[:, 4:7].copy().view(np.float32)
nq = np.linalg.norm(q, axis=1)
dinit = np.sqrt(np.minimum(d2, 1e6))
slot = rec[:, 2]
# true distance
m = match.cpu().numpy(); nnc = nn.cpu().numpy().reshape(B, -1)
b = slot // (H * W); px = slot % (H * W)
mt = np.stack([m[b, c].reshape(len(b), -1)[np.arange(len(b)), 0] if False else m[b, c].reshape(B, -1)[b, px] if False else m.reshape(B, 6, -1)[b, c, px] for c in range(3)], axis=1)
dtrue = np.linalg.norm(q - mt, axis=1)
theta_i = np.degrees(np.arcsin(np.clip(dinit / nq, 0, 1))); theta_t = np.degrees(np.arcsin(np.clip(dtrue / nq, 0, 1)))
vres, hres = 26.5 / 63, 359.8 / 2047
def wsize(th): return (2 * th / vres + 2) * (2 * th / hres + 2)
for name, arr in (("d_init", dinit), ("d_true", dtrue), ("|q|", nq), ("win_init(cand)", wsize(theta_i)), ("win_true(cand)", wsize(theta_t))):
    print(f"{name:16s} median {np.median(arr):9.3f} mean {arr.mean():9.3f} p90 {np.percentile(arr, 90):9.3f} p99 {np.percentile(arr, 99):9.3f}")
print("no init candidate:", int((d2 > 1e200).sum()))
print("sum win_init %.3g  sum win_true %.3g" % (wsize(theta_i).clip(0, H * W).sum(), wsize(theta_t).clip(0, H * W).sum()))
