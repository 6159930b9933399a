import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "exp_streams.hip"), "-o", "/tmp/exp_streams.so"], check=True)
lib = ctypes.CDLL("/tmp/exp_streams.so")
dev = torch.device("cuda:0")
HW, B = 64 * 2048, 8
buf = torch.randn(B * 13 * HW, device=dev)
other = torch.randn(64_000_000, device=dev)           # evict caches between launches
out = torch.zeros(B * 128 * 4, device=dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in (0, 1, 2):
    for _ in range(20):
        other.add_(1.0)                                # 512 MB of traffic: flushes L2 and the 256 MB infinity cache
        lib.run_streams(mode, vp(buf), HW, B, vp(out), st)
    for _ in range(20):                                # back to back: data may sit in the infinity cache
        lib.run_streams(mode, vp(buf), HW, B, vp(out), st)
torch.cuda.synchronize()
print("done")
