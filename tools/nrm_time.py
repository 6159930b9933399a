import sys, os, torch, time
sys.path.insert(0, os.getcwd())
import delora_amd._lib as L
L.LIB_PATH = sys.argv[1]
import bench
from delora_amd import geometry as G
from delora_amd.deploy.step_geometry import HipStepGeometry
dev = torch.device("cuda:0")
X = type("X", (), dict(height=64, width=2048, batch=8, amp="", channels_last=False, point_order="raster"))()
cfg = bench.build_config(X, dev)
batch = bench.make_batch(X, 0, dev)
sensor = G.Sensor.from_config(cfg, "kitti")
geo = HipStepGeometry()
prep = geo.prepare(batch, sensor, (3, 5, 0.5, 10))
img = prep["images"]
torch.cuda.synchronize()
x = img.reshape(-1, 4, 64, 2048)
for _ in range(5): G.normals(x, 3, 5, 0.5, 10)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): G.normals(x, 3, 5, 0.5, 10)
b.record(); torch.cuda.synchronize()
print(sys.argv[1].split("/")[-4] if "ref" in sys.argv[1] else "product", f"{a.elapsed_time(b) / 50 * 1e3:.1f} us per dl_normals (16 images 64x2048)")
