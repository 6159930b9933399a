#!/usr/bin/env python3
"""Loss trajectory of the full-size step: eager vs captured graph, for a given cnn_impl (modules | auto)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from delora_amd.deploy.trainer import Trainer
from delora_amd.deploy.graph_step import GraphedStep
from delora_amd.data.dataset import ListDataset
impl = sys.argv[1]
dev = torch.device("cuda:0"); torch.cuda.set_device(0)


def setup():
    args = bench.parse(["--batch", "8", "--rotate", "1"])
    cfg = bench.build_config(args, dev); cfg["cnn_impl"] = impl
    torch.manual_seed(1234)
    host = bench.make_batch(args, 0)
    tr = Trainer(cfg, dataset=ListDataset(list(host)))
    bench.identity_pretrained_state(tr.raw_model)
    return tr, bench.to_device(host, dev)


def one(tr, batch):
    tr.optimizer.zero_grad(set_to_none=True)
    ep, _ = tr.step(preprocessed_dicts=[dict(s) for s in batch], epoch_losses=tr.new_epoch_losses())
    return float(ep["loss_epoch"])


if len(sys.argv) > 2 and sys.argv[2] == "nostem":
    from delora_amd.models import ring_conv
    ring_conv.stem_supported = lambda *a, **k: False
if len(sys.argv) > 2 and sys.argv[2] == "nowino":
    from delora_amd.models import ring_conv
    ring_conv.USE_WINOGRAD = False
if len(sys.argv) > 2 and sys.argv[2] == "nocl":
    from delora_amd.models.resnet_modified import ResNetModified
    ResNetModified.trunk_weights_channels_last = lambda self: self
if len(sys.argv) > 2 and sys.argv[2] == "sgd":
    _orig_setup = setup

    def setup():
        tr, b = _orig_setup()
        tr.optimizer = torch.optim.SGD(tr.raw_model.parameters(), lr=1e-3)
        return tr, b
tr, batch = setup()
p3 = None
eager = []
for i in range(4):
    if i == 3:
        p3 = {k: p.detach().clone() for k, p in tr.raw_model.named_parameters()}
    eager.append(one(tr, batch))
g_e = {k: p.grad.detach().clone() for k, p in tr.raw_model.named_parameters()}
p_e = {k: p.detach().clone() for k, p in tr.raw_model.named_parameters()}
tr, batch = setup()
gs = GraphedStep(tr, batch, warmup=3)
ep, _ = gs(); torch.cuda.synchronize()
print(impl, sys.argv[2:], "eager step 4", eager[3], "graph replay 0", float(ep["loss_epoch"]), flush=True)
rows = []
for k, p in tr.raw_model.named_parameters():
    upd_e = p_e[k] - p3[k]
    upd_g = p.detach() - p3[k]
    rows.append((float((upd_g - upd_e).norm() / upd_e.norm().clamp_min(1e-30)), float((p.grad - g_e[k]).norm() / g_e[k].norm().clamp_min(1e-30)), k))
rows.sort(reverse=True)
for r in rows[:6]:
    print("   update rel diff %.3g  grad rel diff %.3g  %s" % r, flush=True)
print("   params with update diff > 1e-3:", sum(r[0] > 1e-3 for r in rows), "of", len(rows), "; grad diff > 1e-3:", sum(r[1] > 1e-3 for r in rows), flush=True)

# localise: the pose the network gives for the static batch with the parameters as they are now (eager forward, no update)
# against the pose the next replay computes from the same parameters
from delora_amd.models.model_parts import GeometryHandler
sensor = tr.img_projection.sensor("kitti")
with torch.no_grad():
    prep = tr.geo.prepare(gs.static_batch, sensor, tr._normal_params("kitti"))
    t_e, q_e = tr.raw_model(prep["stacked"])
    T_e = GeometryHandler.get_transformation_matrix_quaternion(t_e, q_e, dev).clone()
ep, T_g = gs(); torch.cuda.synchronize()
print("   eager-forward pose vs replay-1 pose: max abs diff", float((T_e - T_g).abs().max()), "| replay 1 loss", float(ep["loss_epoch"]), flush=True)
