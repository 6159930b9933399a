"""The channels-last fp32 MFMA convolutions of the pose CNN (delora_amd/csrc/conv.hip through the C ABI) against plain torch
fp32 ops: ``F.conv2d(F.pad(x, (1,1,0,0), 'circular'), w, padding=(1,0))`` -- the reference's layer (src/models/
resnet_modified.py:97-98, :162-168) -- forward, both backward passes and the fused epilogues; and the whole residual trunk
(ring_conv.RingTrunk) against the module path on the same weights.  Tolerance 1e-4 relative (fp32 summation order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
REL = 1e-4          # the contract of north_star
TIGHT = 1e-5        # what the operator-level comparisons assert: ~10x the largest deviation measured on MI355X (<= 1e-6)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _ref_conv(x_nchw, w, stride, ks):
    if ks == 3:
        return F.conv2d(F.pad(x_nchw, (1, 1, 0, 0), mode="circular"), w, stride=stride, padding=(1, 0))
    return F.conv2d(x_nchw, w, stride=stride)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


SHAPES = [  # N, H, W, C, K, ks, stride
    (2, 8, 128, 64, 64, 3, (1, 1)), (1, 4, 64, 128, 128, 3, (1, 1)), (2, 8, 32, 64, 128, 3, (1, 1)),
    (1, 8, 256, 256, 64, 3, (1, 1)), (2, 8, 256, 64, 128, 3, (1, 2)), (1, 8, 128, 64, 64, 3, (2, 2)),
    (2, 8, 128, 64, 128, 1, (1, 2)), (1, 8, 128, 64, 64, 1, (2, 2)),
    # images that do not divide into tiles (round 4): the feature maps of the reference's shipped 64x720 image are 180, 90, 45 and 23
    # pixels wide, layer4 of 64x512 is 32x16; odd widths / heights end in partial tiles, strides round the output size up
    (2, 8, 180, 64, 64, 3, (1, 1)), (1, 6, 90, 64, 128, 3, (1, 1)), (2, 5, 45, 64, 64, 3, (1, 1)), (1, 7, 23, 128, 64, 3, (1, 1)),
    (2, 8, 180, 64, 128, 3, (1, 2)), (1, 8, 90, 128, 64, 3, (1, 2)), (2, 8, 45, 64, 128, 3, (2, 2)), (1, 7, 45, 64, 64, 3, (1, 2)),
    (2, 8, 45, 64, 128, 1, (2, 2)), (1, 5, 90, 64, 64, 1, (1, 2)), (1, 16, 16, 64, 64, 3, (1, 1)), (1, 3, 24, 64, 64, 3, (2, 2)),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_forward_and_both_gradients_against_torch(shape):
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    N, H, W, C, K, ks, stride = shape
    g = torch.Generator(device="cpu").manual_seed(sum(shape[:5]))
    x = torch.randn((N, C, H, W), generator=g).to(dev)
    w = (torch.randn((K, C, ks, ks), generator=g) * 0.1).to(dev)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    w_krsc = w.permute(0, 2, 3, 1).contiguous()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = _ref_conv(xr, wr, stride, ks)
    gy = torch.randn(y_ref.shape, generator=g).to(dev)
    y_ref.backward(gy)
    tag = f"conv {ks}x{ks} s{stride} {N}x{H}x{W} {C}->{K}"
    # forward, plain and with the fused tail  act(conv + shortcut)
    y = rc.conv_nhwc(x_nhwc, w_krsc, stride=stride)
    util.measured(f"{tag}: forward vs torch (relative)", _rel(y.permute(0, 3, 1, 2), y_ref.detach()), bound=TIGHT)
    sc = torch.randn(y.shape, generator=g).to(dev)
    y2 = rc.conv_nhwc(x_nhwc, w_krsc, stride=stride, act=rc.ACT["tanh"], epilogue=rc.EPI_ADD | rc.EPI_ACT, add=sc)
    ref2 = torch.tanh(y_ref.detach() + sc.permute(0, 3, 1, 2))
    util.measured(f"{tag}: forward + shortcut + tanh vs torch (absolute)", float((y2.permute(0, 3, 1, 2) - ref2).abs().max()), bound=5e-5)
    # weight gradient
    gy_nhwc = gy.permute(0, 2, 3, 1).contiguous()
    dw = rc.wgrad_nhwc(x_nhwc, gy_nhwc, ks, stride=stride)
    util.measured(f"{tag}: weight gradient vs torch autograd (relative)", _rel(dw.permute(0, 3, 1, 2), wr.grad), bound=TIGHT)
    # input gradient (stride-1 3x3: the transposed kernel, with the activation derivative and shortcut gradient fused)
    if ks == 3 and stride == (1, 1) and C % 64 == 0:
        dx = rc.conv_nhwc(gy_nhwc, w_krsc, transposed=True)
        util.measured(f"{tag}: input gradient vs torch autograd (relative)", _rel(dx.permute(0, 3, 1, 2), xr.grad), bound=TIGHT)
        ysave = torch.tanh(torch.randn(x_nhwc.shape, generator=g)).to(dev)
        extra = torch.randn(x_nhwc.shape, generator=g).to(dev)
        dx2 = rc.conv_nhwc(gy_nhwc, w_krsc, act=rc.ACT["tanh"], epilogue=rc.EPI_ADD | rc.EPI_DACT, add=extra, dsrc=ysave, transposed=True)
        ref = (xr.grad.permute(0, 2, 3, 1) + extra) * (1 - ysave * ysave)
        util.measured(f"{tag}: fused (dgrad + g) * tanh' vs torch (relative)", _rel(dx2, ref), bound=TIGHT)
    if stride != (1, 1):
        # strided layers: one pass per stride phase (and, for an odd width, the two seam terms), plain and with the 1x1 branch's
        # gradient added on the grid and the activation derivative fused
        dense = ks == 1
        dx = rc.dgrad_strided(gy_nhwc, w_krsc, stride, (H, W), dense=dense)
        ref_dx = xr.grad.permute(0, 2, 3, 1)
        if dense:
            ref_dx = ref_dx[:, ::stride[0], ::stride[1]]
        util.measured(f"{tag}: strided input gradient vs torch autograd (relative)", _rel(dx, ref_dx), bound=TIGHT)
        if not dense:
            addg = torch.randn((N, y.shape[1], y.shape[2], C), generator=g).to(dev)
            ysave = torch.tanh(torch.randn(x_nhwc.shape, generator=g)).to(dev)
            dx2 = rc.dgrad_strided(gy_nhwc, w_krsc, stride, (H, W), act=rc.ACT["tanh"], epilogue=rc.EPI_ADD_GRID | rc.EPI_DACT, add_grid=addg, dsrc=ysave)
            full = torch.zeros_like(ref_dx)
            full[:, ::stride[0], ::stride[1]] = addg
            util.measured(f"{tag}: fused (strided dgrad + grid gradient) * tanh' vs torch (relative)",
                          _rel(dx2, (ref_dx + full) * (1 - ysave * ysave)), bound=TIGHT)


@pytest.mark.parametrize("shape", [(2, 8, 128, 64, 64), (1, 4, 64, 128, 128), (2, 8, 32, 64, 128), (1, 8, 256, 64, 64), (1, 32, 64, 512, 512),
                                   # overhanging tile groups and partial tiles (64x720: 180 / 90 / 45 / 23 wide; 64x512: layer4 32x16)
                                   (2, 64, 180, 64, 64), (1, 64, 90, 128, 128), (2, 64, 45, 64, 64), (1, 32, 23, 128, 64), (1, 32, 16, 64, 64),
                                   (1, 7, 5, 64, 64), (1, 1, 2, 64, 64)])
def test_winograd_forward_and_input_gradient_against_torch(shape):
    """Fused Winograd F(2x2,3x3) (csrc/wino.hip) against torch's direct convolution of the wrapped image: forward with the
    fused shortcut + tanh, input gradient with the fused activation derivative; all three tile layouts."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    N, H, W, C, K = shape
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn((N, C, H, W), generator=g).to(dev)
    w = (torch.randn((K, C, 3, 3), generator=g) * (0.5 / np.sqrt(9 * C))).to(dev).contiguous(memory_format=torch.channels_last)
    assert rc.wino_ok(H, W, C, K)
    xr = x.clone().requires_grad_(True)
    y_ref = _ref_conv(xr, w, (1, 1), 3)
    gy = torch.randn(y_ref.shape, generator=g).to(dev)
    y_ref.backward(gy)
    uf, ub = rc.wino_weights(w)
    x_nhwc, gy_nhwc = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
    tag = f"winograd {N}x{H}x{W} {C}->{K}"
    y = rc.wino_conv(x_nhwc, uf, K)
    util.measured(f"{tag}: forward vs torch direct convolution (relative)", _rel(y.permute(0, 3, 1, 2), y_ref.detach()), bound=TIGHT)
    sc = torch.randn(y.shape, generator=g).to(dev)
    y2 = rc.wino_conv(x_nhwc, uf, K, act=rc.ACT["tanh"], epilogue=rc.EPI_ADD | rc.EPI_ACT, add=sc)
    util.measured(f"{tag}: forward + shortcut + tanh vs torch (absolute)",
                  float((y2.permute(0, 3, 1, 2) - torch.tanh(y_ref.detach() + sc.permute(0, 3, 1, 2))).abs().max()), bound=2e-5)
    dx = rc.wino_conv(gy_nhwc, ub, C)
    util.measured(f"{tag}: input gradient vs torch autograd (relative)", _rel(dx.permute(0, 3, 1, 2), xr.grad), bound=TIGHT)
    ysave = torch.tanh(torch.randn(x_nhwc.shape, generator=g)).to(dev)
    dx2 = rc.wino_conv(gy_nhwc, ub, C, act=rc.ACT["tanh"], epilogue=rc.EPI_DACT, dsrc=ysave)
    util.measured(f"{tag}: fused dgrad * tanh' vs torch (relative)", _rel(dx2, xr.grad.permute(0, 2, 3, 1) * (1 - ysave * ysave)), bound=TIGHT)
    # and against this library's own direct kernel (same layout, same epilogue)
    y_direct = rc.conv_nhwc(x_nhwc, rc.weight_storage(w))
    util.measured(f"{tag}: Winograd vs the direct MFMA kernel (relative)", _rel(y, y_direct), bound=TIGHT)


def test_conv_rejects_channel_counts_that_do_not_tile():
    """Image sizes are free since round 4 (tiles hang over the edges); channel counts must still be multiples of the 64-channel
    output block / 8-channel reduction chunk -- narrower test networks take the module path."""
    from delora_amd import _lib
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    x = torch.zeros((1, 4, 24, 64), device=dev)
    with pytest.raises(_lib.DeloraHipError):
        rc.conv_nhwc(x, torch.zeros((48, 3, 3, 64), device=dev))          # 48 output channels
    with pytest.raises(_lib.DeloraHipError):
        rc.conv_nhwc(torch.zeros((1, 4, 24, 60), device=dev), torch.zeros((64, 3, 3, 60), device=dev))
    assert rc.supported((1, 4, 24, 64), ((64, 64, (1, 1), False),)) and rc.supported((8, 64, 180, 64), ((64, 128, (1, 2), True),))
    assert not rc.supported((1, 4, 24, 32), ((32, 32, (1, 1), False),))


@pytest.mark.parametrize("act,wino,size", [("tanh", True, (16, 1024)), ("tanh", False, (16, 1024)), ("relu", True, (16, 1024)),
                                           # the reference's shipped image sizes (config/config_datasets.yaml:21, :47) and odd ones
                                           ("tanh", True, (64, 720)), ("tanh", False, (64, 720)), ("tanh", True, (64, 512)),
                                           ("relu", True, (18, 360)), ("tanh", True, (7, 100))])
def test_hip_trunk_matches_module_path(act, wino, size, monkeypatch):
    """The full-width network (64..512 channels) on a 16x1024 pair -- and on the reference's shipped 64x720 / 64x512 images, whose
    feature maps do not divide into tiles (720: 180, 90, 45, 23 wide: odd widths, a stride-2 layer on an odd width) -- : the
    channels-last HIP stem + trunk against the module path (library convolutions + ring ops) with the same weights: poses, and the
    gradient of EVERY parameter."""
    from delora_amd.models import ring_conv
    from delora_amd.models.model import OdometryModel
    dev = _dev()
    monkeypatch.setattr(ring_conv, "USE_WINOGRAD", wino)
    cfg = util.repo_config(size[0], size[1], device="cuda:0", activation_fct=act)
    torch.manual_seed(5)
    m_hip = OdometryModel(dict(cfg, cnn_impl="hip")).to(dev)
    assert m_hip.resnet.hip_path_takes(size[0], size[1], batch=2)
    m_hip.resnet.trunk_weights_channels_last()
    m_mod = OdometryModel(dict(cfg, cnn_impl="modules")).to(dev)
    m_mod.load_state_dict(m_hip.state_dict())
    x = torch.randn((2, 8, size[0], size[1]), device=dev)
    out = []
    for m in (m_hip, m_mod):
        t, q = m(x)
        (t.square().sum() + (q * torch.arange(1, 5, device=dev)).sum()).backward()
        out.append((t.detach(), q.detach()))
    tagn = f"trunk[{act}{',winograd' if wino else ',direct'}{'' if size == (16, 1024) else ',%dx%d' % size}]"
    util.measured(f"{tagn}: translation hip vs modules (relative)", _rel(out[0][0], out[1][0]), bound=1e-5)       # measured 4-8e-7
    util.measured(f"{tagn}: quaternion hip vs modules (relative)", _rel(out[0][1], out[1][1]), bound=1e-5)
    worst, name = 0.0, ""
    errs = []
    referee = {}
    if size == (64, 720):
        # At this size the LIBRARY's weight gradient of the 8-channel conv1 (module path) deviates from this library's by 4.2e-4
        # (measured in round 4; every other parameter agrees to 1e-5).  Referee: the same network under torch autograd in float64 on
        # the CPU -- the HIP stem must agree with it, the module path's deviation is recorded.
        m_cpu = OdometryModel(dict(util.repo_config(size[0], size[1], device="cpu", activation_fct=act), cnn_impl="modules")).double()
        m_cpu.load_state_dict({k: v.detach().cpu().double() for k, v in m_hip.state_dict().items()})
        t, q = m_cpu(x.cpu().double())
        (t.square().sum() + (q * torch.arange(1, 5, dtype=torch.float64)).sum()).backward()
        referee = {"resnet.conv1.weight": m_cpu.resnet.conv1.weight.grad.float().to(dev)}
    for (k, p), (_, p2) in zip(m_hip.named_parameters(), m_mod.named_parameters()):
        assert p.grad is not None and p.grad.shape == p.shape, k
        if k in referee:
            ref = referee[k]
            util.measured(f"{tagn}: {k} gradient, HIP stem vs torch-CPU float64 (relative)", float((p.grad - ref).norm() / ref.norm()), bound=5e-5)
            util.measured(f"{tagn}: {k} gradient, module path (library convolution) vs torch-CPU float64 (relative)", float((p2.grad - ref).norm() / ref.norm()))
            continue
        e = float((p.grad - p2.grad).norm() / p2.grad.norm().clamp_min(1e-30))
        errs.append(e)
        if e > worst:
            worst, name = e, k
        if e > 2e-5:
            print(f"  gradient of {k}: relative difference {e:.3e}")
    # measured 4-7e-6 (tanh) and 7-9e-7 (relu) -- as long as no relu mask flips.  The module path's library convolutions differ from run
    # to run in their last bits (the poses of this very test vary between 6e-7 and 9e-7 across runs), and ONE pre-activation near zero
    # that the two paths round to different sides is a flipped mask: the gradient of every weight upstream of it then differs by 2-3e-4
    # (1.3e-3 for conv1), everything downstream stays at 1e-6 -- seen once in round 5 (a flip in layer3.0).  The relu bound covers
    # isolated flips; a wrong kernel shows up in the tanh cases (no discontinuity) at 5e-5 and in the operator tests.
    # (advisor, round 5) a flip disturbs the parameters UPSTREAM of it only, so the best-agreeing quarter of the parameters stays at the tight
    # bound whatever flips -- a wrong relu kernel or `dact` epilogue would move all of them; the relu epilogues themselves are compared
    # with torch operator by operator in test_winograd_relu_epilogues_against_torch (no second path, no flips)
    util.measured(f"{tagn}: worst relative parameter-gradient difference hip vs modules ({name})", worst, bound=(5e-5 if act == "tanh" else 5e-3))
    util.measured(f"{tagn}: 25th percentile of the relative parameter-gradient differences hip vs modules", float(np.quantile(errs, 0.25)), bound=1e-5)
    assert m_hip.resnet.layer1[0].conv1.weight.grad.stride() == m_hip.resnet.layer1[0].conv1.weight.stride()


@pytest.mark.parametrize("shape", [(2, 16, 64, 64, 64), (1, 8, 128, 128, 64), (1, 6, 36, 64, 128)])
def test_winograd_relu_epilogues_against_torch(shape):
    """The relu epilogues of the fused Winograd kernel (`wn_act` / `wn_dact`, the generic store loop) against torch, operator by operator:
    forward + shortcut + relu, and the input gradient times relu'(saved activation) with a fused shortcut gradient.  The masks come from
    tensors BOTH sides are given, so no pre-activation can fall on different sides of zero."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    N, H, W, C, K = shape
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + 1)
    x = torch.randn((N, C, H, W), generator=g).to(dev)
    w = (torch.randn((K, C, 3, 3), generator=g) * (0.5 / np.sqrt(9 * C))).to(dev).contiguous(memory_format=torch.channels_last)
    xr = x.clone().requires_grad_(True)
    y_ref = _ref_conv(xr, w, (1, 1), 3)
    gy = torch.randn(y_ref.shape, generator=g).to(dev)
    y_ref.backward(gy)
    uf, ub = rc.wino_weights(w)
    x_nhwc, gy_nhwc = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
    sc = torch.randn((N, H, W, K), generator=g).to(dev)
    tag = f"winograd relu {N}x{H}x{W} {C}->{K}"
    y = rc.wino_conv(x_nhwc, uf, K, act=rc.ACT["relu"], epilogue=rc.EPI_ADD | rc.EPI_ACT, add=sc)
    pre = y_ref.detach().permute(0, 2, 3, 1) + sc
    safe = pre.abs() > 1e-4                                  # (away from the kink the two evaluations agree on the side)
    util.measured(f"{tag}: forward + shortcut + relu vs torch (absolute, |pre-activation| > 1e-4)",
                  float(((y - torch.relu(pre)).abs() * safe).max()), bound=2e-5)
    ysave = torch.relu(torch.randn(x_nhwc.shape, generator=g)).to(dev)             # a saved relu output: zeros and positives
    gsc = torch.randn(x_nhwc.shape, generator=g).to(dev)
    dx = rc.wino_conv(gy_nhwc, ub, C, act=rc.ACT["relu"], epilogue=rc.EPI_ADD | rc.EPI_DACT, add=gsc, dsrc=ysave)
    want = (xr.grad.permute(0, 2, 3, 1) + gsc) * (ysave > 0)
    util.measured(f"{tag}: fused (dgrad + shortcut gradient) * relu' vs torch (relative)", _rel(dx, want), bound=TIGHT)
    assert bool((dx[ysave <= 0] == 0).all())


def test_hip_trunk_is_used_by_the_full_size_step_and_falls_back_for_small_networks():
    from delora_amd.models.model import OdometryModel
    dev = _dev()
    big = OdometryModel(util.repo_config(64, 2048, device="cuda:0")).to(dev)
    assert big.resnet.hip_trunk_applicable((8, 64, 512, 64), torch.zeros(1, device=dev))
    small = OdometryModel(util.repo_config(16, 128, device="cuda:0", factor_fewer_resnet_channels=8, resnet_outputs=64)).to(dev)
    assert not small.resnet.hip_trunk_applicable((2, 16, 32, 8), torch.zeros(1, device=dev))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not big.resnet.hip_trunk_applicable((8, 64, 512, 64), torch.zeros(1, device=dev))


def _ref_stem(x, w1, act):
    a = _ref_conv(x, w1, (1, 2), 3)
    a = torch.tanh(a) if act == "tanh" else torch.relu(a)
    return F.max_pool2d(F.pad(a, (1, 1, 0, 0), mode="circular"), kernel_size=3, stride=(1, 2), padding=(1, 0))


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_stem_pooling_kernels_against_torch(act):
    """dl_pool3x3s12_nhwc_fwd / _bwd against F.max_pool2d of the wrapped map and torch autograd through act + pooling:
    values bit-equal, gradient with respect to the pre-activation within rounding of the sums."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(3)
    pre = torch.randn((2, 8, 6, 64), generator=g).to(dev).requires_grad_(True)            # NCHW
    pre.data[0, :, 0, :4] = 3.0                                                          # ties: the first position must win
    a = torch.tanh(pre) if act == "tanh" else torch.relu(pre)
    y_ref = F.max_pool2d(F.pad(a, (1, 1, 0, 0), mode="circular"), kernel_size=3, stride=(1, 2), padding=(1, 0))
    gy = torch.randn(y_ref.shape, generator=g).to(dev)
    y_ref.backward(gy)
    a_nhwc = a.detach().permute(0, 2, 3, 1).contiguous()
    y, win = rc.pool_fwd(a_nhwc)
    assert torch.equal(y.permute(0, 3, 1, 2), y_ref.detach())
    assert int(win.min()) >= 0 and int(win.max()) <= 8
    gc = rc.pool_bwd(gy.permute(0, 2, 3, 1).contiguous(), a_nhwc, win, rc.ACT[act])
    util.measured(f"stem pooling[{act}]: gradient w.r.t. the pre-activation vs torch autograd (absolute)",
                  float((gc.permute(0, 3, 1, 2) - pre.grad).abs().max()), bound=2e-6)


@pytest.mark.parametrize("act,size", [("tanh", (16, 1024)), ("relu", (16, 1024)), ("tanh", (16, 720)), ("relu", (9, 100)), ("tanh", (64, 720))])
def test_stem_function_against_torch(act, size):
    """RingStem (transposing copy + MFMA conv1 with the activation in the epilogue + pooling; backward: pooling gather +
    weight gradient) against F.pad(circular) + conv2d + act + F.pad(circular) + max_pool2d under torch autograd."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(9)
    x = (torch.randn((2, 8, size[0], size[1]), generator=g) * 3.0).to(dev)
    w1 = (torch.randn((64, 8, 3, 3), generator=g) * 0.05).to(dev).requires_grad_(True)
    assert rc.stem_supported(tuple(x.shape), 64)
    y = rc.RingStem.apply(x, w1, rc.ACT[act])                                            # [N,H,W/4,64]
    gy = torch.randn(y.shape, generator=g).to(dev)
    y.backward(gy)
    dw = w1.grad.clone()
    w1.grad = None
    # reference: the same ops under torch autograd in float64 on the CPU (at 64x720, batch 2 the GPU library's own weight gradient of this
    # 8-channel layer deviates by 4e-4 from this kernel, which agrees with float64 to 4e-7 -- so the library is not the yardstick here)
    xr = x.detach().cpu().double().requires_grad_(True)
    w1r = w1.detach().cpu().double().requires_grad_(True)
    y_ref = _ref_stem(xr, w1r, act)
    y_ref.backward(gy.permute(0, 3, 1, 2).cpu().double())
    y_ref, xr_grad, w1r_grad = y_ref.detach().float().to(dev), xr.grad.float().to(dev), w1r.grad.float().to(dev)
    util.measured(f"stem[{act},{size[0]}x{size[1]}]: pooled output vs torch (absolute)", float((y.permute(0, 3, 1, 2) - y_ref).abs().max()), bound=2e-5)
    util.measured(f"stem[{act},{size[0]}x{size[1]}]: conv1 weight gradient vs torch autograd (relative)", _rel(dw, w1r_grad), bound=TIGHT)     # measured 3-8e-7 (relu: no mask flips on this seeded input)
    # and the input gradient (not needed by the training step: the image carries none)
    x2 = x.clone().requires_grad_(True)
    rc.RingStem.apply(x2, w1, rc.ACT[act]).backward(gy)
    util.measured(f"stem[{act},{size[0]}x{size[1]}]: input gradient vs torch autograd (relative)", _rel(x2.grad, xr_grad), bound=TIGHT)     # measured 3-8e-7 (relu: no mask flips on this seeded input)


def test_mean_hw_kernel_against_torch():
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn((3, 8, 64, 512), generator=g).to(dev).requires_grad_(True)
    y = rc.MeanHW.apply(x)
    gy = torch.randn(y.shape, generator=g).to(dev)
    y.backward(gy)
    gx = x.grad.clone()
    x.grad = None
    y_ref = x.mean(dim=(1, 2))
    y_ref.backward(gy)
    util.measured("global average pooling (channels-last): kernel vs torch.mean (absolute)", float((y - y_ref).abs().max()), bound=1e-6)
    assert torch.allclose(gx, x.grad, rtol=0, atol=1e-9)


@pytest.mark.parametrize("shape", [(2, 8, 64, 128, 128), (1, 4, 32, 256, 64), (2, 8, 90, 128, 128), (1, 7, 45, 128, 64), (1, 32, 23, 128, 128), (1, 3, 16, 128, 64)])
def test_winograd_domain_weight_gradient_against_the_direct_kernel_and_torch(shape, monkeypatch):
    """dl_wino_wgrad3x3_nhwc_f32 (weight gradient accumulated in the Winograd domain, csrc/wino.hip) against this library's direct
    kernel and torch autograd on the same data, incl. image rows at the zero-padded border and the wrap-around columns."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    N, H, W, C, K = shape
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn((N, C, H, W), generator=g).to(dev)
    w = (torch.randn((K, C, 3, 3), generator=g) * 0.05).to(dev).requires_grad_(True)
    y = _ref_conv(x, w, (1, 1), 3)
    gy = torch.randn(y.shape, generator=g).to(dev)
    y.backward(gy)
    x_nhwc, gy_nhwc = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
    monkeypatch.setattr(rc, "USE_WINOGRAD_WGRAD", True)
    dw_wino = rc.wgrad_nhwc(x_nhwc, gy_nhwc, 3)
    monkeypatch.setattr(rc, "USE_WINOGRAD_WGRAD", False)
    dw_direct = rc.wgrad_nhwc(x_nhwc, gy_nhwc, 3)
    tag = f"winograd-domain wgrad {N}x{H}x{W} {C}->{K}"
    util.measured(f"{tag}: vs torch autograd (relative)", _rel(dw_wino.permute(0, 3, 1, 2), w.grad), bound=TIGHT)
    util.measured(f"{tag}: vs the direct MFMA kernel (relative)", _rel(dw_wino, dw_direct), bound=TIGHT)
    assert not torch.equal(dw_wino, dw_direct)          # the two paths really are different kernels


def test_batched_weight_gradients_equal_the_single_launches():
    """dl_wino_wgrad3x3_batch_nhwc_f32 / dl_conv2d_wgrad_batch_nhwc_f32 (a run of layers in merged launches with fewer pixel slabs
    each; ring_conv.wgrad_batch routes as wgrad_nhwc does) against the single-layer entry points: the same products, only the slab
    boundaries of the fp32 sums differ.  Winograd-domain and direct kernels, strided and 1x1 layers, images that do not divide; a batch
    of one (its own slab plan) within the same tolerance."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    shapes = [(2, 16, 128, 128, 128, 3, (1, 1)), (2, 8, 64, 256, 256, 3, (1, 1)), (1, 8, 32, 256, 128, 3, (1, 1)), (2, 16, 128, 64, 64, 3, (1, 1)),
              (2, 16, 64, 64, 128, 3, (1, 1)), (2, 8, 256, 64, 128, 3, (1, 2)), (2, 16, 128, 128, 256, 3, (2, 2)), (2, 8, 128, 64, 128, 1, (1, 2)),
              (2, 8, 64, 128, 256, 1, (2, 2)), (2, 8, 90, 128, 128, 3, (1, 1)), (2, 5, 45, 128, 128, 3, (1, 1)), (2, 8, 45, 64, 128, 3, (2, 2)),
              (2, 8, 180, 64, 64, 3, (1, 1))]
    items = []
    for (N, H, W, C, K, ks, st) in shapes:
        x = torch.randn((N, H, W, C), generator=g).to(dev)
        gy = torch.randn((N, rc.out_size(H, st[0]), rc.out_size(W, st[1]), K), generator=g).to(dev)
        items.append((x, gy, ks, st))
    batched = rc.wgrad_batch(items)
    for (x, gy, ks, st), dw_b, shp in zip(items, batched, shapes):
        dw_1 = rc.wgrad_nhwc(x, gy, ks, stride=st)
        scale = float(dw_1.abs().max())
        util.measured(f"batched vs single fp32 weight gradient {shp}: max |diff| / max |dw|", float((dw_b - dw_1).abs().max()) / scale, bound=3e-6)
        assert float((rc.wgrad_batch([(x, gy, ks, st)])[0] - dw_1).abs().max()) <= 3e-6 * scale, shp


def test_epilogue_tanh_is_within_two_ulp_of_float64():
    """The activation of the convolution epilogues (csrc/common.h: dl_tanh -- branch-free, exp2 / rcp above |x| = 0.625, an odd
    polynomial below) measured in isolation: a 1x1 convolution with the identity as its weight reproduces its input exactly (one
    product with 1 and sums with exact zeros on the fp32 matrix cores), so y = tanh(x) as the kernels compute it.  Bound: 2.5 ulp of
    the float32 result against numpy's float64 tanh over [-12, 12] and down to 1e-30; NaN propagates."""
    dev = _dev()
    from delora_amd.models import ring_conv as rc
    C = 64
    g = torch.Generator(device="cpu").manual_seed(9)
    H, W = 16, 256
    x = (torch.rand((1, H, W, C), generator=g) * 24.0 - 12.0)
    x[0, 0] = torch.linspace(-0.7, 0.7, W * C).view(W, C)                       # dense around the switch point of the two formulas
    x[0, 1] = torch.logspace(-30, 0, W * C).view(W, C) * torch.where(torch.rand((W, C), generator=g) < 0.5, -1.0, 1.0)
    x[0, 2, 0, 0] = 0.0
    x[0, 2, 0, 1] = -0.0                            # (reaches the activation as +0: the sum with the other channels' +0 products)
    x[0, 2, 2, :] = float("nan")                    # a whole pixel: NaN x 0 in the identity GEMM spreads over the pixel's channels
    w = torch.eye(C).view(C, 1, 1, C).contiguous()
    y = rc.conv_nhwc(x.to(dev), w.to(dev), stride=(1, 2), act=rc.ACT["tanh"], epilogue=rc.EPI_ACT).cpu()
    xs = x[:, :, ::2, :]
    ref = np.tanh(xs.double().numpy())
    got = y.double().numpy()
    nan = np.isnan(ref)
    assert np.isnan(got[nan]).all() and not np.isnan(got[~nan]).any()
    ulp = np.spacing(np.abs(ref[~nan]).astype(np.float32)).astype(np.float64)
    err = np.abs(got[~nan] - ref[~nan]) / ulp
    util.measured("epilogue tanh vs float64 (ulp of the float32 result)", float(err.max()), bound=2.5)
    assert got[0, 2, 0, 0] == 0.0 and got[0, 2, 0, 1] == 0.0


@pytest.mark.parametrize("act,B", [("tanh", 8), ("relu", 3), ("tanh", 1), ("tanh", 16)])
def test_fused_heads_against_the_torch_modules(act, B):
    """csrc/heads.hip (fc -> two two-layer heads -> whole-batch quaternion norm; 3 launches forward, 4 backward) against the same
    layers as torch modules under autograd (reference src/models/model.py:74-83, :114): outputs, the gradient of the pooled feature and
    of all ten parameters."""
    from delora_amd.models.model import OdometryModel
    from delora_amd.models import model_parts
    dev = _dev()
    cfg = util.repo_config(64, 2048, device="cuda:0", activation_fct=act)
    torch.manual_seed(3)
    m = OdometryModel(cfg).to(dev)
    g = torch.Generator(device="cpu").manual_seed(B)
    feat = torch.randn((B, 512), generator=g).to(dev)
    gt, gr = torch.randn((B, 3), generator=g).to(dev), torch.randn((B, 4), generator=g).to(dev)
    names = ["resnet.fc.weight", "resnet.fc.bias", "fully_connected_rotation.1.weight", "fully_connected_rotation.1.bias",
             "fully_connected_rotation.3.weight", "fully_connected_rotation.3.bias", "fully_connected_translation.1.weight",
             "fully_connected_translation.1.bias", "fully_connected_translation.3.weight", "fully_connected_translation.3.bias"]
    params = dict(m.named_parameters())
    # torch modules
    x1 = feat.clone().requires_grad_(True)
    t1, r1 = m._heads(m.resnet.fc(x1))
    ((t1 * gt).sum() + (r1 * gr).sum()).backward()
    ref = {n: params[n].grad.clone() for n in names}
    gx_ref = x1.grad.clone()
    m.zero_grad(set_to_none=True)
    # fused
    x2 = feat.clone().requires_grad_(True)
    t2, r2 = model_parts.FusedHeads.apply(x2, 2 if act == "relu" else 1, *[params[n] for n in names])
    ((t2 * gt).sum() + (r2 * gr).sum()).backward()
    tag = f"fused heads[{act},B={B}]"
    util.measured(f"{tag}: translation vs torch modules (relative)", _rel(t2, t1.detach()), bound=TIGHT)
    util.measured(f"{tag}: rotation vs torch modules (relative)", _rel(r2, r1.detach()), bound=TIGHT)
    util.measured(f"{tag}: gradient of the pooled feature (relative)", _rel(x2.grad, gx_ref), bound=TIGHT)
    worst = max(_rel(params[n].grad, ref[n]) for n in names)
    util.measured(f"{tag}: worst parameter gradient vs torch autograd (relative)", worst, bound=TIGHT)


def test_batched_weight_transform_is_the_single_layer_transform_bit_for_bit():
    """`dl_wino_weights_batch_f32` (one launch per autograd segment; LDS-staged so that its stores are contiguous runs) against
    `dl_wino_weights_f32` (one thread per (k, c), scattered 4-byte stores): the same arithmetic per element, so the Winograd-domain
    weights must be EQUAL -- for every channel pair of the trunk, with and without the backward set, and for a layer whose input
    channels do not divide into the batched kernel's 32-channel blocks (plain launches)."""
    from delora_amd.models import ring_conv as rc
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(64, 64), (128, 64), (128, 128), (256, 128), (256, 256), (512, 256), (512, 512)]
    ws = [(torch.randn((k, c, 3, 3), generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last) for k, c in shapes]
    for want_bwd in (True, False):
        got = rc.wino_weights_batch(ws, want_bwd=want_bwd)
        for w, (uf, ub) in zip(ws, got):
            rf, rb = rc.wino_weights(w, want_bwd=want_bwd)
            assert torch.equal(uf, rf) and (not want_bwd or torch.equal(ub, rb)), tuple(w.shape)
    odd = [(torch.randn((64, 24, 3, 3), generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last), ws[0]]
    for w, (uf, ub) in zip(odd, rc.wino_weights_batch(odd)):
        rf, rb = rc.wino_weights(w)
        assert torch.equal(uf, rf) and torch.equal(ub, rb), tuple(w.shape)
