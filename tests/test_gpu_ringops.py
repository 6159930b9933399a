"""The fused elementwise glue of the pose CNN (delora_amd/csrc/ringops.hip) against the separate torch ops the reference
uses (src/models/resnet_modified.py:97-102,159-177): F.pad(..., 'circular'), tanh/relu, residual add, MaxPool2d.
Forward values and arg-max routing are compared bit-exactly where the arithmetic is the same single operation; tanh
goes through the device's libm on both sides."""
import pytest
import torch
import torch.nn.functional as F

from delora_amd.models.ring_ops import ring_act_pad, ring_act_pool_pad

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _act(v, act):
    return torch.tanh(v) if act == "tanh" else (torch.relu(v) if act == "relu" else v)


def _seed(*parts):
    return sum((i + 1) * ord(c) for i, c in enumerate(repr(parts))) % (2 ** 31)


def _wrap(v):
    return F.pad(v, (1, 1, 0, 0), mode="circular")


@pytest.mark.parametrize("act", ["none", "tanh", "relu"])
@pytest.mark.parametrize("pad", [True, False])
@pytest.mark.parametrize("res_kind", [None, "dense", "padded"])
@pytest.mark.parametrize("shape", [(2, 3, 5, 64), (1, 2, 1, 7), (3, 1, 4, 130)])
def test_ring_act_pad_matches_torch_ops(act, pad, res_kind, shape):
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(_seed(act, pad, res_kind, shape))
    N, C, H, W = shape
    x = torch.randn(shape, generator=gen).to(dev).requires_grad_(True)
    res = None
    if res_kind == "dense":
        res = torch.randn(shape, generator=gen).to(dev).requires_grad_(True)
    elif res_kind == "padded":
        res = torch.randn((N, C, H, W + 2), generator=gen).to(dev).requires_grad_(True)
    out = ring_act_pad(x, act, pad=pad, residual=res)
    xr = x.detach().clone().requires_grad_(True)
    rr = res.detach().clone().requires_grad_(True) if res is not None else None
    v = xr if rr is None else xr + (rr if res_kind == "dense" else rr[..., 1:-1])
    want = _act(v, act)
    want = _wrap(want) if pad else want
    assert out.shape == want.shape
    assert torch.equal(out, want) if act != "tanh" else torch.allclose(out, want, rtol=0, atol=2e-7)
    g = torch.randn(want.shape, generator=gen).to(dev)
    out.backward(g)
    want.backward(g)
    assert torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-6)
    if res is not None:
        assert torch.allclose(res.grad, rr.grad, rtol=1e-6, atol=1e-6)


def _stem_reference(x, act, own_activation=True):
    """The reference's separate ops.  With ``own_activation`` the tanh values come from the library's own elementwise
    kernel (tested above): both sides then see bit-identical activations and the comparison isolates pooling, padding
    and gradient routing from last-bit differences between two builds of the device libm."""
    a = ring_act_pad(x, act, pad=False) if own_activation else _act(x, act)
    return _wrap(F.max_pool2d(_wrap(a), kernel_size=3, stride=(1, 2), padding=(1, 0)))


@pytest.mark.parametrize("act", ["tanh", "relu", "none"])
@pytest.mark.parametrize("shape", [(2, 3, 6, 64), (1, 2, 1, 8), (2, 1, 2, 9), (1, 1, 3, 2), (1, 2, 5, 131), (1, 2, 7, 34), (3, 1, 9, 256)])
@pytest.mark.parametrize("ties", [False, True])
def test_ring_act_pool_pad_matches_torch_ops(act, shape, ties):
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(_seed(act, shape, ties))
    x0 = torch.randn(shape, generator=gen) * 2.0
    if ties:
        x0 = torch.round(x0)                      # many equal values inside a window: the arg-max rule decides the routing
    x = x0.to(dev).requires_grad_(True)
    xr = x0.to(dev).requires_grad_(True)
    out = ring_act_pool_pad(x, act)
    want = _stem_reference(xr, act)
    assert out.shape == want.shape
    assert torch.equal(out, want)                 # max of identical activation values: bit-exact
    assert torch.allclose(out, _stem_reference(xr.detach(), act, own_activation=False), rtol=0, atol=2e-7)
    g = torch.randn(want.shape, generator=gen).to(dev)
    out.backward(g)
    want.backward(g)
    assert torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-6)
    # the routing itself: the same set of input positions receives gradient
    assert torch.equal(x.grad != 0, xr.grad != 0)


def test_ring_act_pool_pad_at_stem_size_and_saturation():
    """Stem-sized tensor; values up to +-12 so that many tanh outputs collide at +-1 (post-activation ties whose
    pre-activation values differ -- the case where pooling before the activation would route differently)."""
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(5)
    x0 = torch.randn((2, 16, 64, 1024), generator=gen) * 6.0
    x = x0.to(dev).requires_grad_(True)
    xr = x0.to(dev).requires_grad_(True)
    out = ring_act_pool_pad(x, "tanh")
    want = _stem_reference(xr, "tanh")
    assert torch.equal(out, want)
    g = torch.randn(want.shape, generator=gen).to(dev)
    out.backward(g)
    want.backward(g)
    assert torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-7)


def test_ring_ops_propagate_nan_like_torch():
    dev = _dev()
    x0 = torch.randn((1, 1, 3, 8))
    x0[0, 0, 1, 3] = float("nan")
    out = ring_act_pool_pad(x0.to(dev), "relu")
    want = _stem_reference(x0.to(dev), "relu", own_activation=False)
    assert torch.equal(torch.isnan(out), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(want, nan=7.0))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_ring_ops_half_storage_match_the_separate_torch_ops(dtype, act):
    """fp16 / bf16 storage (autocast): the fused kernels compute in fp32 and round every stored element once; the separate
    torch ops round after the add as well, so values agree to one rounding step of the storage type, and the stem's
    arg-max (compared on ROUNDED activations, like a max-pool fed by a separate activation kernel) routes identically up
    to ties of the differently rounded sums."""
    dev = _dev()
    eps = 1e-3 if dtype == torch.float16 else 8e-3
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((2, 4, 6, 64), generator=gen).to(dev, dtype).requires_grad_(True)
    res = torch.randn((2, 4, 6, 66), generator=gen).to(dev, dtype).requires_grad_(True)
    y = ring_act_pad(x, act, pad=True, residual=res)
    assert y.dtype == dtype
    gy = torch.randn(y.shape, generator=gen).to(dev, dtype)
    y.backward(gy)
    xr, rr = x.detach().float().requires_grad_(True), res.detach().float().requires_grad_(True)
    y_ref = _wrap(_act(xr + rr[..., 1:-1], act))
    y_ref.backward(gy.float())
    assert float((y.float() - y_ref).abs().max()) <= 2 * eps * max(1.0, float(y_ref.abs().max()))
    assert float((x.grad.float() - xr.grad).abs().max()) <= 4 * eps * float(xr.grad.abs().max())
    assert float((res.grad.float() - rr.grad).abs().max()) <= 4 * eps * float(rr.grad.abs().max())
    # stem
    xs = torch.randn((2, 3, 6, 64), generator=gen).to(dev, dtype).requires_grad_(True)
    ys = ring_act_pool_pad(xs, act)
    assert ys.dtype == dtype
    gs = torch.randn(ys.shape, generator=gen).to(dev, dtype)
    ys.backward(gs)
    xsr = xs.detach().float().requires_grad_(True)
    a = ring_act_pad(xs.detach(), act, pad=False).float()   # the activation as a separate kernel stores it (same libm)
    ref = _wrap(F.max_pool2d(_wrap(a), kernel_size=3, stride=(1, 2), padding=(1, 0)))
    assert torch.equal(ys.float(), ref.detach())
    # gradient: route through the same windows with the fp32 activation derivative of the rounded output
    a2 = _act(xsr, act)
    ref2 = _wrap(F.max_pool2d(_wrap(a2), kernel_size=3, stride=(1, 2), padding=(1, 0)))
    ref2.backward(gs.float())
    close = (xs.grad.float() - xsr.grad).abs() <= 4 * eps * float(xsr.grad.abs().max()) + 1e-6
    assert float(close.float().mean()) >= 0.98              # a few windows tie differently after rounding
