"""Fused [residual add] + activation + wrap-around width padding for the pose CNN (autograd Function over
``dl_ring_act_pad_fwd/bwd`` of libdelora_hip.so).

``ring_act_pad(x, act, pad=True, residual=None)`` returns ``act(x + residual)`` padded by one wrapped column on each
side of W (``pad=False``: unpadded).  ``residual`` may be a *padded* tensor ``[N,C,H,W+2]`` whose interior is the
residual -- the natural form of a block input here, because every activation lives in padded form between
convolutions -- or a dense ``[N,C,H,W]`` tensor (the 1x1 down-sampling branch).  On CUDA tensors the HIP kernels are mandatory (the call raises if the library is
missing); on CPU tensors -- the CPU baseline and the CPU unit tests, where the device itself is the user's choice --
the same function is evaluated with torch ops.
"""
import ctypes

import torch
import torch.nn.functional as F

from .. import _lib

ACT = {"none": 0, "tanh": 1, "relu": 2}
DTYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}      # storage types of the *_t entry points (fp32 arithmetic)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _RingActPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, act, pad):
        lib = _lib.load()
        x = x.contiguous()
        N, C, H, W = x.shape
        rows = N * C * H
        out = torch.empty((N, C, H, W + 2 * pad), dtype=x.dtype, device=x.device)
        res_pitch = res_off = 0
        res_kind = 0                                   # 0 none, 1 dense [N,C,H,W], 2 padded [N,C,H,W+2] (interior used)
        if residual is not None:
            residual = residual.contiguous()
            if residual.shape == (N, C, H, W):
                res_pitch, res_off, res_kind = W, 0, 1
            elif residual.shape == (N, C, H, W + 2):
                res_pitch, res_off, res_kind = W + 2, 1, 2
            else:
                raise ValueError(f"residual shape {tuple(residual.shape)} fits neither [N,C,H,W] nor [N,C,H,W+2]")
        _lib.check(lib.dl_ring_act_pad_fwd_t(_ptr(x), _ptr(residual), res_pitch, res_off, rows, W, pad, act, DTYPE[x.dtype],
                                             _ptr(out), _stream()), "dl_ring_act_pad_fwd")
        ctx.save_for_backward(out)
        ctx.meta = (rows, W, pad, act, res_kind)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        rows, W, pad, act, res_kind = ctx.meta
        grad_out = grad_out.contiguous() if grad_out.dtype == y.dtype else grad_out.to(y.dtype).contiguous()
        N, C, H = y.shape[:3]
        grad_x = torch.empty((N, C, H, W), dtype=y.dtype, device=y.device)
        grad_res = torch.empty((N, C, H, W + 2), dtype=y.dtype, device=y.device) if res_kind == 2 else None
        _lib.check(lib.dl_ring_act_pad_bwd_t(_ptr(grad_out), _ptr(y), rows, W, pad, act, DTYPE[y.dtype], _ptr(grad_x),
                                             _ptr(grad_res), _stream()), "dl_ring_act_pad_bwd")
        if res_kind == 1:
            grad_res = grad_x                          # d(x + r)/dr = d/dx: the same tensor serves both
        return grad_x, grad_res, None, None


class _RingActPoolPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        lib = _lib.load()
        x = x.contiguous()
        N, C, H, W = x.shape
        Wo = (W - 1) // 2 + 1
        out = torch.empty((N, C, H, Wo + 2), dtype=x.dtype, device=x.device)
        win = torch.empty((N, C, H, Wo), dtype=torch.int8, device=x.device)
        _lib.check(lib.dl_ring_act_pool_pad_fwd_t(_ptr(x), N * C, H, W, act, DTYPE[x.dtype], _ptr(out), _ptr(win), _stream()),
                   "dl_ring_act_pool_pad_fwd")
        ctx.save_for_backward(out, win)
        ctx.meta = (N, C, H, W, act)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        y, win = ctx.saved_tensors
        N, C, H, W, act = ctx.meta
        grad_out = grad_out.contiguous() if grad_out.dtype == y.dtype else grad_out.to(y.dtype).contiguous()
        grad_x = torch.empty((N, C, H, W), dtype=y.dtype, device=y.device)
        _lib.check(lib.dl_ring_act_pool_pad_bwd_t(_ptr(grad_out), _ptr(y), _ptr(win), N * C, H, W, act, DTYPE[y.dtype],
                                                  _ptr(grad_x), _stream()), "dl_ring_act_pool_pad_bwd")
        return grad_x, None


def ring_act_pool_pad(x, act="tanh"):
    """Stem of the pose CNN: ``act``, wrap-around padding, ``MaxPool2d(3, stride=(1,2), padding=(1,0))`` and the
    wrap-around padding of the pooled map (reference resnet_modified.py:100-102 + the F.pad of the next convolution) --
    one HIP kernel each way on CUDA fp32 tensors, the separate torch ops otherwise."""
    if x.is_cuda and x.dtype in DTYPE:
        N, C, H, W = x.shape
        per_sample = C * H * (W + 2)
        if N * per_sample >= 2 ** 31 and N > 1:          # the stem kernels index with 32 bits: split the batch
            n = max(1, (2 ** 31 - 1) // per_sample)
            return torch.cat([_RingActPoolPad.apply(x[i:i + n], ACT[act]) for i in range(0, N, n)], dim=0)
        return _RingActPoolPad.apply(x, ACT[act])
    v = torch.tanh(x) if act == "tanh" else (torch.relu(x) if act == "relu" else x)
    v = F.max_pool2d(F.pad(v, (1, 1, 0, 0), mode="circular"), kernel_size=3, stride=(1, 2), padding=(1, 0))
    return F.pad(v, (1, 1, 0, 0), mode="circular")


def ring_act_pad(x, act="none", pad=True, residual=None):
    """``act(x + residual)`` with one wrapped column added on each side of W (``pad=False``: unpadded).  ``residual`` is
    either dense ``[N,C,H,W]`` or a padded ``[N,C,H,W+2]`` tensor whose interior is the residual."""
    if x.is_cuda and x.dtype in DTYPE:
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
        return _RingActPad.apply(x, residual, ACT[act], 1 if pad else 0)
    # CPU device: plain torch ops, same function
    v = x
    if residual is not None:
        v = x + (residual if residual.shape[-1] == x.shape[-1] else residual[..., 1:-1]).to(x.dtype)
    if act == "tanh":
        v = torch.tanh(v)
    elif act == "relu":
        v = torch.relu(v)
    return F.pad(v, (1, 1, 0, 0), mode="circular") if pad else v
