"""Channels-last convolutions of the pose CNN on the fp32 matrix cores: thin torch wrappers over ``dl_conv2d_nhwc_f32`` /
``dl_conv2d_wgrad_nhwc_f32`` (include/delora_hip.h) and the autograd Function of the whole residual trunk.

Reference: ``ResNetModified._forward_impl`` / ``BasicBlock.forward`` (src/models/resnet_modified.py:95-120, :159-177) --
17 x (F.pad circular + Conv2d) + tanh + residual adds under torch autograd.  Here the trunk (layer1..layer4) is ONE
autograd Function on channels-last activations ``[N,H,W,C]``:

  forward   per block   y1 = act(conv(x, w1));  y2 = act(conv(y1, w2) + shortcut)       2 (+1) launches, nothing else
  backward  per block   dW2 = wgrad(y1, g2);  g1 = dgrad(g2, w2) * act'(y1);  dW1 = wgrad(x, g1);
                        g2_prev = (dgrad(g1, w1) + g2) * act'(x)                         4 (+2) launches, nothing else

where g denotes a gradient with respect to a PRE-activation: the activation derivative of the previous block and the
shortcut gradient are folded into the epilogue of the input-gradient convolution, so no elementwise kernel runs between
two convolutions.  The wrap-around of the image width and the zero rows above / below are addressing inside the kernels:
no padded copy of an activation exists.  Weights are the torch parameters themselves in channels_last memory format
(``[K][3][3][C]`` storage); weight gradients are written in the same layout.

The strided layers' input gradients run one pass per stride phase with exactly the taps whose phase matches
(``dgrad_strided``); no library convolution is left in the trunk.
"""
import ctypes
import os

import torch

from .. import _lib

ACT = {"none": 0, "tanh": 1, "relu": 2}
USE_WINOGRAD = True          # stride-1 3x3 layers as fused Winograd F(2x2,3x3) (csrc/wino.hip); False: the direct kernel
EPI_ADD, EPI_ACT, EPI_DACT, EPI_ADD_GRID = 1, 2, 4, 8


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Base-address skew of the activation tensors this module allocates (bytes, cycled per allocation; 0 = off).  Tensors that one
# kernel streams together (x, y, shortcut, saved activation) otherwise start at addresses a large power of two apart.
ALLOC_SKEW = 0
_skew_state = {"i": 0}


def _empty(shape, dtype, device):
    if not ALLOC_SKEW:
        return torch.empty(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= int(d)
    esz = torch.empty((), dtype=dtype).element_size()
    _skew_state["i"] = (_skew_state["i"] + 1) % 8
    off = (_skew_state["i"] * ALLOC_SKEW) // esz
    return torch.empty((n + (8 * ALLOC_SKEW) // esz,), dtype=dtype, device=device)[off:off + n].view(shape)


def weight_storage(w):
    """The parameter ``[K,C,k,k]`` viewed as its channels_last storage ``[K,k,k,C]`` (no copy when it already has that
    memory format -- ``OdometryModel`` converts its convolution weights once)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def grad_for(dw_krsc, w_param):
    """The weight gradient ``dW [K,k,k,C]`` as the gradient tensor of the parameter ``[K,C,k,k]`` (or of its ``(shape, stride)``
    pair): the same memory, with the parameter's own strides.  For a 1x1 filter the permuted view has strides (C,1,C,C) while the
    parameter keeps (C,1,1,1) -- the same layout, but DistributedDataParallel compares strides literally and would copy the
    gradient into its bucket view on every step ("Grad strides do not match bucket view strides")."""
    shape, stride = (tuple(w_param.shape), tuple(w_param.stride())) if torch.is_tensor(w_param) else w_param
    g = dw_krsc.permute(0, 3, 1, 2)
    if tuple(g.stride()) != stride and shape[2] == 1 and shape[3] == 1 and stride[0] == shape[1] and stride[1] == 1:
        g = g.as_strided(shape, stride)
    return g


def out_size(n, stride):
    """Output extent of the reference's padded convolutions (3x3 with one pad pixel per side -- circular on the width, zeros on the
    height -- or 1x1 unpadded; src/models/resnet_modified.py:97-102, :159-177): floor((n - 1) / stride) + 1 = ceil(n / stride)."""
    return -(-int(n) // int(stride))


def conv_nhwc(x, w_krsc, stride=(1, 1), act=0, epilogue=0, add=None, dsrc=None, transposed=False):
    """``y = epilogue(conv(x, w))`` on channels-last fp32 tensors.  x ``[N,H,W,C]``; w ``[K,k,k,C]`` (``transposed``: the
    forward weight ``[C,k,k,K]`` of the layer whose input gradient is computed, x then being the output gradient)."""
    lib = _lib.load()
    N, H, W, C = x.shape
    ks = w_krsc.shape[1]
    K = w_krsc.shape[3] if transposed else w_krsc.shape[0]
    y = _empty((N, out_size(H, stride[0]), out_size(W, stride[1]), K), torch.float32, x.device)
    _lib.check(lib.dl_conv2d_nhwc_f32(_ptr(x), _ptr(w_krsc), _ptr(y), _ptr(add), _ptr(dsrc), N, H, W, C, K, ks, stride[0],
                                      stride[1], int(transposed), int(act), int(epilogue), _stream()), "dl_conv2d_nhwc_f32")
    return y


USE_WINOGRAD_WGRAD = True    # stride-1 3x3 weight gradients with >= 128 input channels in the Winograd domain (csrc/wino.hip)


def wgrad_nhwc(x, g, ks, stride=(1, 1)):
    """``dW [K,k,k,C]`` (channels_last storage of the parameter gradient) from input x ``[N,H,W,C]`` and output gradient g.
    Stride-1 3x3 layers with at least 128 input channels take the Winograd-domain kernel (6-9 % faster there; the 64-channel
    layers and everything strided stay on the direct kernel)."""
    lib = _lib.load()
    N, H, W, C = x.shape
    K = g.shape[3]
    dw = torch.empty((K, ks, ks, C), dtype=torch.float32, device=x.device)
    if USE_WINOGRAD_WGRAD and ks == 3 and tuple(stride) == (1, 1) and C >= 128:
        nbytes = lib.dl_wino_wgrad_workspace_bytes(N, H, W, C, K)
        if nbytes:
            ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
            _lib.check(lib.dl_wino_wgrad3x3_nhwc_f32(_ptr(x), _ptr(g), _ptr(dw), _ptr(ws), N, H, W, C, K, _stream()),
                       "dl_wino_wgrad3x3_nhwc_f32")
            return dw
    ws = torch.empty((lib.dl_conv2d_wgrad_workspace_bytes(N, H, W, C, K, ks, stride[0], stride[1]) // 4,), dtype=torch.float32,
                     device=x.device)
    _lib.check(lib.dl_conv2d_wgrad_nhwc_f32(_ptr(x), _ptr(g), _ptr(dw), _ptr(ws), N, H, W, C, K, ks, stride[0], stride[1],
                                            _stream()), "dl_conv2d_wgrad_nhwc_f32")
    return dw


def _wgrad_layer_table(part):
    """ctypes table of ``dl_wgrad_layer`` for (x, g, ks, stride) items + the freshly allocated fp32 ``dW [K,k,k,C]`` of each."""
    arr = (_lib.WgradLayer * len(part))()
    dws = []
    for j, (x, g, ks, stride) in enumerate(part):
        N, H, W, C = x.shape
        K = g.shape[3]
        dw = torch.empty((K, ks, ks, C), dtype=torch.float32, device=x.device)
        dws.append(dw)
        arr[j].x, arr[j].g, arr[j].dw = x.data_ptr(), g.data_ptr(), dw.data_ptr()
        arr[j].N, arr[j].H, arr[j].W, arr[j].C, arr[j].K = N, H, W, C, K
        arr[j].ksize, arr[j].stride_h, arr[j].stride_w = ks, stride[0], stride[1]
    return arr, dws


def _wgrad_batch_call(items, size_fn, run_fn, what, *extra):
    lib = _lib.load()
    out = []
    for i0 in range(0, len(items), _lib.WGRAD_BATCH):
        part = items[i0:i0 + _lib.WGRAD_BATCH]
        arr, dws = _wgrad_layer_table(part)
        ap = ctypes.cast(arr, ctypes.c_void_p)
        nbytes = getattr(lib, size_fn)(ap, len(part))
        if not nbytes:
            raise _lib.DeloraHipError(what + ": " + (lib.dl_last_error() or b"").decode())
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=part[0][0].device)
        _lib.check(getattr(lib, run_fn)(ap, len(part), _ptr(ws), *extra, _stream()), what)
        out += dws
    return out


def wgrad_batch(items):
    """The fp32 weight gradients of several layers in merged launches (include/delora_hip.h: ``dl_wgrad_layer``): ``items`` = list of
    (x, g, ks, stride); returns ``dW [K,k,k,C]`` per item.  Stride-1 3x3 layers with at least 128 input channels take the Winograd-domain
    kernel (as ``wgrad_nhwc`` decides), the rest the direct one; within each family the layers that share a kernel run in ONE launch
    with far fewer pixel slabs (fp32 partial copies of the gradient) per layer than a launch of their own needs."""
    wino, direct = [], []
    for i, (x, g, ks, stride) in enumerate(items):
        N, H, W, C = x.shape
        K = g.shape[3]
        # (the shape rule of dl_wino_wgrad_workspace_bytes, restated here: one library call per layer and step is host time)
        if (USE_WINOGRAD_WGRAD and ks == 3 and tuple(stride) == (1, 1) and C >= WINO_WGRAD_MIN_C_BATCHED and C % 64 == 0 and K % 64 == 0 and W >= 2
                and N * H * W * max(C, K) < 2 ** 31 and H * W * max(C, K) < 2 ** 30):
            wino.append(i)
        else:
            direct.append(i)
    out = [None] * len(items)
    if wino:
        for i, dw in zip(wino, _wgrad_batch_call([items[i] for i in wino], "dl_wino_wgrad3x3_batch_workspace_bytes", "dl_wino_wgrad3x3_batch_nhwc_f32",
                                                 "dl_wino_wgrad3x3_batch_nhwc_f32")):
            out[i] = dw
    if direct:
        for i, dw in zip(direct, _wgrad_batch_call([items[i] for i in direct], "dl_conv2d_wgrad_batch_workspace_bytes", "dl_conv2d_wgrad_batch_nhwc_f32",
                                                   "dl_conv2d_wgrad_batch_nhwc_f32")):
            out[i] = dw
    return out


def wino_ok(H, W, C, K):
    """Whether the fused Winograd F(2x2,3x3) kernel (csrc/wino.hip) takes a stride-1 3x3 layer of this shape: any image size (groups
    of 64 tiles hang over the edge of images that do not divide; odd sizes end in partial tiles), channel counts that tile."""
    return H >= 1 and W >= 1 and C % 8 == 0 and K % 64 == 0


def wino_weights(w_param, want_fwd=True, want_bwd=True):
    """Winograd-domain weights of a 3x3 layer: (u_fwd, u_bwd: 16 K C floats each in the blocked layout of csrc/wino.hip: wn_u_index) from the parameter ``[K,C,3,3]``."""
    lib = _lib.load()
    w = weight_storage(w_param)
    K, C = w.shape[0], w.shape[3]
    n = lib.dl_wino_weights_floats(K, C)
    uf = torch.empty((n,), dtype=torch.float32, device=w.device) if want_fwd else None
    ub = torch.empty((n,), dtype=torch.float32, device=w.device) if want_bwd else None
    _lib.check(lib.dl_wino_weights_f32(_ptr(w), _ptr(uf), _ptr(ub), K, C, _stream()), "dl_wino_weights_f32")
    return uf, ub


def wino_weights_batch(w_params, want_bwd=True):
    """Winograd-domain weights of several 3x3 layers in ONE launch (``dl_wino_weights_batch_f32``): a list of (u_fwd, u_bwd) in the
    order of ``w_params``; the outputs of all layers are views of two allocations."""
    lib = _lib.load()
    ws = [weight_storage(w) for w in w_params]
    out = []
    if not ws:
        return out
    sizes = [int(lib.dl_wino_weights_floats(w.shape[0], w.shape[3])) for w in ws]
    dev = ws[0].device
    uf_all = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
    ub_all = torch.empty((sum(sizes),), dtype=torch.float32, device=dev) if want_bwd else None
    off = 0
    for n in sizes:
        out.append((uf_all[off:off + n], ub_all[off:off + n] if want_bwd else None))
        off += n
    for i0 in range(0, len(ws), _lib.WINO_BATCH):
        part = list(range(i0, min(i0 + _lib.WINO_BATCH, len(ws))))
        arr = (_lib.WinoLayer * len(part))()
        for j, i in enumerate(part):
            arr[j].w = ws[i].data_ptr()
            arr[j].u_fwd = out[i][0].data_ptr()
            arr[j].u_bwd = out[i][1].data_ptr() if want_bwd else None
            arr[j].K, arr[j].C = ws[i].shape[0], ws[i].shape[3]
        _lib.check(lib.dl_wino_weights_batch_f32(ctypes.cast(arr, ctypes.c_void_p), len(part), _stream()), "dl_wino_weights_batch_f32")
    return out


def wino_conv(x, u, K, act=0, epilogue=0, add=None, dsrc=None):
    """Stride-1 3x3 convolution of x ``[N,H,W,C]`` with Winograd-domain weights ``u`` (u_fwd: forward; u_bwd with the
    output gradient as x: input gradient) -> ``[N,H,W,K]``, epilogue as conv_nhwc."""
    lib = _lib.load()
    N, H, W, C = x.shape
    y = _empty((N, H, W, K), torch.float32, x.device)
    # a launch with few tile groups (batch 1) splits its input channels over otherwise idle CUs and needs scratch for the partial sums
    key = (N, H, W, C, K)
    nbytes = _WINO_WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _WINO_WS_BYTES[key] = int(lib.dl_wino_conv3x3_workspace_bytes(N, H, W, C, K))
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device) if nbytes else None
    _lib.check(lib.dl_wino_conv3x3_nhwc_f32(_ptr(x), _ptr(u), _ptr(y), _ptr(add), _ptr(dsrc), N, H, W, C, K, int(act),
                                            int(epilogue), _ptr(ws), _stream()), "dl_wino_conv3x3_nhwc_f32")
    return y


_WINO_WS_BYTES = {}


def supported(x_shape, blocks):
    """Whether the HIP trunk can run these shapes: channel counts multiples of 64.  The image size is free: tiles hang over the edge
    of feature maps that do not divide, odd widths / heights included -- the reference's shipped 64x720 image (feature maps 180, 90,
    45 and 23 pixels wide, config/config_datasets.yaml:21) and 64x512 (layer4: 32x16, :47) run here like BASELINE's 64x2048."""
    return _supported(x_shape, blocks, 1 << 31)


def _supported(x_shape, blocks, max_elements):
    """``max_elements``: the C side indexes a tensor with 32-bit element (fp32) / byte (half) offsets and rejects N*H*W*max(C,K) at or
    above 2^31 (conv.hip, wino.hip) resp. 2^30 (convh.hip, wgradh.hip); the Winograd kernels additionally keep a sample's H*W*C below
    2^30.  The same bounds here let cnn_impl 'auto' fall back to the module path instead of raising from the C side."""
    N, H, W, C = x_shape
    if C % 64 or H < 1 or W < 1:
        return False
    for (cin, cout, stride, _) in blocks:
        if cin % 64 or cout % 64 or stride[0] not in (1, 2) or stride[1] not in (1, 2) or (stride[0] == 2 and stride[1] == 1):
            return False
        wide = max(cin, cout)
        if N * H * W * wide >= max_elements or H * W * wide >= (1 << 30):
            return False
        H, W = out_size(H, stride[0]), out_size(W, stride[1])
    return True


def _seam_workspace(N, H, W, C, ks, stride, dense, device):
    """fp32 ``[N,H,2,C]`` scratch for the two seam terms of a stride-2-in-width 3x3 layer on an ODD image width (the stride phases do
    not close under the wrap-around there; csrc/conv.hip: k_dgrad_oddw_seam), else None."""
    if ks == 3 and stride[1] == 2 and W % 2 == 1 and W >= 3 and not dense:
        return torch.empty((N, H, 2, C), dtype=torch.float32, device=device)
    return None


def dgrad_strided(g, w_krsc, stride, in_hw, act=0, epilogue=0, add_grid=None, dsrc=None, dense=False):
    """Input gradient of a strided layer whose INPUT image is ``in_hw`` = (H, W): g ``[N,Ho,Wo,K]`` (Ho = ceil(H / sh), Wo = ceil(W /
    sw)) with the layer's forward weight ``[K,k,k,C]`` -> ``[N,H,W,C]`` (``dense``: the 1x1 layers' gradient kept on the grid
    ``[N,Ho,Wo,C]``)."""
    lib = _lib.load()
    N, Ho, Wo, K = g.shape
    H, W = int(in_hw[0]), int(in_hw[1])
    assert Ho == out_size(H, stride[0]) and Wo == out_size(W, stride[1]), (g.shape, in_hw, stride)
    ks, C = w_krsc.shape[1], w_krsc.shape[3]
    shape = (N, Ho, Wo, C) if dense else (N, H, W, C)
    dx = _empty(shape, torch.float32, g.device)
    seam = _seam_workspace(N, H, W, C, ks, stride, dense, g.device)
    _lib.check(lib.dl_conv2d_dgrad_strided_nhwc_f32(_ptr(g), _ptr(w_krsc), _ptr(dx), _ptr(add_grid), _ptr(dsrc), N, H, W, K, C, ks,
                                                    stride[0], stride[1], int(dense), int(act), int(epilogue), _ptr(seam), _stream()),
               "dl_conv2d_dgrad_strided_nhwc_f32")
    return dx


def pool_fwd(a):
    """Stem max-pooling (3x3, stride (1,2), wrap-around width) of a ``[N,H,W,C]`` -> (y ``[N,H,W/2,C]``, win int8)."""
    lib = _lib.load()
    N, H, W, C = a.shape
    y = torch.empty((N, H, W // 2, C), dtype=torch.float32, device=a.device)
    win = torch.empty((N, H, W // 2, C), dtype=torch.int8, device=a.device)
    _lib.check(lib.dl_pool3x3s12_nhwc_fwd(_ptr(a), N, H, W, C, _ptr(y), _ptr(win), _stream()), "dl_pool3x3s12_nhwc_fwd")
    return y, win


def pool_bwd(g, a, win, act):
    """Gradient of ``pool_fwd(act(.))`` with respect to the PRE-activation conv1 output: ``[N,H,W,C]``."""
    lib = _lib.load()
    N, H, W, C = a.shape
    out = torch.empty_like(a)
    _lib.check(lib.dl_pool3x3s12_nhwc_bwd(_ptr(g), _ptr(a), _ptr(win), N, H, W, C, int(act), _ptr(out), _stream()),
               "dl_pool3x3s12_nhwc_bwd")
    return out


class MeanHW(torch.autograd.Function):
    """Global average pooling of a channels-last map ``[N,H,W,C]`` -> ``[N,C]`` (avgpool + flatten of the reference) as one
    deterministic kernel; the backward is the broadcast of ``g / (H*W)``."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        N, H, W, C = x.shape
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        _lib.check(lib.dl_mean_hw_nhwc_f32(_ptr(x), N, H * W, C, _ptr(y), _stream()), "dl_mean_hw_nhwc_f32")
        ctx.shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = ctx.shape
        return (g * (1.0 / (H * W))).view(N, 1, 1, C).expand(N, H, W, C)


def stem_supported(x_shape, out_channels):
    """Whether RingStem takes an input ``[N,C,H,W]``: 8 input channels per chunk, 64-channel output tiles, a width that halves twice."""
    N, C, H, W = x_shape
    return C % 8 == 0 and C % 16 != 0 and out_channels % 64 == 0 and W % 4 == 0 and W >= 4 and H >= 1


class RingStem(torch.autograd.Function):
    """conv1 (3x3, stride (1,2), wrap-around width) + activation + the 3x3 / stride (1,2) max-pooling of the stem on
    channels-last tensors (reference resnet_modified.py:97-102): ``[N,8,H,W]`` planar input -> ``[N,H,W/4,64]``.  Forward:
    one transposing copy of the 8-channel input, the direct MFMA convolution with the activation in its epilogue, the
    pooling kernel.  Backward: pooling backward + activation derivative + conv1's weight gradient in ONE kernel
    (``dl_stem_wgrad_f32``); only when the gradient with respect to the INPUT image is wanted (never in training: the range image
    is data) the three steps run separately with the library's convolution backward.  The 134 MB pre-pooling map is kept for
    the backward (the pooled output is not: the trunk saves it)."""

    @staticmethod
    def forward(ctx, x, w1, act):
        x8 = x.permute(0, 2, 3, 1).contiguous()
        a = conv_nhwc(x8, weight_storage(w1), stride=(1, 2), act=act, epilogue=EPI_ACT if act else 0)
        y, win = pool_fwd(a)
        ctx.save_for_backward(x8, a, win, w1)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        x8, a, win, w1 = ctx.saved_tensors
        want_x = ctx.needs_input_grad[0]
        lib = _lib.load()
        N, H, W, _ = x8.shape
        nbytes = 0 if want_x or a.shape[3] != 64 or x8.shape[3] != 8 else lib.dl_stem_wgrad_workspace_bytes(N, H, W)
        if nbytes:
            # the usual case (the range image is data): pooling backward + act' + conv1's weight gradient in ONE kernel, the
            # gradient with respect to conv1's pre-activation (134 MB at batch 8) is never written (csrc/stem.hip: k_stem_wgrad)
            ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x8.device)
            dw = torch.empty((64, 8, 3, 3), dtype=torch.float32, device=x8.device)
            _lib.check(lib.dl_stem_wgrad_f32(_ptr(g.contiguous()), _ptr(a), _ptr(win), _ptr(x8), N, H, W, int(ctx.act), _ptr(ws), _ptr(dw),
                                             _stream()), "dl_stem_wgrad_f32")
            return None, dw, None
        gc = pool_bwd(g.contiguous(), a, win, ctx.act)
        xp = torch.cat((x8[:, :, -1:], x8, x8[:, :, :1]), dim=2).permute(0, 3, 1, 2)       # wrapped, channels_last strides
        gx, dw, _ = torch.ops.aten.convolution_backward(gc.permute(0, 3, 1, 2), xp, w1, None, (1, 2), (1, 0), (1, 1), False,
                                                        (0, 0), 1, (want_x, True, False))
        if want_x:                                     # fold the two wrap columns back
            gxp = gx
            gx = gxp[..., 1:-1].clone()
            gx[..., -1] += gxp[..., 0]
            gx[..., 0] += gxp[..., -1]
        return (gx if want_x else None), dw, None


# How the trunk is cut into autograd Functions.  Each Function hands its weight gradients to autograd when its backward has run, so
# the cuts decide what DistributedDataParallel can overlap with the rest of the backward; each Function also costs the host
# ~0.1 ms per step (apply + saved tensors + backward dispatch), and the eager step is only just GPU-bound:
#   "mono"   one Function for layer1..layer4 (single process: nothing to overlap)
#   "layer"  [layer1 + layer2] [layer3] [layer4]: layer4 (71 % of the gradient bytes) and layer3 (18 %) are in flight while the
#            rest of the backward runs -- what deploy/trainer.py selects under DDP
#   "block"  one Function per residual block
TRUNK_SEGMENTS = "mono"
# Fewest input channels of a stride-1 3x3 layer whose weight gradient takes the Winograd-domain kernel inside a merged launch.  A single
# launch keeps 128 (wgrad_nhwc): a 64-channel layer has ONE output tile, i.e. 256 slab partials of 262 KB each -- the direct kernel won.
# Merged with the other layers of the segment it needs 64 slabs, and 2.25x fewer multiplications decide (A/B in DESIGN.md 4.6).
WINO_WGRAD_MIN_C_BATCHED = int(os.environ.get("DELORA_WINO_WGRAD_MIN_C", "64"))
# Weight gradients of a segment in merged launches at its end (wgrad_batch_h) instead of one launch per layer inside the chain
WGRAD_BATCHED = os.environ.get("DELORA_WGRAD_BATCHED", "1") != "0"      # (0: one launch per layer, for A/B measurements)
# Deferring a segment's weight gradients keeps every inter-layer GRADIENT map of the segment alive until they are computed (the
# activations are saved tensors and alive anyway).  At B=8, 64x2048 that is ~1 GB for the 19 layers of a mono trunk; above this many
# bytes the collected layers are flushed early (one more merged launch), so that peak memory stays bounded for large batches / tall
# images (advisor, round 4).
WGRAD_PENDING_MAX_BYTES = int(os.environ.get("DELORA_WGRAD_PENDING_MAX_BYTES", str(2 << 30)))


class _PendingWgrads:
    """(gradient slot, (x, g, filter size, stride)) items of a segment's backward, flushed into ``grads`` through ``batch_fn`` (merged
    launches) or ``single_fn`` at the end of the segment -- or earlier, once the gradient maps they keep alive pass
    ``WGRAD_PENDING_MAX_BYTES``."""

    def __init__(self, grads, meta, batch_fn, single_fn):
        self.grads, self.meta, self.batch_fn, self.single_fn = grads, meta, batch_fn, single_fn
        self.items, self.bytes, self.flushes = [], 0, 0

    def append(self, item):
        self.items.append(item)
        g = item[1][1]
        self.bytes += g.numel() * g.element_size()
        if self.bytes > WGRAD_PENDING_MAX_BYTES:
            self.flush()

    def flush(self):
        if not self.items:
            return
        if WGRAD_BATCHED:
            for (gi, _), dw in zip(self.items, self.batch_fn([it for _, it in self.items])):
                self.grads[gi] = grad_for(dw, self.meta[gi])
        else:
            for gi, (xx, gg, ks, st) in self.items:
                self.grads[gi] = grad_for(self.single_fn(xx, gg, ks, stride=st), self.meta[gi])
        self.items, self.bytes = [], 0
        self.flushes += 1
# Order in which the segment Functions finished their backward passes in this process (tests: DDP overlap) -- appended to when a list
BACKWARD_TRACE = None


def _segments(blocks, mode):
    """Index ranges [(b0, b1), ...] of the blocks each autograd Function covers."""
    nb = len(blocks)
    if mode == "block":
        return [(b, b + 1) for b in range(nb)]
    if mode == "layer":
        cuts = [b for b in range(1, nb) if blocks[b][3] and blocks[b][0] >= 128]      # first block of layer3 and of layer4
        edges = [0] + cuts + [nb]
        return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
    return [(0, nb)]


class RingSegment(torch.autograd.Function):
    """A run of residual blocks of the pose CNN (reference BasicBlock.forward, src/models/resnet_modified.py:159-177) on channels-
    last fp32 activations: ``forward(x, act, blocks, first, last, *weights)`` with blocks = tuple of (cin, cout, stride,
    has_downsample) and the weights in block order (conv1, conv2[, downsample]); returns the last block's activated output.

    Private gradient convention BETWEEN two segments (the tensors never leave ``ring_trunk``): the gradient a segment receives for
    its output is already multiplied by ``act'(output)`` -- it is the gradient with respect to the PRE-activation -- because
    the consumer folds that factor into the epilogue of its input-gradient convolution (``EPI_DACT``): no elementwise kernel
    runs between two convolutions, across segment boundaries either.  Only the ``last`` segment receives a true ``dL/dy`` (from
    the pooling) and applies ``act'`` itself; only the ``first`` one returns a true ``dL/dx`` (x = the pooled stem output,
    whose activation derivative belongs to the stem)."""

    @staticmethod
    def forward(ctx, x0, act, blocks, first, last, *weights):
        """Stride-1 3x3 layers run as fused Winograd F(2x2,3x3) when ``USE_WINOGRAD`` and the shape tiles (their
        Winograd-domain weights for the backward pass are produced by the same launch and kept for it); the strided and 1x1
        layers run the direct kernel."""
        saved, x, wi = [x0], x0, 0
        ubwd = []
        need_bwd = any(ctx.needs_input_grad)
        # which layers of this segment run as Winograd (shape walk), and their transformed weights in ONE launch
        plan, (N, H, W) = [], x0.shape[:3]
        for (cin, cout, stride, has_ds) in blocks:
            use1 = USE_WINOGRAD and stride == (1, 1) and wino_ok(H, W, cin, cout)
            H, W = out_size(H, stride[0]), out_size(W, stride[1])
            use2 = USE_WINOGRAD and wino_ok(H, W, cout, cout)
            plan.append((use1, use2))
        wlist, wj = [], 0
        for (cin, cout, stride, has_ds), (use1, use2) in zip(blocks, plan):
            if use1:
                wlist.append(weights[wj])
            if use2:
                wlist.append(weights[wj + 1])
            wj += 3 if has_ds else 2
        us = iter(wino_weights_batch(wlist, want_bwd=need_bwd))
        for (cin, cout, stride, has_ds), (use1, use2) in zip(blocks, plan):
            w1p, w2p = weights[wi], weights[wi + 1]
            wd = weight_storage(weights[wi + 2]) if has_ds else None
            wi += 3 if has_ds else 2
            if use1:
                uf, ub1 = next(us)
                y1 = wino_conv(x, uf, cout, act=act, epilogue=EPI_ACT)
            else:
                ub1 = None
                y1 = conv_nhwc(x, weight_storage(w1p), stride=stride, act=act, epilogue=EPI_ACT)
            shortcut = conv_nhwc(x, wd, stride=stride) if has_ds else x
            if use2:
                uf, ub2 = next(us)
                y2 = wino_conv(y1, uf, cout, act=act, epilogue=EPI_ADD | EPI_ACT, add=shortcut)
            else:
                ub2 = None
                y2 = conv_nhwc(y1, weight_storage(w2p), act=act, epilogue=EPI_ADD | EPI_ACT, add=shortcut)
            ubwd += [ub1, ub2]
            saved += [y1, y2]
            x = y2
        ctx.act, ctx.blocks, ctx.first, ctx.last = act, blocks, first, last
        # the Winograd-domain backward weights travel with the saved tensors (released with the graph, covered by autograd's
        # in-place version check like the raw weights); layers on the direct kernel have none
        ctx.ubwd_mask = tuple(u is not None for u in ubwd)
        ctx.save_for_backward(*saved, *weights, *[u for u in ubwd if u is not None])
        return x

    @staticmethod
    def backward(ctx, dy):
        act, blocks = ctx.act, ctx.blocks
        nb = len(blocks)
        saved = ctx.saved_tensors
        nu = sum(ctx.ubwd_mask)
        acts, weights = saved[:1 + 2 * nb], saved[1 + 2 * nb:len(saved) - nu]
        u_it = iter(saved[len(saved) - nu:])
        ubwd = [next(u_it) if has else None for has in ctx.ubwd_mask]
        grads = [None] * len(weights)
        g2 = dy.contiguous()
        if ctx.last:                                        # a true dL/dy: the activation derivative is applied here
            y_last = acts[-1]
            if act == ACT["tanh"]:
                g2 = g2 * (1.0 - y_last * y_last)
            elif act == ACT["relu"]:
                g2 = g2 * (y_last > 0).to(g2.dtype)
        wi = len(weights)
        # The weight gradients do not feed the chain of input gradients: they are collected (x, g stay alive) and computed together at
        # the end of the segment -- merged launches need far fewer slab partials than one launch per layer (wgrad_batch)
        pending = _PendingWgrads(grads, weights, wgrad_batch, wgrad_nhwc)
        for b in range(nb - 1, -1, -1):
            cin, cout, stride, has_ds = blocks[b]
            wi -= 3 if has_ds else 2
            w1p, w2p = weights[wi], weights[wi + 1]
            x, y1 = acts[2 * b], acts[2 * b + 1]
            first = ctx.first and b == 0                    # x0 is the pooled stem output: its act' belongs to the stem
            pending.append((wi + 1, (y1, g2, 3, (1, 1))))
            ub1, ub2 = ubwd[2 * b], ubwd[2 * b + 1]
            if ub2 is not None:
                g1 = wino_conv(g2, ub2, cout, act=act, epilogue=EPI_DACT, dsrc=y1)
            else:
                g1 = conv_nhwc(g2, weight_storage(w2p), act=act, epilogue=EPI_DACT, dsrc=y1, transposed=True)
            pending.append((wi, (x, g1, 3, stride)))
            if not has_ds:
                epi = EPI_ADD if first else (EPI_ADD | EPI_DACT)
                if ub1 is not None:
                    g2 = wino_conv(g1, ub1, cin, act=act, epilogue=epi, add=g2, dsrc=None if first else x)
                else:
                    g2 = conv_nhwc(g1, weight_storage(w1p), act=act, epilogue=epi, add=g2, dsrc=None if first else x, transposed=True)
            else:
                wdp = weights[wi + 2]
                pending.append((wi + 2, (x, g2, 1, stride)))
                # down-sampling branch on the grid, then one pass per stride phase of the 3x3 layer with it and act'(x) fused
                dxb = dgrad_strided(g2, weight_storage(wdp), stride, x.shape[1:3], dense=True)
                epi = EPI_ADD_GRID if first else (EPI_ADD_GRID | EPI_DACT)
                g2 = dgrad_strided(g1, weight_storage(w1p), stride, x.shape[1:3], act=act, epilogue=epi, add_grid=dxb, dsrc=None if first else x)
        pending.flush()
        if BACKWARD_TRACE is not None:
            BACKWARD_TRACE.append(("segment", blocks[0][0], blocks[-1][1], nb))
        return (g2, None, None, None, None, *grads)


def _run_segments(fn, x, act, blocks, weights, mode, last_applies_act):
    segs = _segments(blocks, mode or TRUNK_SEGMENTS)
    wi = 0
    for i, (b0, b1) in enumerate(segs):
        nw = sum(3 if blocks[b][3] else 2 for b in range(b0, b1))
        x = fn.apply(x, act, tuple(blocks[b0:b1]), i == 0, last_applies_act and i == len(segs) - 1, *weights[wi:wi + nw])
        wi += nw
    return x


def ring_trunk(x0, act, blocks, weights, segments=None):
    """layer1..layer4 of the pose CNN: x0 ``[N,H,W,C0]`` channels-last fp32, already activated (the pooled stem output); blocks =
    tuple of (cin, cout, stride, has_downsample); weights in block order (conv1, conv2[, downsample]).  Returns the last
    feature map ``[N,H',W',C']``.  ``segments``: how the trunk is cut into autograd Functions (default ``TRUNK_SEGMENTS``)."""
    return _run_segments(RingSegment, x0, act, blocks, list(weights), segments, True)


class RingTrunk:
    """Round-2 interface kept for callers: ``RingTrunk.apply(x0, act, blocks, *weights)`` = ``ring_trunk``."""

    @staticmethod
    def apply(x0, act, blocks, *weights):
        return ring_trunk(x0, act, blocks, list(weights))


# ---------------------------------------------------------------------------------------------------------------------
# Half precision (autocast): the same trunk on v_mfma_f32_32x32x16_bf16 / _f16 (csrc/convh.hip, wgradh.hip).  Activations and
# gradients live in bf16 / fp16 channels-last, accumulation and elementwise tails are fp32, the parameters stay fp32 (their
# half-precision copies in the two layouts the kernels read are rebuilt per call), weight gradients come back in fp32.

DTYPE_CODE = {torch.float16: 1, torch.bfloat16: 2}          # DL_DTYPE_F16 / DL_DTYPE_BF16


def weights_h(w_param, dtype, want_fwd=True, want_bwd=True):
    """Half-precision copies of a convolution parameter ``[K,C,k,k]``: (w_fwd ``[k*k,K,C]``, w_bwd ``[k*k,C,K]``)."""
    lib = _lib.load()
    w = weight_storage(w_param)
    K, ks, _, C = w.shape
    wf = torch.empty((ks * ks, K, C), dtype=dtype, device=w.device) if want_fwd else None
    wb = torch.empty((ks * ks, C, K), dtype=dtype, device=w.device) if want_bwd else None
    _lib.check(lib.dl_conv_weights_h(_ptr(w), _ptr(wf), _ptr(wb), K, ks * ks, C, DTYPE_CODE[dtype], _stream()), "dl_conv_weights_h")
    return wf, wb


def weights_batch_h(w_params, dtype, want_bwd=True):
    """``weights_h`` for a list of parameters in ONE launch (``dl_conv_weights_batch_h``): list of (w_fwd, w_bwd)."""
    lib = _lib.load()
    ws = [weight_storage(w) for w in w_params]
    out = []
    for w in ws:
        K, ks, _, C = w.shape
        out.append((torch.empty((ks * ks, K, C), dtype=dtype, device=w.device),
                    torch.empty((ks * ks, C, K), dtype=dtype, device=w.device) if want_bwd else None))
    for i0 in range(0, len(ws), _lib.CONVH_BATCH):
        part = list(range(i0, min(i0 + _lib.CONVH_BATCH, len(ws))))
        arr = (_lib.ConvHLayer * len(part))()
        for j, i in enumerate(part):
            K, ks, _, C = ws[i].shape
            arr[j].w = ws[i].data_ptr()
            arr[j].w_fwd = out[i][0].data_ptr()
            arr[j].w_bwd = out[i][1].data_ptr() if want_bwd else None
            arr[j].K, arr[j].taps, arr[j].C = K, ks * ks, C
        _lib.check(lib.dl_conv_weights_batch_h(ctypes.cast(arr, ctypes.c_void_p), len(part), DTYPE_CODE[dtype], _stream()), "dl_conv_weights_batch_h")
    return out


def conv_nhwc_h(x, w_prepared, ks, stride=(1, 1), act=0, epilogue=0, add=None, dsrc=None, transposed=False):
    """``y = epilogue(conv(x, w))`` on half-precision channels-last tensors; ``w_prepared`` = ``w_fwd`` of the layer, or (``transposed``)
    ``w_bwd`` of the layer whose input gradient is wanted, x then being the output gradient."""
    lib = _lib.load()
    N, H, W, C = x.shape
    K = w_prepared.shape[1]
    y = torch.empty((N, out_size(H, stride[0]), out_size(W, stride[1]), K), dtype=x.dtype, device=x.device)
    _lib.check(lib.dl_conv2d_nhwc_h(_ptr(x), _ptr(w_prepared), _ptr(y), _ptr(add), _ptr(dsrc), N, H, W, C, K, ks, stride[0], stride[1],
                                    int(transposed), DTYPE_CODE[x.dtype], int(act), int(epilogue), _stream()), "dl_conv2d_nhwc_h")
    return y


def dgrad_strided_h(g, w_bwd, ks, stride, in_hw, act=0, epilogue=0, add_grid=None, dsrc=None, dense=False):
    """Input gradient of a strided layer whose INPUT image is ``in_hw`` = (H, W), from its half-precision output gradient
    ``[N,Ho,Wo,K]`` (Ho = ceil(H / sh), Wo = ceil(W / sw)) and ``w_bwd [k*k,C,K]`` -> ``[N,H,W,C]`` (``dense``: ``[N,Ho,Wo,C]``)."""
    lib = _lib.load()
    N, Ho, Wo, K = g.shape
    H, W = int(in_hw[0]), int(in_hw[1])
    assert Ho == out_size(H, stride[0]) and Wo == out_size(W, stride[1]), (g.shape, in_hw, stride)
    C = w_bwd.shape[1]
    shape = (N, Ho, Wo, C) if dense else (N, H, W, C)
    dx = torch.empty(shape, dtype=g.dtype, device=g.device)
    seam = _seam_workspace(N, H, W, C, ks, stride, dense, g.device)
    _lib.check(lib.dl_conv2d_dgrad_strided_nhwc_h(_ptr(g), _ptr(w_bwd), _ptr(dx), _ptr(add_grid), _ptr(dsrc), N, H, W, K, C, ks, stride[0],
                                                  stride[1], int(dense), DTYPE_CODE[g.dtype], int(act), int(epilogue), _ptr(seam), _stream()),
               "dl_conv2d_dgrad_strided_nhwc_h")
    return dx


def wgrad_nhwc_h(x, g, ks, stride=(1, 1)):
    """fp32 ``dW [K,k,k,C]`` (channels_last storage of the parameter gradient) from half-precision input x and output gradient g."""
    lib = _lib.load()
    N, H, W, C = x.shape
    K = g.shape[3]
    nbytes = lib.dl_conv2d_wgrad_h_workspace_bytes(N, H, W, C, K, ks, stride[0], stride[1])
    if not nbytes:
        raise _lib.DeloraHipError(f"dl_conv2d_wgrad_nhwc_h does not support x {tuple(x.shape)} -> K={K}, kernel {ks}, stride {stride}")
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    dw = torch.empty((K, ks, ks, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.dl_conv2d_wgrad_nhwc_h(_ptr(x), _ptr(g), _ptr(dw), _ptr(ws), N, H, W, C, K, ks, stride[0], stride[1], DTYPE_CODE[x.dtype],
                                          _stream()), "dl_conv2d_wgrad_nhwc_h")
    return dw


def wgrad_batch_h(items):
    """The weight gradients of several layers in one call (``dl_conv2d_wgrad_batch_nhwc_h``): ``items`` = list of (x, g, ks, stride) with
    half-precision channels-last x / g; returns the fp32 ``dW [K,k,k,C]`` of every item.  Layers that share a kernel run in ONE launch
    and need far fewer pixel slabs each than a launch of their own would (include/delora_hip.h); the caller keeps x and g alive until
    the call -- ``RingSegmentH.backward`` defers the gradients of a whole segment to its end."""
    return _wgrad_batch_call(items, "dl_conv2d_wgrad_batch_h_workspace_bytes", "dl_conv2d_wgrad_batch_nhwc_h", "dl_conv2d_wgrad_batch_nhwc_h",
                             DTYPE_CODE[items[0][0].dtype])


def supported_h(x_shape, blocks):
    """Whether the half-precision HIP trunk can run these shapes: as ``supported`` (channel counts multiples of 64, any image size:
    tiles hang over the edges of feature maps that do not divide)."""
    return _supported(x_shape, blocks, 1 << 30)


class CastToHalf(torch.autograd.Function):
    """fp32 -> bf16 / fp16 copy of the pooled stem output (one launch); the backward converts the gradient back."""

    @staticmethod
    def forward(ctx, x, dtype):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        _lib.check(lib.dl_cast_f32_to_h(_ptr(x), _ptr(y), x.numel(), DTYPE_CODE[dtype], _stream()), "dl_cast_f32_to_h")
        return y

    @staticmethod
    def backward(ctx, g):
        return g.float(), None


class RingSegmentH(torch.autograd.Function):
    """``RingSegment`` in half precision: x / outputs / inter-segment gradients in bf16 or fp16, fp32 parameters, fp32 weight
    gradients (same private gradient convention, same launch structure; one weight-conversion launch per segment in addition)."""

    @staticmethod
    def forward(ctx, x0, act, blocks, first, last, *weights):
        dtype = x0.dtype
        need_bwd = any(ctx.needs_input_grad)
        saved, wbs, x, wi = [x0], [], x0, 0
        prepared = weights_batch_h(list(weights), dtype, want_bwd=need_bwd)        # every convolution of the segment: one launch
        for (cin, cout, stride, has_ds) in blocks:
            (w1f, w1b), (w2f, w2b) = prepared[wi], prepared[wi + 1]
            wdf, wdb = prepared[wi + 2] if has_ds else (None, None)
            wi += 3 if has_ds else 2
            y1 = conv_nhwc_h(x, w1f, 3, stride=stride, act=act, epilogue=EPI_ACT)
            shortcut = conv_nhwc_h(x, wdf, 1, stride=stride) if has_ds else x
            y2 = conv_nhwc_h(y1, w2f, 3, act=act, epilogue=EPI_ADD | EPI_ACT, add=shortcut)
            wbs += [w1b, w2b] + ([wdb] if has_ds else [])
            saved += [y1, y2]
            x = y2
        ctx.act, ctx.blocks, ctx.first, ctx.last = act, blocks, first, last
        ctx.n_w = len(weights)
        ctx.w_meta = [(tuple(w.shape), tuple(w.stride())) for w in weights]
        ctx.save_for_backward(*saved, *(wbs if need_bwd else []))
        return x

    @staticmethod
    def backward(ctx, dy):
        act, blocks = ctx.act, ctx.blocks
        nb = len(blocks)
        saved = ctx.saved_tensors
        acts, wbs = saved[:1 + 2 * nb], saved[1 + 2 * nb:]
        grads = [None] * ctx.n_w
        g2 = dy.contiguous()
        if ctx.last:                                    # never taken by ring_trunk_h (the pooling Function hands over the pre-
            y_last = acts[-1]                           # activation gradient); kept for stand-alone use of a segment
            if act == ACT["tanh"]:
                g2 = (g2.float() * (1.0 - y_last.float() ** 2)).to(g2.dtype)
            elif act == ACT["relu"]:
                g2 = g2 * (y_last > 0).to(g2.dtype)
        wi = ctx.n_w
        # The weight gradients do not feed the chain of input gradients: they are collected (x, g kept alive) and computed together
        # at the end of the segment -- merged launches need far fewer slab partials than one launch per layer (wgrad_batch_h)
        pending = _PendingWgrads(grads, ctx.w_meta, wgrad_batch_h, wgrad_nhwc_h)
        for b in range(nb - 1, -1, -1):
            cin, cout, stride, has_ds = blocks[b]
            wi -= 3 if has_ds else 2
            w1b, w2b = wbs[wi], wbs[wi + 1]
            x, y1 = acts[2 * b], acts[2 * b + 1]
            first = ctx.first and b == 0
            pending.append((wi + 1, (y1, g2, 3, (1, 1))))
            g1 = conv_nhwc_h(g2, w2b, 3, act=act, epilogue=EPI_DACT, dsrc=y1, transposed=True)
            pending.append((wi, (x, g1, 3, stride)))
            if not has_ds:
                epi = EPI_ADD if first else (EPI_ADD | EPI_DACT)
                g2 = conv_nhwc_h(g1, w1b, 3, act=act, epilogue=epi, add=g2, dsrc=None if first else x, transposed=True)
            else:
                pending.append((wi + 2, (x, g2, 1, stride)))
                dxb = dgrad_strided_h(g2, wbs[wi + 2], 1, stride, x.shape[1:3], dense=True)
                epi = EPI_ADD_GRID if first else (EPI_ADD_GRID | EPI_DACT)
                g2 = dgrad_strided_h(g1, w1b, 3, stride, x.shape[1:3], act=act, epilogue=epi, add_grid=dxb, dsrc=None if first else x)
        pending.flush()
        if BACKWARD_TRACE is not None:
            BACKWARD_TRACE.append(("segment", blocks[0][0], blocks[-1][1], nb))
        return (g2, None, None, None, None, *grads)


class MeanHWActH(torch.autograd.Function):
    """Global average pooling of the last half-precision feature map -> fp32 ``[N,C]``; its backward is fused with the activation
    derivative of the block that produced the map and returns the gradient with respect to that block's PRE-activation (the
    private convention of ``RingSegmentH``)."""

    @staticmethod
    def forward(ctx, y, act):
        lib = _lib.load()
        N, H, W, C = y.shape
        feat = torch.empty((N, C), dtype=torch.float32, device=y.device)
        _lib.check(lib.dl_mean_hw_nhwc_h(_ptr(y), N, H * W, C, DTYPE_CODE[y.dtype], _ptr(feat), _stream()), "dl_mean_hw_nhwc_h")
        ctx.save_for_backward(y)
        ctx.act = act
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        N, H, W, C = y.shape
        g = torch.empty_like(y)
        _lib.check(lib.dl_mean_hw_bwd_act_h(_ptr(gfeat.contiguous().float()), _ptr(y), N, H * W, C, ctx.act, DTYPE_CODE[y.dtype], _ptr(g),
                                            _stream()), "dl_mean_hw_bwd_act_h")
        return g, None


def ring_trunk_h(x0, act, blocks, dtype, weights, segments=None):
    """layer1..layer4 + global average pooling in half precision: x0 ``[N,H,W,C0]`` fp32 channels-last (the pooled stem output),
    fp32 parameters in block order; returns the pooled features ``[N,C']`` in fp32."""
    x = _run_segments(RingSegmentH, CastToHalf.apply(x0, dtype), act, blocks, list(weights), segments, False)
    return MeanHWActH.apply(x, act)


class RingTrunkH:
    """``RingTrunkH.apply(x0, act, blocks, dtype, *weights)`` = ``ring_trunk_h`` (interface of the first version, one Function)."""

    @staticmethod
    def apply(x0, act, blocks, dtype, *weights):
        return ring_trunk_h(x0, act, blocks, dtype, list(weights))
