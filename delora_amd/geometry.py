"""Batched geometry of the DeLORA training step on MI355X: thin torch wrappers over the C ABI.

Every function takes/returns CUDA(HIP) tensors, allocates outputs with torch's caching allocator,
passes raw device pointers plus torch's *current stream* to libdelora_hip.so, and never
synchronises.  Layouts are those of include/delora_hip.h: planar fp32 range images
``[S,4,H,W]`` (x,y,z,range) and normals ``[S,3,H,W]`` for everything that is streamed, packed
``[S,H,W,4]`` twins ((x,y,z,range) / (nx,ny,nz,0)) for everything that is gathered from, int32 pixel maps.
"""
import ctypes

import torch

from . import _lib

LOSS_POINT_TO_POINT, LOSS_POINT_TO_PLANE, LOSS_PLANE_TO_PLANE, LOSS_NORMAL_LINEAR, LOSS_PO2PO_ALONE = 1, 2, 4, 8, 16


class Sensor:
    """Projection parameters of one dataset, resolved once from the reference's flat config
    (``config[dataset]["vertical_cells"|"horizontal_cells"|"vertical_field_of_view"]`` and
    ``config["horizontal_field_of_view"]``, radians; reference src/utility/projection.py:16,50-52)."""

    def __init__(self, height, width, vfov, hfov):
        self.H, self.W = int(height), int(width)
        self.vfov = (float(vfov[0]), float(vfov[1]))
        self.hfov = (float(hfov[0]), float(hfov[1]))
        self.struct = _lib.SensorStruct(self.H, self.W, self.hfov[0], self.hfov[1], self.vfov[0], self.vfov[1])

    @classmethod
    def from_config(cls, config, dataset):
        d = config[dataset]
        return cls(d["vertical_cells"], d["horizontal_cells"], d["vertical_field_of_view"],
                   config["horizontal_field_of_view"])

    def key(self):
        return (self.H, self.W, self.vfov, self.hfov)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DeloraHipError("delora_amd geometry needs tensors on the GPU (no CPU implementation)")


def _planar(t, channels):
    """(data_ptr tensor, scan stride) of a ``[S,channels,H,W]`` tensor whose inner three dims are dense."""
    S, C, H, W = t.shape
    if C < channels or t.stride(3) != 1 or t.stride(2) != W or t.stride(1) != H * W or t.dtype != torch.float32:
        raise ValueError(f"expected planar fp32 [S,>={channels},H,W] with dense inner dims, got {tuple(t.shape)} / {t.stride()}")
    return t, t.stride(0)


def project(points, offsets, max_points, sensor, want_uv=False, want_kept=True, want_packed=True):
    """Range images of S scans.  points ``[C,sumN]`` fp32 (rows 0..2 = xyz), offsets ``[S+1]`` int32 (device).
    Returns dict(image4 [S,4,H,W], aux [S,C-3,H,W]|None, packed [S,H,W,4]|None, packed_aux [S,H,W,4]|None (C >= 6: the
    stored normals), pix2pt [S,H,W] int32, kept [S] int32|None, uv [3,sumN]|None = u, v and range of every point)."""
    lib = _lib.load()
    _require_cuda(points, offsets)
    if points.dtype != torch.float32 or points.dim() != 2 or points.stride(1) != 1:
        raise ValueError("points must be fp32 [C,sumN] with unit inner stride")
    if offsets.dtype != torch.int32:
        raise ValueError("offsets must be int32")
    C, S = points.shape[0], offsets.numel() - 1
    H, W, dev = sensor.H, sensor.W, points.device
    image4 = torch.empty((S, 4, H, W), dtype=torch.float32, device=dev)
    aux = torch.empty((S, C - 3, H, W), dtype=torch.float32, device=dev) if C > 3 else None
    packed = torch.empty((S, H, W, 4), dtype=torch.float32, device=dev) if want_packed else None
    packed_aux = torch.empty((S, H, W, 4), dtype=torch.float32, device=dev) if (want_packed and C >= 6) else None
    pix2pt = torch.empty((S, H, W), dtype=torch.int32, device=dev)
    kept = torch.empty((S,), dtype=torch.int32, device=dev) if want_kept else None
    n_cols = points.shape[1]
    ws = torch.empty((lib.dl_project_workspace_bytes(S, H, W, n_cols, C) // 8,), dtype=torch.int64, device=dev)
    uv = torch.empty((3, points.shape[1]), dtype=torch.float32, device=dev) if want_uv else None
    if uv is not None and points.stride(0) != uv.stride(0):
        raise ValueError("want_uv needs a dense points buffer")
    _lib.check(lib.dl_project(_ptr(points), points.stride(0), n_cols, _ptr(offsets), S, C, int(max_points),
                              ctypes.byref(sensor.struct), _ptr(image4), _ptr(aux), _ptr(packed), _ptr(packed_aux),
                              _ptr(pix2pt), _ptr(ws), _ptr(kept), _ptr(uv), _stream()), "dl_project")
    return {"image4": image4, "aux": aux, "packed": packed, "packed_aux": packed_aux, "pix2pt": pix2pt, "kept": kept,
            "uv": uv}


def normals(image4, half_rows=3, half_cols=5, epsilon_range=0.5, min_neighbors=10, want_packed=False):
    """Normals ``[S,3,H,W]`` of range images ``[S,>=3,H,W]`` (zero vector = no normal); with ``want_packed`` also the
    packed twin ``[S,H,W,4]`` -> (planar, packed)."""
    lib = _lib.load()
    _require_cuda(image4)
    t, ss = _planar(image4, 3)
    S, _, H, W = t.shape
    out = torch.empty((S, 3, H, W), dtype=torch.float32, device=t.device)
    pk = torch.empty((S, H, W, 4), dtype=torch.float32, device=t.device) if want_packed else None
    _lib.check(lib.dl_normals(_ptr(t), ss, S, H, W, int(half_rows), int(half_cols), float(epsilon_range),
                              int(min_neighbors), _ptr(out), _ptr(pk), _stream()), "dl_normals")
    return (out, pk) if want_packed else out


def pack_image(planar):
    """Packed twin ``[S,H,W,4]`` of a planar ``[S,3|4,H,W]`` tensor (torch ops; for callers that did not get it from
    project()/normals())."""
    S, C, H, W = planar.shape
    out = torch.zeros((S, H, W, 4), dtype=torch.float32, device=planar.device)
    out[..., :min(C, 4)] = planar[:, :4].permute(0, 2, 3, 1)
    return out


def _packed(t):
    S, H, W, C = t.shape
    if C != 4 or t.stride(3) != 1 or t.stride(2) != 4 or t.stride(1) != 4 * W or t.dtype != torch.float32:
        raise ValueError(f"expected packed fp32 [S,H,W,4] with dense inner dims, got {tuple(t.shape)} / {t.stride()}")
    return t, t.stride(0)


def nn_correspond(src_image4, src_normals, tgt_packed, tgt_normals_packed, T, sensor, need_without_normals=False,
                  want_visible=True, want_match=True):
    """Exact nearest target pixel of every transformed source point (source planar, target packed ``[B,H,W,4]``).
    Returns (nn_pix [B,H,W] int32, visible [B]|None, match [B,6,H,W]|None): ``match`` holds, per SOURCE pixel, the matched
    target point (planes 0..2) and normal (planes 3..5) -- the operand stream of icp_loss()."""
    lib = _lib.load()
    _require_cuda(src_image4, tgt_packed, T)
    s, s_ss = _planar(src_image4, 3)
    t, t_ss = _packed(tgt_packed)
    tn, tn_ss = _packed(tgt_normals_packed) if tgt_normals_packed is not None else (None, 0)
    n, n_ss = _planar(src_normals, 3) if src_normals is not None else (None, 0)
    B, H, W = s.shape[0], sensor.H, sensor.W
    Tc = T.detach().contiguous().float()
    nn = torch.empty((B, H, W), dtype=torch.int32, device=s.device)
    match = torch.empty((B, 6, H, W), dtype=torch.float32, device=s.device) if want_match else None
    vis = torch.empty((B,), dtype=torch.int32, device=s.device) if want_visible else None
    ws = torch.empty((lib.dl_nn_workspace_bytes(B, H, W) // 8 + 1,), dtype=torch.int64, device=s.device)
    _lib.check(lib.dl_nn_correspond(_ptr(s), s_ss, _ptr(n), n_ss, _ptr(t), t_ss, _ptr(tn), tn_ss, _ptr(Tc), B,
                                    ctypes.byref(sensor.struct), int(bool(need_without_normals)), _ptr(nn),
                                    _ptr(match), _ptr(vis), _ptr(ws), _stream()), "dl_nn_correspond")
    return nn, vis, match


# bench.py sets this to a callable returning a fresh timer handle (LossTimers.new): the streaming loss kernel of every
# training step is then launched with its own begin/end timestamps attached, so that its duration can be measured
# inside real steps (dl_icp_loss_partial_timed)
LOSS_TIMER_FACTORY = None


class LossTimers:
    """A pool of dl_timer handles; ``new()`` hands one to each launch, ``elapsed_ms()`` reads them all back."""

    def __init__(self, reserve=0):
        """``reserve`` timers are created up front: creating the HIP events of a timer while the stream is busy costs the host
        milliseconds (measured: a 5 ms autocast step became a 10.6 ms step with one timer created per step)."""
        self.lib = _lib.load()
        self.handles = []
        self.used = 0
        for _ in range(int(reserve)):
            self._create()

    def _create(self):
        h = ctypes.c_void_p()
        _lib.check(self.lib.dl_timer_create(ctypes.byref(h)), "dl_timer_create")
        self.handles.append(h)

    def new(self):
        if self.used == len(self.handles):
            self._create()
        self.used += 1
        return self.handles[self.used - 1]

    def elapsed_ms(self):
        out = []
        for h in self.handles[:self.used]:
            ms = ctypes.c_float()
            _lib.check(self.lib.dl_timer_elapsed_ms(h, ctypes.byref(ms)), "dl_timer_elapsed_ms")
            out.append(ms.value)
        return out

    def close(self):
        for h in self.handles:
            self.lib.dl_timer_destroy(h)
        self.handles = []
        self.used = 0


class _IcpLoss(torch.autograd.Function):
    """loss_terms[B,3] = (po2po, po2pl, pl2pl) as a differentiable function of T[B,4,4]; the
    correspondences are constants, exactly as the gather indices are in the reference."""

    @staticmethod
    def forward(ctx, T, src_image4, src_normals, match, nn_pix, flags):
        lib = _lib.load()
        s, s_ss = _planar(src_image4, 3)
        sn, sn_ss = _planar(src_normals, 3)
        mt, mt_ss = _planar(match, 6)
        B, _, H, W = s.shape
        dev = s.device
        Tc = T.detach().contiguous().float()
        loss_terms = torch.empty((B, 3), dtype=torch.float32, device=dev)
        counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
        grad_terms = torch.empty((B, 3, 12), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.dl_icp_loss_workspace_bytes(B, H, W) // 4,), dtype=torch.float32, device=dev)
        if LOSS_TIMER_FACTORY is None:
            _lib.check(lib.dl_icp_loss_partial(_ptr(s), s_ss, _ptr(sn), sn_ss, _ptr(mt), mt_ss, _ptr(nn_pix), _ptr(Tc),
                                               B, H, W, int(flags), _ptr(ws), _stream()), "dl_icp_loss_partial")
        else:                                                           # bench.py: the same launch with timestamps attached
            _lib.check(lib.dl_icp_loss_partial_timed(_ptr(s), s_ss, _ptr(sn), sn_ss, _ptr(mt), mt_ss, _ptr(nn_pix), _ptr(Tc),
                                                     B, H, W, int(flags), _ptr(ws), LOSS_TIMER_FACTORY(), _stream()),
                       "dl_icp_loss_partial_timed")
        _lib.check(lib.dl_icp_loss_reduce(_ptr(ws), B, H, W, int(flags), _ptr(loss_terms), _ptr(counts),
                                          _ptr(grad_terms), _stream()), "dl_icp_loss_reduce")
        ctx.save_for_backward(grad_terms)
        ctx.mark_non_differentiable(counts)
        return loss_terms, counts

    @staticmethod
    def backward(ctx, g_terms, _g_counts):
        lib = _lib.load()
        (grad_terms,) = ctx.saved_tensors
        B = grad_terms.shape[0]
        g = g_terms.contiguous().float()
        grad_T = torch.empty((B, 4, 4), dtype=torch.float32, device=g.device)
        _lib.check(lib.dl_icp_loss_bwd(_ptr(grad_terms), _ptr(g), B, _ptr(grad_T), _stream()), "dl_icp_loss_bwd")
        return grad_T, None, None, None, None, None


def icp_loss(T, src_image4, src_normals, match, nn_pix, flags):
    """(loss_terms [B,3], pair_counts [B,2]); loss_terms is differentiable with respect to T.  Source planes
    ``[B,>=3,H,W]`` / ``[B,3,H,W]``, ``match`` ``[B,6,H,W]`` and ``nn_pix`` from nn_correspond()."""
    _require_cuda(T, src_image4, src_normals, match, nn_pix)
    return _IcpLoss.apply(T, src_image4, src_normals, match, nn_pix, flags)


def loss_flags(config):
    """Flag word of dl_icp_loss_fwd from the reference's hyper-parameters (config/hyperparameters.yaml:14-19).

    ``po2po_alone`` (src/losses/icp_losses.py:36-45) pairs EVERY source point with its nearest target and keeps the
    point-to-point term only; the reference defines no pair lists for the normal-based terms in that branch and dies
    with an UnboundLocalError when one of them is enabled as well (:135-146) -- made explicit here."""
    f = 0
    if config.get("po2po_alone", False):
        if config["point_to_plane_loss"] or config["plane_to_plane_loss"]:
            raise Exception("po2po_alone needs point_to_plane_loss and plane_to_plane_loss switched off "
                            "(the reference fails for this combination, icp_losses.py:36-45,135-146).")
        f |= LOSS_PO2PO_ALONE
    if config["point_to_point_loss"]:
        f |= LOSS_POINT_TO_POINT
    if config["point_to_plane_loss"]:
        f |= LOSS_POINT_TO_PLANE
    if config["plane_to_plane_loss"]:
        f |= LOSS_PLANE_TO_PLANE
    if config["normal_loss"] == "linear":
        f |= LOSS_NORMAL_LINEAR
    elif config["normal_loss"] != "squared":
        raise Exception("The normal loss which is defined here is not admissible.")
    return f


def need_without_normals(config):
    """Whether the search must also answer source points WITHOUT a normal: the point-to-point term uses them
    (icp_losses.py:85-100), and po2po_alone uses every source point (:36-45)."""
    return bool(config["point_to_point_loss"]) or bool(config.get("po2po_alone", False))


def nn_bruteforce(src, tgt):
    """Exact NN index into ``tgt [3,Mt]`` for every column of ``src [3,Ms]`` (free-form lists)."""
    lib = _lib.load()
    _require_cuda(src, tgt)
    src, tgt = src.detach().contiguous().float(), tgt.detach().contiguous().float()
    Ms, Mt = src.shape[1], tgt.shape[1]
    nn = torch.empty((Ms,), dtype=torch.int32, device=src.device)
    _lib.check(lib.dl_nn_bruteforce(_ptr(src), max(Ms, 1), Ms, _ptr(tgt), max(Mt, 1), Mt, _ptr(nn), _stream()),
               "dl_nn_bruteforce")
    return nn


def probe_stream_read(src_image4, src_normals, match, nn_pix):
    """Launch the read-only twin of the loss kernel on the same operands (measurement aid, see dl_probe_stream_read)."""
    lib = _lib.load()
    s, s_ss = _planar(src_image4, 3)
    sn, sn_ss = _planar(src_normals, 3)
    mt, mt_ss = _planar(match, 6)
    B, _, H, W = s.shape
    ws = torch.empty((lib.dl_icp_loss_workspace_bytes(B, H, W) // 4,), dtype=torch.float32, device=s.device)
    _lib.check(lib.dl_probe_stream_read(_ptr(s), s_ss, _ptr(sn), sn_ss, _ptr(mt), mt_ss, _ptr(nn_pix), B, H, W, _ptr(ws),
                                        _stream()), "dl_probe_stream_read")
