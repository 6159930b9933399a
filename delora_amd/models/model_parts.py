"""Parameter-free building blocks of the pose network: the horizontal wrap-around padding of a 360 degree
range image and the quaternion -> SE(3) assembly (reference src/models/model_parts.py:16-44)."""
import torch
import torch.nn.functional as F


class CircularPad(torch.nn.Module):
    """Wrap-around padding; the default pads one column on each side of W and nothing on H
    (reference model_parts.py:16-22), because column 0 and column W-1 of a spinning-LiDAR image are neighbours."""

    def __init__(self, padding=(1, 1, 0, 0)):
        super().__init__()
        self.padding = tuple(padding)

    def forward(self, input):
        return F.pad(input, self.padding, mode="circular")

    def extra_repr(self):
        return f"padding={self.padding}"


class _QuatToT(torch.autograd.Function):
    EPS = 1e-12

    @staticmethod
    def forward(ctx, translation, quaternion):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        t, q = translation.contiguous(), quaternion.contiguous()
        B = q.shape[0]
        T = torch.empty((B, 4, 4), dtype=torch.float32, device=q.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.dl_quat_to_T_fwd(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(q.data_ptr()), B, _QuatToT.EPS,
                                        ctypes.c_void_p(T.data_ptr()), st), "dl_quat_to_T_fwd")
        ctx.save_for_backward(q)
        return T

    @staticmethod
    def backward(ctx, grad_T):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        (q,) = ctx.saved_tensors
        g = grad_T.contiguous()
        B = q.shape[0]
        gt = torch.empty((B, 3), dtype=torch.float32, device=q.device)
        gq = torch.empty((B, 4), dtype=torch.float32, device=q.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.dl_quat_to_T_bwd(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(g.data_ptr()), B, _QuatToT.EPS,
                                        ctypes.c_void_p(gt.data_ptr()), ctypes.c_void_p(gq.data_ptr()), st), "dl_quat_to_T_bwd")
        return gt, gq


class GeometryHandler:
    """Quaternion (x, y, z, w) + translation -> homogeneous transform, differentiable torch ops.

    The reference delegates the rotation to kornia 0.3.0 (model_parts.py:30-31, pinned in
    conda/DeLORA-py3.9.yml:53), which is not part of the reference tree: its published formula is restated
    here -- normalise with eps 1e-12, then R = (1 - 2 v.v) I + 2 v v^T + 2 w [v]x written out per element.
    """

    def __init__(self, config):
        self.device = config["device"]

    @staticmethod
    def quaternion_to_rot_matrix(quaternion):
        q = F.normalize(quaternion, p=2.0, dim=-1, eps=1e-12)
        x, y, z, w = q.unbind(dim=-1)
        xx, yy, zz = 2.0 * x * x, 2.0 * y * y, 2.0 * z * z
        xy, xz, yz = 2.0 * y * x, 2.0 * z * x, 2.0 * z * y
        wx, wy, wz = 2.0 * x * w, 2.0 * y * w, 2.0 * z * w
        rows = (torch.stack((1.0 - (yy + zz), xy - wz, xz + wy), dim=-1),
                torch.stack((xy + wz, 1.0 - (xx + zz), yz - wx), dim=-1),
                torch.stack((xz - wy, yz + wx, 1.0 - (xx + yy)), dim=-1))
        return torch.stack(rows, dim=-2).reshape(-1, 3, 3)

    @staticmethod
    def get_transformation_matrix_quaternion(translation, quaternion, device):
        """``T[B,4,4] = [[R, t], [0, 1]]`` (reference model_parts.py:38-44).  fp32 CUDA tensors take one HIP kernel each way
        (dl_quat_to_T_fwd / _bwd: the ~35 element-wise torch ops below and their autograd are ~100 launches per step)."""
        if translation.is_cuda and quaternion.is_cuda and translation.dtype == torch.float32 and quaternion.dtype == torch.float32:
            return _QuatToT.apply(translation.reshape(-1, 3), quaternion.reshape(-1, 4))
        R = GeometryHandler.quaternion_to_rot_matrix(quaternion)
        B = R.shape[0]
        top = torch.cat((R, translation.reshape(B, 3, 1).to(R.dtype)), dim=2)
        bottom = torch.cat((R.new_zeros((B, 1, 3)), R.new_ones((B, 1, 1))), dim=2)   # device-side fills: graph-capturable
        return torch.cat((top, bottom), dim=1)


class FusedHeads(torch.autograd.Function):
    """fc -> the two two-layer heads -> whole-batch quaternion norm as ``dl_heads_fwd`` / ``dl_heads_bwd`` (csrc/heads.hip): three launches
    forward and four backward instead of ~45 small library launches (reference: src/models/resnet_modified.py:118-120 ``fc``,
    src/models/model.py:74-83 the heads, :114 the norm).  ``forward(x [B,F], act, fc.w, fc.b, rot.1.w, rot.1.b, rot.3.w, rot.3.b,
    tr.1.w, tr.1.b, tr.3.w, tr.3.b) -> (translation [B,3], rotation [B,4])``; fp32 CUDA tensors, B <= 16."""

    ORDER = ("fc_w", "fc_b", "r1_w", "r1_b", "r3_w", "r3_b", "t1_w", "t1_b", "t3_w", "t3_b")

    @staticmethod
    def _struct(tensors):
        from .. import _lib
        st = _lib.HeadsParams()
        for name, t in zip(FusedHeads.ORDER, tensors):
            setattr(st, name, t.data_ptr())
        return st

    @staticmethod
    def forward(ctx, x, act, *params):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        x = x.contiguous()
        params = tuple(p.contiguous() for p in params)
        B, F = x.shape
        R, Hd = params[0].shape[0], params[2].shape[0]
        dev = x.device
        a1 = torch.empty((B, R), dtype=torch.float32, device=dev)
        a2 = torch.empty((B, 2, Hd), dtype=torch.float32, device=dev)
        # the two outputs are tensors of their own: views of one scratch buffer that is also saved for the backward would share a
        # version counter with a saved tensor, and an in-place op of a caller on either output would trip autograd (advisor, round 4)
        small = torch.empty((B * 4 + 1,), dtype=torch.float32, device=dev)           # rot_raw [B,4] | norm: what the backward needs
        rot_raw, norm = small[:4 * B].view(B, 4), small[4 * B:]
        translation = torch.empty((B, 3), dtype=torch.float32, device=dev)
        rotation = torch.empty((B, 4), dtype=torch.float32, device=dev)
        st = FusedHeads._struct(params)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())                                  # noqa: E731
        _lib.check(lib.dl_heads_fwd(vp(x), ctypes.byref(st), B, F, R, Hd, int(act), vp(a1), vp(a2), vp(rot_raw), vp(translation), vp(rotation),
                                    vp(norm), stream), "dl_heads_fwd")
        ctx.act = int(act)
        ctx.save_for_backward(x, a1, a2, small, *params)
        return translation, rotation

    @staticmethod
    def backward(ctx, g_translation, g_rotation):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        x, a1, a2, small, *params = ctx.saved_tensors
        B, F = x.shape
        R, Hd = params[0].shape[0], params[2].shape[0]
        rot_raw, norm = small[:4 * B], small[4 * B:]
        dev = x.device
        gt = (g_translation if g_translation is not None else torch.zeros((B, 3), device=dev)).contiguous().float()
        gr = (g_rotation if g_rotation is not None else torch.zeros((B, 4), device=dev)).contiguous().float()
        grads = [torch.empty_like(p) for p in params]
        gx = torch.empty_like(x)
        ws = torch.empty((lib.dl_heads_bwd_workspace_bytes(B, F, R, Hd) // 4,), dtype=torch.float32, device=dev)
        st, gs = FusedHeads._struct(params), FusedHeads._struct(grads)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())                                  # noqa: E731
        _lib.check(lib.dl_heads_bwd(vp(x), ctypes.byref(st), B, F, R, Hd, ctx.act, vp(a1), vp(a2), vp(rot_raw), vp(norm), vp(gt), vp(gr),
                                    ctypes.byref(gs), vp(gx), vp(ws), stream), "dl_heads_bwd")
        return (gx, None, *grads)
