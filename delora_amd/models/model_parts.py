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
