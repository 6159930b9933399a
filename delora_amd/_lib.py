"""ctypes binding of libdelora_hip.so (include/delora_hip.h).

The library is the only implementation of the geometry path: if it is missing or cannot be
loaded this module raises -- there is no CPU or torch fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdelora_hip.so")
ABI_VERSION = 7


class DeloraHipError(RuntimeError):
    pass


class SensorStruct(ctypes.Structure):
    """``dl_sensor`` of include/delora_hip.h."""
    _fields_ = [("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("hfov0", ctypes.c_double), ("hfov1", ctypes.c_double),
                ("vfov0", ctypes.c_double), ("vfov1", ctypes.c_double)]


_vp, _i32, _i64, _u32, _f32, _sz = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32,
                                    ctypes.c_float, ctypes.c_size_t)
_SP = ctypes.POINTER(SensorStruct)

# name -> (restype, argtypes); mirrors the declarations of include/delora_hip.h one to one
SIGNATURES = {
    "dl_abi_version": (_i32, []),
    "dl_last_error": (ctypes.c_char_p, []),
    "dl_project_workspace_bytes": (_sz, [_i32, _i32, _i32, _i64, _i32]),
    "dl_project": (_i32, [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dl_normals": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp]),
    "dl_nn_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "dl_nn_correspond": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _SP, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dl_icp_loss_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "dl_icp_loss_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _u32,
                               _vp, _vp, _vp, _vp, _vp]),
    "dl_icp_loss_partial": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _u32, _vp, _vp]),
    "dl_icp_loss_reduce": (_i32, [_vp, _i32, _i32, _i32, _u32, _vp, _vp, _vp, _vp]),
    "dl_icp_loss_bwd": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "dl_ring_act_pool_pad_fwd": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dl_ring_act_pool_pad_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "dl_timer_create": (_i32, [ctypes.POINTER(ctypes.c_void_p)]),
    "dl_timer_destroy": (_i32, [_vp]),
    "dl_timer_elapsed_ms": (_i32, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "dl_icp_loss_partial_timed": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _u32, _vp, _vp, _vp]),
    "dl_probe_stream_read": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp]),
    "dl_nn_bruteforce": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp]),
    "dl_ring_act_pad_fwd": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "dl_ring_act_pad_bwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dl_ring_act_pad_fwd_t": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dl_ring_act_pad_bwd_t": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dl_ring_act_pool_pad_fwd_t": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dl_ring_act_pool_pad_bwd_t": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dl_conv2d_nhwc_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _u32, _vp]),
    "dl_conv2d_dgrad_strided_nhwc_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                                _u32, _vp, _vp]),
    "dl_conv2d_wgrad_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "dl_conv2d_wgrad_nhwc_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dl_wino_weights_floats": (_sz, [_i32, _i32]),
    "dl_wino_weights_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "dl_wino_weights_batch_f32": (_i32, [_vp, _i32, _vp]),
    "dl_wino_conv3x3_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "dl_wino_conv3x3_nhwc_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _u32, _vp, _vp]),
    "dl_wino_wgrad_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "dl_wino_wgrad3x3_nhwc_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dl_pool3x3s12_nhwc_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dl_pool3x3s12_nhwc_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dl_stem_wgrad_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "dl_stem_wgrad_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dl_quat_to_T_fwd": (_i32, [_vp, _vp, _i32, ctypes.c_float, _vp, _vp]),
    "dl_quat_to_T_bwd": (_i32, [_vp, _vp, _i32, ctypes.c_float, _vp, _vp, _vp]),
    "dl_mean_hw_nhwc_f32": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "dl_conv_weights_h": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dl_conv_weights_batch_h": (_i32, [_vp, _i32, _i32, _vp]),
    "dl_conv2d_nhwc_h": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _u32, _vp]),
    "dl_conv2d_dgrad_strided_nhwc_h": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                              _u32, _vp, _vp]),
    "dl_cast_f32_to_h": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "dl_heads_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dl_heads_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "dl_heads_bwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dl_mean_hw_nhwc_h": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dl_mean_hw_bwd_act_h": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dl_conv2d_wgrad_h_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "dl_conv2d_wgrad_nhwc_h": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dl_wgrad_batch_plan": (_i64, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "dl_wino_wgrad3x3_batch_workspace_bytes": (_sz, [_vp, _i32]),
    "dl_wino_wgrad3x3_batch_nhwc_f32": (_i32, [_vp, _i32, _vp, _vp]),
    "dl_conv2d_wgrad_batch_workspace_bytes": (_sz, [_vp, _i32]),
    "dl_conv2d_wgrad_batch_nhwc_f32": (_i32, [_vp, _i32, _vp, _vp]),
    "dl_conv2d_wgrad_batch_h_workspace_bytes": (_sz, [_vp, _i32]),
    "dl_conv2d_wgrad_batch_nhwc_h": (_i32, [_vp, _i32, _vp, _i32, _vp]),
    "dl_profile_begin": (_i32, [_i32, ctypes.c_char_p]),
    "dl_profile_end": (_i32, [_vp, _i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "dl_profile_pause": (_i32, [_i32]),
}


class WinoLayer(ctypes.Structure):
    """``dl_wino_layer`` of include/delora_hip.h."""
    _fields_ = [("w", ctypes.c_void_p), ("u_fwd", ctypes.c_void_p), ("u_bwd", ctypes.c_void_p), ("K", ctypes.c_int32), ("C", ctypes.c_int32)]


WINO_BATCH = 16


class HeadsParams(ctypes.Structure):
    """``dl_heads_params`` of include/delora_hip.h (ten pointers: parameters, or buffers for their gradients)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("fc_w", "fc_b", "r1_w", "r1_b", "r3_w", "r3_b", "t1_w", "t1_b", "t3_w", "t3_b")]


class ConvHLayer(ctypes.Structure):
    """``dl_convh_layer`` of include/delora_hip.h."""
    _fields_ = [("w", ctypes.c_void_p), ("w_fwd", ctypes.c_void_p), ("w_bwd", ctypes.c_void_p), ("K", ctypes.c_int32), ("taps", ctypes.c_int32),
                ("C", ctypes.c_int32)]


CONVH_BATCH = 32


class WgradLayer(ctypes.Structure):
    """``dl_wgrad_layer`` (= ``dl_wgrad_h_layer``) of include/delora_hip.h."""
    _fields_ = [("x", ctypes.c_void_p), ("g", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("N", ctypes.c_int32), ("H", ctypes.c_int32),
                ("W", ctypes.c_int32), ("C", ctypes.c_int32), ("K", ctypes.c_int32), ("ksize", ctypes.c_int32),
                ("stride_h", ctypes.c_int32), ("stride_w", ctypes.c_int32)]


WgradHLayer = WgradLayer
WGRAD_BATCH = 24


class ProfileRow(ctypes.Structure):
    """``dl_profile_row`` of include/delora_hip.h."""
    _fields_ = [("name", ctypes.c_char * 96), ("launches", ctypes.c_int32), ("ms", ctypes.c_double), ("flop", ctypes.c_double),
                ("bytes", ctypes.c_double)]


def profile_begin(max_launches, only_kernel=None):
    """Open the launch profile; ``only_kernel`` restricts it to one kernel family (e.g. "k_wino_conv")."""
    check(load().dl_profile_begin(int(max_launches), only_kernel.encode() if only_kernel else None), "dl_profile_begin")


def profile_pause(paused):
    """Launches pass untimed while the open profile is paused."""
    check(load().dl_profile_pause(1 if paused else 0), "dl_profile_pause")


def profile_end(capacity=512):
    """Close the launch profile: list of dicts (name, launches, ms, flop, bytes), plus the number of untimed launches."""
    rows = (ProfileRow * capacity)()
    n, untimed = ctypes.c_int32(), ctypes.c_int32()
    check(load().dl_profile_end(ctypes.cast(rows, ctypes.c_void_p), capacity, ctypes.byref(n), ctypes.byref(untimed)), "dl_profile_end")
    out = [{"name": rows[i].name.decode(), "launches": rows[i].launches, "ms": rows[i].ms, "flop": rows[i].flop, "bytes": rows[i].bytes}
           for i in range(min(n.value, capacity))]
    return out, untimed.value

_lib = None


def load():
    """Load (once) and return the bound library; raises DeloraHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeloraHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C delora_amd/csrc`. The geometry path has no fallback implementation.")
    # torch ships its own HIP runtime (libamdhip64); it must be resident BEFORE this library is mapped so that both
    # share one runtime -- loading ours first would pull a second copy from /opt/rocm and split the device context.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise DeloraHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.dl_abi_version() != ABI_VERSION:
        raise DeloraHipError(f"libdelora_hip.so ABI {lib.dl_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise DeloraHipError(f"{what} failed ({status}): {load().dl_last_error().decode()}")
