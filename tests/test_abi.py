"""The C-ABI library builds, loads without a GPU and exports every symbol include/delora_hip.h declares."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "delora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for name in ("dl_project", "dl_normals", "dl_nn_correspond", "dl_icp_loss_fwd", "dl_icp_loss_bwd",
                 "dl_nn_bruteforce", "dl_abi_version", "dl_last_error"):
        assert name in syms


def test_library_loads_and_exports_every_declared_symbol():
    from delora_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.dl_abi_version() == _lib.ABI_VERSION
    assert lib.dl_project_workspace_bytes(2, 64, 2048, 0, 3) == 2 * 64 * 2048 * 8                       # the key plane alone
    assert lib.dl_project_workspace_bytes(2, 64, 2048, 1000, 3) == 2 * 64 * 2048 * 8 + 1000 * 16          # + (x,y,z,range) per point
    assert lib.dl_project_workspace_bytes(2, 64, 2048, 1000, 6) == 2 * 64 * 2048 * 8 + 1000 * 32          # + the stored normals
    assert lib.dl_nn_workspace_bytes(1, 4, 8) >= 4 * 8 * 16
    assert lib.dl_icp_loss_workspace_bytes(8, 64, 2048) > 0


def test_argument_validation_reports_through_last_error():
    from delora_amd import _lib
    lib = _lib.load()
    rc = lib.dl_icp_loss_bwd(None, None, 0, None, None)
    assert rc < 0 and b"dl_icp_loss_bwd" in lib.dl_last_error()
    rc = lib.dl_normals(None, 0, 1, 4, 4, 3, 5, 0.5, 10, None, None, None)
    assert rc < 0 and b"null" in lib.dl_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from delora_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DeloraHipError):
        _lib.load()


def test_cpu_tensors_are_rejected_not_emulated():
    import torch
    from delora_amd import _lib, geometry
    sensor = geometry.Sensor(16, 128, (-0.4, 0.03), (-3.1, 3.1))
    with pytest.raises(_lib.DeloraHipError):
        geometry.normals(torch.zeros(1, 4, 16, 128))


def test_split_k_plan_of_the_winograd_launcher():
    """`dl_wino_conv3x3_workspace_bytes` = the launch plan of `dl_wino_conv3x3_nhwc_f32` seen from outside (pure host arithmetic; without
    a GPU the CU count falls back to 256, an MI355X's): launches that fill the chip take no scratch -- every stride-1 layer of BASELINE's
    64x2048, batch 8 and of the shipped 64x720 at batch 8 --, the reference's default batch 1 at 64x720 splits the input channels of its
    128- / 256- / 512-channel layers over idle CUs (ranges of at least 32 channels, all workgroups in one round of the CUs), and
    64-channel layers are never split."""
    from delora_amd import _lib
    lib = _lib.load()
    ws = lib.dl_wino_conv3x3_workspace_bytes

    def splits(N, H, W, C, K):
        b = ws(N, H, W, C, K)
        assert b % (N * H * W * K * 4) == 0
        return b // (N * H * W * K * 4)

    for (H, W, C) in ((64, 512, 64), (64, 256, 128), (64, 128, 256), (32, 64, 512)):                 # 64x2048, batch 8
        assert splits(8, H, W, C, C) == 0
    for (H, W, C) in ((64, 180, 64), (64, 90, 128), (64, 45, 256), (32, 23, 512)):                   # 64x720, batch 8
        assert splits(8, H, W, C, C) == 0
    assert splits(1, 64, 180, 64, 64) == 0                                                           # 8 chunks: left alone
    assert splits(1, 64, 90, 128, 128) == 4 and splits(1, 64, 45, 256, 256) == 4 and splits(1, 32, 23, 512, 512) == 8
    assert splits(1, 16, 256, 128, 128) == 4 and splits(2, 8, 32, 512, 512) == 16                    # (layer2 / layer4 of the smoke test's 16x1024 pair)
    assert ws(0, 64, 64, 64, 64) == 0 and ws(1, 64, 64, 60, 64) == 0                                # (bad shapes: no plan, the launch rejects them)
