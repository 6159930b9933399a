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
