"""The C-ABI library loads and exports every symbol include/dcap.h declares (no compute: CPU-safe)."""
import os
import re

import pytest


def test_library_builds_loads_and_exports_every_declared_symbol(repo_root):
    import __graft_entry__ as G
    G.build()
    from image_captioning_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(repo_root, "include", "dcap.h")).read()
    declared = set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dc_version() == _lib.ABI_VERSION == 600


def test_missing_library_fails_loudly(monkeypatch):
    from image_captioning_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdcap_hip.so")
    with pytest.raises(_lib.DcapError, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from image_captioning_amd import ops, _lib
    with pytest.raises(_lib.DcapError):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8))
