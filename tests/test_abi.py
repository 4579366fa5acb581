"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "domainrag_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(drag_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(built_lib):
    from domain_rag_amd import _lib
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/domainrag_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_version_and_error_plumbing(built_lib):
    assert built_lib.drag_version() >= 100
    # argument validation happens before any launch, so it is checkable without a GPU
    rc = built_lib.drag_gemm_bf16(None, None)
    assert rc != 0
    assert b"null" in built_lib.drag_last_error()
    rc = built_lib.drag_cosine_topk_f32(None, None, 10, 512, 1, 5, None, None, None, None)
    assert rc != 0 and b"null" in built_lib.drag_last_error()


def test_ops_refuse_cpu_tensors(built_lib):
    import pytest
    import torch
    from domain_rag_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


def test_missing_library_fails_loudly(monkeypatch, built_lib):
    import pytest
    from domain_rag_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdomainrag_hip.so")
    with pytest.raises(RuntimeError, match="no CPU"):
        _lib.load()
