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


def test_f32_stage_entry_points_validate_before_launch(built_lib):
    """LaMa / float32-CLIP entry points reject bad arguments with a message and no launch (checkable without a GPU)"""
    import ctypes
    from domain_rag_amd import _lib
    a = _lib.Conv2dF32Args()
    assert built_lib.drag_conv2d_f32(ctypes.byref(a), None) != 0 and b"null" in built_lib.drag_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    a.x = a.w = a.y = p.value
    a.B, a.Hi, a.Wi, a.Cin, a.ldx, a.Ho, a.Wo, a.Cout, a.ldy = 1, 4, 4, 6, 6, 4, 4, 8, 8
    a.KH = a.KW = 3; a.stride = 1; a.pad = 1
    assert built_lib.drag_conv2d_f32(ctypes.byref(a), None) != 0 and b"multiples of 4" in built_lib.drag_last_error()
    a.Cin = a.ldx = 8; a.Ho = 5
    assert built_lib.drag_conv2d_f32(ctypes.byref(a), None) != 0 and b"output size" in built_lib.drag_last_error()
    a.Ho = 4; a.pad_mode = 1; a.transposed = 1
    assert built_lib.drag_conv2d_f32(ctypes.byref(a), None) != 0 and b"transposed" in built_lib.drag_last_error()
    assert built_lib.drag_rfft2_f32(p, p, p, 1, 4, 4, 6, 6, p, p, None) != 0 and b"multiples of 4" in built_lib.drag_last_error()
    assert built_lib.drag_irfft2_f32(p, p, p, None, 1, 4, 4, 8, 4, 0, p, p, None) != 0 and b"bad shape" in built_lib.drag_last_error()
    assert built_lib.drag_attention_small_f32(p, p, 1, 65, 2, 64, 384, 128, 0.125, None) != 0 and b"T <= 64" in built_lib.drag_last_error()
    assert built_lib.drag_layernorm_f32(p, p, p, p, 4, 2048, 2048, 2048, 1e-5, None) != 0 and b"D <= 1024" in built_lib.drag_last_error()
    assert built_lib.drag_lama_prepare_u8(p, p, p, 8, 8, 4, 8, None) != 0 and b"bad shape" in built_lib.drag_last_error()


def test_gemm_tile_policy_is_a_function_of_the_launch_shape(built_lib):
    """drag_gemm_bf16_choice / drag_gemm_bf16_pair_merges are host-only: the policy's decisions on the shapes DESIGN.md quotes, and
    the property the batch-invariance tests rest on (the policy never looks at data, only at M, N, K)"""
    c = built_lib.drag_gemm_bf16_choice
    assert c(42696, 0, 9216, 3072) == 3 and c(42696, 0, 3072, 15360) == 3          # the headline's Linears: persistent 256x256, the 4-wave kernel (round 5)
    assert c(42696, 0, 9216, 3072 + 64) == 2 and c(9928, 0, 9216, 3072) == 3       # an odd number of K-steps: the 8-wave kernel; the text stream (1404 tiles)
    assert c(9928, 0, 3072, 3072) == 2 and c(9928, 0, 3072, 12288) == 3 and c(1536, 0, 21504, 3072) == 2   # 468 tiles: only with long K loops; configs[1]'s 504 tiles: 8 waves
    assert c(1536, 0, 3072, 15360) == 133                                           # one exact round of 96x192 tiles (configs[1])
    assert c(1536, 0, 12288, 3072) == 2 and c(512, 0, 12288, 3072) == 143           # 288 256x256 tiles beat 1536 96x128 | one exact round of 128x192
    assert c(512, 0, 3072, 3072) == 24 and c(1024, 0, 3072, 3072) == 43 and c(729, 0, 4096, 1152) == 33   # one round of 64x128 | 128x128 | 96x128 ring tiles
    assert c(1458, 0, 4304, 1152) == 0                                              # 408 128x128 tiles, two per CU: t128
    assert c(8, 0, 18432, 3072) == 14                                               # AdaLN modulation: 32-row tiles
    assert c(1024, 512, 9216, 3072) == 2                                            # merged q|k|v pair at batch 1: 216 tiles
    assert c(1024, 512, 3072, 3072) == 143 and c(1024, 512, 3072, 12288) == 143     # (8 + 4) x 16 = 192 tiles of 128x192: one round
    m = built_lib.drag_gemm_bf16_pair_merges
    assert m(1024, 512, 9216, 3072) == 1 and m(1024, 512, 3072, 3072) == 1 and m(1024, 512, 3072, 12288) == 1
    assert m(1024, 512, 12288, 3072) == 0                                           # one 256x256 round + one 128x192 round alone, two 256x256 rounds merged
    assert m(32768, 4096, 9216, 3072) == 0 and m(42696, 4096, 3072, 12288) == 0     # each fills the chip alone
    for name, v in (("gemm_pair", 1), ("gemm_pair", 2)):
        assert built_lib.drag_set_option(name.encode(), v) == 0
        assert m(1024, 512, 9216, 3072) == (0 if v == 1 else 1) and m(32768, 4096, 9216, 3072) == (0 if v == 1 else 1)
    assert built_lib.drag_set_option(b"gemm_pair", 0) == 0
    k = built_lib.drag_gemm_bf16_cost
    # the single block's to_q|k|v + proj_mlp: fused at 1536 rows (two exact 256x256 rounds), not at the headline's 42 696 (55 vs 24 + 32 rounds)
    assert k(1536, 0, 21504, 3072) * 10 <= (k(1536, 0, 9216, 3072) + k(1536, 0, 12288, 3072)) * 9
    assert k(42696, 0, 21504, 3072) * 10 > (k(42696, 0, 9216, 3072) + k(42696, 0, 12288, 3072)) * 9
