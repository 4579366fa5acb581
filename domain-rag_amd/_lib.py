"""ctypes loader for libdomainrag_hip.so — the only way host code reaches the HIP kernels."""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libdomainrag_hip.so")
if os.environ.get("DRAG_LIB"):          # A/B across builds of the library (scripts/): another libdomainrag_hip.so, same ABI
    LIB_PATH = os.environ["DRAG_LIB"]

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class GemmArgs(ctypes.Structure):
    """mirror of ``drag_gemm_args`` (include/domainrag_hip.h)"""
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("gate", c_void_p), ("resid", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("a_rows_per_batch", c_int), ("a_batch_stride", c_int64),
        ("ldc", c_int), ("c_rows_per_batch", c_int), ("c_batch_stride", c_int64),
        ("ldg", c_int), ("act", c_int), ("act_n0", c_int), ("out_f32", c_int),
        ("C2", c_void_p), ("ldc2", c_int), ("n_split", c_int),
    ]


class ConvArgs(ctypes.Structure):
    """mirror of ``drag_conv_args``"""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("bias", c_void_p), ("resid", c_void_p)] + \
               [(n, c_int) for n in ("B", "Ho", "Wo", "Hp", "Wp", "Cin", "Cout", "ldy", "stride", "oy", "ox", "act")]


class ResampleArgs(ctypes.Structure):
    """mirror of ``drag_resample_args``"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("tmp", c_void_p), ("batch", c_int), ("channels", c_int),
                ("src_h", c_int), ("src_w", c_int), ("src_image_stride", c_int64), ("src_row_stride", c_int),
                ("out_h", c_int), ("out_w", c_int), ("dst_image_stride", c_int64), ("dst_row_stride", c_int),
                ("kx", c_void_p), ("bx", c_void_p), ("ksize_x", c_int), ("ky", c_void_p), ("by", c_void_p), ("ksize_y", c_int),
                ("tmp_row0", c_int), ("tmp_rows", c_int), ("src_col0", c_int), ("src_row0", c_int)]


class Conv2dF32Args(ctypes.Structure):
    """mirror of ``drag_conv2d_f32_args``"""
    _fields_ = [(n, c_void_p) for n in ("x", "w", "y", "scale", "shift", "addend", "resid")] + \
               [(n, c_int) for n in ("B", "Hi", "Wi", "Cin", "ldx", "Ho", "Wo", "Cout", "ldy", "ld_add", "ld_res",
                                     "KH", "KW", "stride", "pad", "pad_mode", "transposed", "act")]


# name -> (restype, argtypes); every symbol include/domainrag_hip.h declares
SIGNATURES = {
    "drag_version": (c_int, []),
    "drag_last_error": (ctypes.c_char_p, []),
    "drag_set_option": (c_int, [ctypes.c_char_p, c_int]),
    "drag_experiments_built": (c_int, []),
    "drag_gemm_bf16": (c_int, [ctypes.POINTER(GemmArgs), c_void_p]),
    "drag_gemm_bf16_pair": (c_int, [ctypes.POINTER(GemmArgs), ctypes.POINTER(GemmArgs), c_void_p]),
    "drag_gemm_bf16_pair_merges": (c_int, [c_int, c_int, c_int, c_int]),
    "drag_gemm_bf16_choice": (c_int, [c_int, c_int, c_int, c_int]),
    "drag_conv3x3_bf16_choice": (c_int, [c_int64, c_int, c_int]),
    "drag_gemm_bf16_cost": (c_int64, [c_int, c_int, c_int, c_int]),
    "drag_gemm_set_workspace": (c_int, [c_void_p, c_int64]),
    "drag_gemm_bf16_splitk_slices": (c_int, [ctypes.POINTER(GemmArgs)]),
    "drag_gemm_bf16_pair_splitk_slices": (c_int, [ctypes.POINTER(GemmArgs), ctypes.POINTER(GemmArgs)]),
    "drag_qk_norm_rope_vt_bf16": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_float, c_void_p]),
    "drag_attention_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_int64, c_int, c_int64, c_float, c_void_p]),
    "drag_k_norm_rope_vt_bf16": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_float, c_void_p]),
    "drag_attention_qprep_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_int64, c_int, c_int64, c_float] + [c_void_p] * 4 + [c_int, c_float, c_void_p]),
    "drag_attention_bf16_choice": (c_int, [c_int, c_int, c_int]),
    "drag_attention_v_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_int64, c_int, c_int64, c_float] + [c_void_p] * 4 + [c_int, c_float, c_void_p]),
    "drag_layernorm_modulate_bf16": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_int64, c_int, c_int, c_float, c_void_p]),
    "drag_act_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "drag_timestep_embedding_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "drag_flow_euler_step_bf16": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "drag_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "drag_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "drag_cast_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "drag_cosine_topk_workspace_bytes": (c_int64, [c_int64, c_int]),
    "drag_cosine_topk_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "drag_cosine_scores_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "drag_l2_normalize_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "drag_patchify_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "drag_resample_u8": (c_int, [ctypes.POINTER(ResampleArgs), c_void_p]),
    "drag_scale_sum_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "drag_resnet_stem_style_workspace_bytes": (c_int64, [c_int] * 3),
    "drag_resnet_stem_style_f32": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p, c_void_p]),
    "drag_patchify_f32_nchw": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "drag_conv3x3_bf16": (c_int, [ctypes.POINTER(ConvArgs), c_void_p]),
    "drag_groupnorm_workspace_bytes": (c_int64, [c_int] * 4),
    "drag_groupnorm_silu_bf16": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_float, c_void_p, c_void_p]),
    "drag_pad_copy_bf16": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "drag_softmax_rows_f32_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "drag_unpack_latents_bf16": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_float, c_float, c_void_p]),
    "drag_sample_pack_latents_bf16": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_float, c_float, c_void_p]),
    "drag_image_preprocess_u8": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "drag_image_postprocess_u8": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "drag_mask_pack_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "drag_flow_euler_rows_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "drag_scale_noise_rows_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "drag_conv2d_f32": (c_int, [ctypes.POINTER(Conv2dF32Args), c_void_p]),
    "drag_rfft2_f32": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p] * 3),
    "drag_irfft2_f32": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p] * 3),
    "drag_lama_prepare_u8": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "drag_vit_prepare_u8": (c_int, [c_void_p, c_void_p, c_int64, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "drag_vit_prepare_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "drag_layernorm_f32": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int64, c_int64, c_float, c_void_p]),
    "drag_clip_embed_ln_f32": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_float, c_void_p]),
    "drag_attention_small_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_float, c_void_p]),
    "drag_lama_blend_u8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "drag_cv_resize_linear_u8_f32": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "drag_file_sizes": (c_int, [c_void_p, c_int64, c_void_p, c_int]),
    "drag_read_files": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int]),
    "drag_jpeg_parse": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "drag_jpeg_decode_rgb": (c_int, [c_void_p] * 4 + [c_int, c_int64, c_int64, c_void_p, c_int64] + [c_void_p] * 5),
    "drag_png_plan": (c_int, [c_int] * 4 + [c_void_p, c_void_p]),
    "drag_png_encode": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the library, binding every declared symbol.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "domain-rag_amd has no CPU / eager fallback.")
    # torch must initialise ITS HIP runtime first: libdomainrag_hip.so then binds to the already-loaded libamdhip64
    # instead of pulling a second copy into the process (which cannot see the GPU)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().drag_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
