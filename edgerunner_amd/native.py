"""ctypes binding of ``libedgerunner_hip.so`` (C ABI: include/edgerunner_hip.h).

PyTorch is plumbing here: it owns the device tensors whose ``data_ptr()`` are
handed to the library and the HIP stream the work is enqueued on.  There is no
fallback: if the HIP extension is missing or fails to load this module raises,
it never routes around it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

PKG = os.path.dirname(os.path.abspath(__file__))
# ER_LIB_PATH: A/B handle of the measurement scripts (another BUILD of the same library, e.g. build/ab/*.so); a path that
# does not exist fails as loudly as a missing default build
LIB_PATH = os.environ.get("ER_LIB_PATH") or os.path.join(PKG, "libedgerunner_hip.so")

ER_F32, ER_F16, ER_BF16 = 0, 1, 2
ER_COND_NONE, ER_COND_POINT, ER_COND_POINT_LATENT = 0, 1, 2
ER_GREEDY, ER_SAMPLE = 0, 1
ER_GRAMMAR_NONE, ER_GRAMMAR_NAIVE9, ER_GRAMMAR_LR_ABSCO = 0, 1, 2
ER_METO_LR_ABSCO, ER_METO_LR = 0, 1
ER_NUM_KERNEL_KINDS = 8

# every symbol include/edgerunner_hip.h declares (tests/test_abi.py checks the .so exports them all)
EXPORTS = [
    "er_abi_version", "er_last_error", "er_create", "er_destroy", "er_load_tensor", "er_finalize_weights",
    "er_kv_reserve", "er_encode_cond", "er_embed_tokens", "er_prefill", "er_logits", "er_feed", "er_decode",
    "er_meto_decode", "er_meto_encode", "er_dit_create", "er_dit_destroy", "er_dit_load_tensor",
    "er_dit_finalize_weights", "er_dit_project_cond", "er_dit_encode_image", "er_dit_forward", "er_dit_sample",
    "er_set_row_streams", "er_plan_decode", "er_ctx_plan", "er_plan_gemm_tile", "er_kernel_kind_name", "er_profile_decode_kernels", "er_profile_decode_kernels_at", "er_last_decode_ms",
    "er_k_gemv", "er_k_attn_decode", "er_k_attn_outproj3", "er_k_gemm", "er_k_gemm_f16", "er_k_gemm_hh", "er_k_gemm_hh_qkv", "er_k_gemm_hh_geglu", "er_k_gemm_f16s", "er_k_flash_attn_f16", "er_k_flash_attn_hh", "er_k_flash_attn_f32", "er_k_flash_attn_f16s", "er_k_layernorm", "er_k_softmax", "er_k_sample_head",
]


ER_ATTN_SPLIT1, ER_ATTN_SPLIT2, ER_ATTN_BALANCED, ER_ATTN_STREAM = 1, 2, 3, 4


class ErDecodePlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("batched", "decode_version", "attn_kernel", "attn_chunks", "merge_launch",
                                         "launches_per_layer")]


class ErConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "hidden_dim", "num_heads", "num_layers", "intermediate_dim", "vocab_size", "max_positions",
        "num_cond_tokens", "point_hidden_dim", "point_num_heads", "point_latent_size", "point_latent_dim",
        "point_freq_dim", "num_face_buckets", "cond_mode", "pad_token_id", "bos_token_id", "eos_token_id",
        "weight_dtype", "kv_dtype")] + [("ln_eps", C.c_float)]


class ErDitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden_dim", "num_heads", "num_layers", "latent_size", "latent_dim", "clip_dim",
                                         "clip_layers", "clip_heads", "clip_mlp_dim", "clip_image_size", "clip_patch", "weight_dtype")]


class ErDecodeParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("top_k", C.c_int32), ("grammar", C.c_int32),
                ("max_new_tokens", C.c_int32), ("min_new_tokens", C.c_int32), ("seed", C.c_uint64)]


class NativeError(RuntimeError):
    pass


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the HIP extension (no GPU needed just to load it and list symbols)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NativeError(
            f"HIP extension not built: {p} is missing. Run `python -m edgerunner_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for this path.")
    lib = C.CDLL(p)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.er_abi_version.restype = ci
    lib.er_last_error.restype = C.c_char_p
    lib.er_kernel_kind_name.restype = C.c_char_p
    lib.er_kernel_kind_name.argtypes = [ci]
    lib.er_create.argtypes = [C.POINTER(ErConfig), ci, C.POINTER(vp)]
    lib.er_destroy.argtypes = [vp]
    lib.er_load_tensor.argtypes = [vp, C.c_char_p, vp, ci, ci, C.POINTER(C.c_int64), ci]
    lib.er_finalize_weights.argtypes = [vp]
    lib.er_kv_reserve.argtypes = [vp, ci, ci]
    lib.er_encode_cond.argtypes = [vp, vp, ci, ci, C.POINTER(C.c_int32), vp, vp]
    lib.er_embed_tokens.argtypes = [vp, C.POINTER(C.c_int32), ci, ci, vp, vp]
    lib.er_prefill.argtypes = [vp, vp, ci, ci, vp]
    lib.er_logits.argtypes = [vp, vp, vp]
    lib.er_feed.argtypes = [vp, C.POINTER(C.c_int32), vp]
    lib.er_decode.argtypes = [vp, C.POINTER(ErDecodeParams), vp, C.POINTER(C.c_int32), vp]
    lib.er_profile_decode_kernels.argtypes = [vp, ci, C.POINTER(C.c_float), C.POINTER(C.c_double), vp]
    lib.er_profile_decode_kernels_at.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_float), C.POINTER(C.c_double), vp]
    lib.er_last_decode_ms.argtypes = [vp, C.POINTER(C.c_float)]
    i32p = C.POINTER(C.c_int32)
    lib.er_meto_decode.argtypes = [i32p, ci, ci, ci, C.POINTER(C.c_float), i32p, i32p, i32p, i32p, i32p]
    lib.er_meto_encode.argtypes = [C.POINTER(C.c_float), ci, i32p, ci, ci, ci, i32p, i32p, i32p, i32p, i32p]
    lib.er_dit_create.argtypes = [C.POINTER(ErDitConfig), ci, C.POINTER(vp)]
    lib.er_dit_destroy.argtypes = [vp]
    lib.er_dit_load_tensor.argtypes = [vp, C.c_char_p, vp, ci, ci, C.POINTER(C.c_int64), ci]
    lib.er_dit_finalize_weights.argtypes = [vp]
    lib.er_dit_project_cond.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.er_dit_encode_image.argtypes = [vp, vp, ci, ci, ci, vp, vp]
    lib.er_dit_forward.argtypes = [vp, vp, vp, C.POINTER(C.c_float), ci, ci, vp, vp]
    lib.er_dit_sample.argtypes = [vp, vp, ci, ci, vp, ci, cf, ci, vp]
    lib.er_plan_decode.argtypes = [ci, ci, ci, ci, ci, C.POINTER(ErDecodePlan)]
    lib.er_ctx_plan.argtypes = [vp, C.POINTER(ErDecodePlan)]
    lib.er_set_row_streams.argtypes = [vp, C.POINTER(C.c_uint32), ci]
    lib.er_plan_gemm_tile.argtypes = [ci, ci, ci]
    lib.er_k_gemv.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp]
    lib.er_k_attn_decode.argtypes = [vp, vp, vp, C.POINTER(C.c_int32), vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.er_k_attn_outproj3.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, vp]
    lib.er_k_gemm.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp]
    lib.er_k_gemm_f16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.er_k_gemm_f16s.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.er_k_gemm_hh.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.er_k_gemm_hh_qkv.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.er_k_gemm_hh_geglu.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.er_k_flash_attn_f16.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.er_k_flash_attn_hh.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.er_k_flash_attn_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.er_k_flash_attn_f16s.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.er_k_layernorm.argtypes = [vp, vp, vp, vp, ci, ci, cf, vp]
    lib.er_k_softmax.argtypes = [vp, ci, ci, ci, ci, vp]
    lib.er_k_sample_head.argtypes = [vp, C.POINTER(ErDecodeParams), ci, ci, ci, ci, ci, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("er_last_error", "er_kernel_kind_name"):
            fn.restype = ci
    if lib.er_abi_version() != 1:
        raise NativeError(f"ABI version mismatch: library {lib.er_abi_version()}, binding 1")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = load_library().er_last_error().decode(errors="replace")
        raise NativeError(f"{what} failed ({rc}): {msg}")
    return rc


def i32_array(values: Sequence[int]):
    return (C.c_int32 * len(values))(*[int(v) for v in values])


def ptr(t) -> C.c_void_p:
    """Device (or host) pointer of a contiguous torch tensor; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr(stream) -> C.c_void_p:
    """hipStream_t of a torch.cuda.Stream (or the current stream when None)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
