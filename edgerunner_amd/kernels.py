"""Torch-tensor wrappers over the single-kernel C-ABI entry points (``er_k_*``).
Used by the GPU unit tests and for debugging; the product path goes through
``NativeShapeOPT`` / ``er_decode``."""
from __future__ import annotations

import ctypes as C

import torch

from . import native


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemv(w, x, bias=None, ln_w=None, ln_b=None, resid=None, relu=False, eps=1e-5, return_xnorm=False):
    """y[b,n] = act(W x_b + bias) (+resid); LayerNorm(x) first when ln_w is given."""
    lib = native.load_library()
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    xn = torch.empty_like(x) if return_xnorm else None
    native.check(lib.er_k_gemv(native.ptr(w), native.ptr(bias), native.ptr(x), native.ptr(ln_w), native.ptr(ln_b),
                               native.ptr(resid), native.ptr(y), native.ptr(xn), B, N, K, int(relu), float(eps), _st()),
                 "er_k_gemv")
    return (y, xn) if return_xnorm else y


def attn_decode(q, k_cache, v_cache, lens, steps=4, variant=native.ER_ATTN_SPLIT2):
    """q [B,H*D] fp32; caches [B,H,Lcap,D] fp32 or fp16; lens: list[int] -> out [B,H*D].
    variant: native.ER_ATTN_SPLIT2 (single-row fallback), ER_ATTN_SPLIT1 (mid-size batches) or ER_ATTN_STREAM (B * heads >= 256)."""
    lib = native.load_library()
    B, H, Lcap, D = k_cache.shape
    out = torch.empty((B, H * D), dtype=torch.float32, device=q.device)
    native.check(lib.er_k_attn_decode(native.ptr(q), native.ptr(k_cache), native.ptr(v_cache), native.i32_array(lens),
                                      native.ptr(out), B, H, D, Lcap, steps, int(k_cache.dtype == torch.float16), int(variant), _st()),
                 "er_k_attn_decode")
    return out


def attn_outproj3(q, k_cache, v_cache, length, wo, bo, resid):
    """Version-3 single-row path: y = Wo . attention(q, K[:length], V[:length]) + bo + resid.
    q [1536] fp32; caches [16,Lcap,96] fp32 or fp16; wo [1536,1536] fp32 or fp16."""
    lib = native.load_library()
    H, Lcap, D = k_cache.shape
    assert (H, D) == (16, 96)
    y = torch.empty((H * D,), dtype=torch.float32, device=q.device)
    native.check(lib.er_k_attn_outproj3(native.ptr(q), native.ptr(k_cache), native.ptr(v_cache), int(length), native.ptr(wo),
                                        native.ptr(bo), native.ptr(resid), native.ptr(y), Lcap,
                                        int(k_cache.dtype == torch.float16), int(wo.dtype == torch.float16), _st()),
                 "er_k_attn_outproj3")
    return y


def gemm(a, b, bias=None, resid=None, b_is_kn=False, relu=False, div=0.0, m=None, n=None, k=None):
    """C = A.op(B) with row strides taken from the (2-D, row-contiguous) tensors."""
    lib = native.load_library()
    M = a.shape[0] if m is None else m
    K = a.shape[1] if k is None else k
    N = (b.shape[1] if b_is_kn else b.shape[0]) if n is None else n
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    native.check(lib.er_k_gemm(native.ptr(a), native.ptr(b), native.ptr(bias), native.ptr(resid), native.ptr(c), M, N, K,
                               a.stride(0), b.stride(0), c.stride(0), int(b_is_kn), int(relu), float(div), _st()),
                 "er_k_gemm")
    return c


def gemm_f16(a, w_half, bias=None, resid=None, relu=False):
    """C = relu?(fp16(A) . W^T + bias) (+resid) on the fp16-input MFMA path; a fp32 [M,K], w_half fp16 [N,K]."""
    lib = native.load_library()
    M, K = a.shape
    N = w_half.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    native.check(lib.er_k_gemm_f16(native.ptr(a), native.ptr(w_half), native.ptr(bias), native.ptr(resid), native.ptr(c), M, N, K,
                                   a.stride(0), w_half.stride(0), c.stride(0), int(relu), _st()), "er_k_gemm_f16")
    return c


def gemm_hh(a, w_half, bias=None, resid=None, relu=False, return_half=False):
    """The same product through the LDS-DMA kernel (both operands fp16 in HBM; a is rounded to an fp16 copy first); K % 64 == 0.
    return_half: also return the epilogue's fp16 copy of the result."""
    lib = native.load_library()
    M, K = a.shape
    N = w_half.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    c16 = torch.empty((M, N), dtype=torch.float16, device=a.device) if return_half else None
    native.check(lib.er_k_gemm_hh(native.ptr(a), native.ptr(w_half), native.ptr(bias), native.ptr(resid), native.ptr(c), native.ptr(c16),
                                  M, N, K, a.stride(0), w_half.stride(0), c.stride(0), int(relu), _st()), "er_k_gemm_hh")
    return (c, c16) if return_half else c


def gemm_hh_qkv(a, w_half, bias, rows_per_batch, force_tile=0):
    """Fused q/k/v projection through the LDS-DMA kernel: returns (qk16 [M, N] whose V third is left as allocated, V^T
    [M / rows_per_batch, heads, 64, rows_per_batch] in the key order of flash_attn_hh_kernel)."""
    lib = native.load_library()
    M, K = a.shape
    N = w_half.shape[0]
    heads = N // 192
    qk16 = torch.zeros((M, N), dtype=torch.float16, device=a.device)
    vt = torch.zeros((M // rows_per_batch, heads, 64, rows_per_batch), dtype=torch.float16, device=a.device)
    native.check(lib.er_k_gemm_hh_qkv(native.ptr(a.contiguous()), native.ptr(w_half.contiguous()), native.ptr(bias), native.ptr(qk16), native.ptr(vt),
                                      M, N, K, rows_per_batch, force_tile, _st()), "er_k_gemm_hh_qkv")
    return qk16, vt


def gemm_hh_geglu(a, w_half, bias, force_tile=0):
    """fp16(GEGLU(fp16(a) . W^T + b)) through the fused LDS-DMA kernel; w_half [2F, K] fp16 in checkpoint order (value rows, gate rows)."""
    lib = native.load_library()
    M, K = a.shape
    F = w_half.shape[0] // 2
    out = torch.empty((M, F), dtype=torch.float16, device=a.device)
    native.check(lib.er_k_gemm_hh_geglu(native.ptr(a.contiguous()), native.ptr(w_half.contiguous()), native.ptr(bias), native.ptr(out), M, F, K,
                                        force_tile, _st()), "er_k_gemm_hh_geglu")
    return out


def gemm_f16s(a, w_half, bias=None, resid=None, relu=False):
    """C = relu?(fp16(A) . W^T + bias) (+resid) on the fp16-input MFMA path; a fp32 [M,K], w_half fp16 [N,K]."""
    lib = native.load_library()
    M, K = a.shape
    N = w_half.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    native.check(lib.er_k_gemm_f16s(native.ptr(a), native.ptr(w_half), native.ptr(bias), native.ptr(resid), native.ptr(c), M, N, K,
                                   a.stride(0), w_half.stride(0), c.stride(0), int(relu), _st()), "er_k_gemm_f16s")
    return c


def flash_attn_f16(q, k, v, heads):
    """q [B,N,H*64], k/v [B,M,H*64] fp32 -> softmax(q k^T / 8) v, [B,N,H*64] (fp16 operands, fp32 accumulate)."""
    lib = native.load_library()
    B, N, _ = q.shape
    M = k.shape[1]
    o = torch.empty_like(q)
    native.check(lib.er_k_flash_attn_f16(native.ptr(q), native.ptr(k), native.ptr(v), native.ptr(o), B, heads, N, M, _st()),
                 "er_k_flash_attn_f16")
    return o


def flash_attn_hh(q, k, v, heads):
    """The same attention through the LDS-DMA kernel (fp16 q / k / v in HBM, V transposed per head; fp16 output widened to fp32)."""
    lib = native.load_library()
    B, N, _ = q.shape
    M = k.shape[1]
    o = torch.empty_like(q)
    native.check(lib.er_k_flash_attn_hh(native.ptr(q), native.ptr(k), native.ptr(v), native.ptr(o), B, heads, N, M, _st()),
                 "er_k_flash_attn_hh")
    return o


def flash_attn_f32(q, k, v, heads, causal=False):
    """q [B,N,H*D], k/v [B,M,H*D] fp32 -> softmax(q k^T / sqrt(D) [+ causal, key j <= i + M - N]) v in exact fp32 on the
    f32-input matrix cores, no score matrix in HBM (head_dim 64 or 96)."""
    lib = native.load_library()
    B, N, HD = q.shape
    M = k.shape[1]
    o = torch.empty_like(q)
    native.check(lib.er_k_flash_attn_f32(native.ptr(q), native.ptr(k), native.ptr(v), native.ptr(o), B, heads, N, M, HD // heads,
                                         int(causal), _st()), "er_k_flash_attn_f32")
    return o



def flash_attn_f16s(q, k, v, heads, causal=False):
    """The same attention for head_dim 96 on the fp16 matrix cores with hi/lo-split q and p; k / v must hold
    fp16-representable values (fast-mode prefix attention of batches)."""
    lib = native.load_library()
    B, N, HD = q.shape
    assert HD // heads == 96
    M = k.shape[1]
    o = torch.empty_like(q)
    native.check(lib.er_k_flash_attn_f16s(native.ptr(q), native.ptr(k), native.ptr(v), native.ptr(o), B, heads, N, M,
                                          int(causal), _st()), "er_k_flash_attn_f16s")
    return o

def layernorm(x, w, b, eps=1e-5):
    lib = native.load_library()
    y = torch.empty_like(x)
    native.check(lib.er_k_layernorm(native.ptr(x), native.ptr(w), native.ptr(b), native.ptr(y), x.shape[0], x.shape[1],
                                    float(eps), _st()), "er_k_layernorm")
    return y


def softmax_(s, cols, causal=False):
    """In place over s [rows, ld]: softmax of the first `cols` (or row+1 if causal) columns, zeros after."""
    lib = native.load_library()
    native.check(lib.er_k_softmax(native.ptr(s), s.shape[0], cols, s.stride(0), int(causal), _st()), "er_k_softmax")
    return s


def sample_head(logits, mode, grammar, step, last_tok, counter, unfinished, top_k=10, min_new=0, seed=0,
                eos=2, pad=0):
    """One sampling-head step. Returns (next_tok, counter, unfinished) lists."""
    lib = native.load_library()
    B, V = logits.shape
    p = native.ErDecodeParams(mode=mode, top_k=top_k, grammar=grammar, max_new_tokens=step + 1,
                              min_new_tokens=min_new, seed=seed)
    nt, co, uo = (C.c_int32 * B)(), (C.c_int32 * B)(), (C.c_int32 * B)()
    native.check(lib.er_k_sample_head(native.ptr(logits), C.byref(p), V, eos, pad, B, step, native.i32_array(last_tok),
                                      native.i32_array(counter), native.i32_array(unfinished), nt, co, uo, _st()),
                 "er_k_sample_head")
    return list(nt), list(co), list(uo)


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., Random123) on the host: counter = 4 words, key = 2 words -> 4 words.
    The same rounds and constants as ``philox4x32_10`` in csrc/k_head.h."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF,
             p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def philox_uniform(seed: int, step: int, row: int) -> float:
    """Host replica of the device sampler's uniform draw (Philox4x32-10, key = seed,
    counter = (step, row, 0, 0), u = (x0 >> 8) * 2^-24)."""
    x = philox4x32_10((step, row, 0, 0), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return float(x[0] >> 8) / 16777216.0
