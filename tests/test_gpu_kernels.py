"""Unit parity of every HIP kernel, called through the C ABI (er_k_*), against a
plain torch reference of the same op (fp64 where cheap, fp32 otherwise) computed on
the same device.  Tolerances are fp32 round-off: these are exact-fp32 kernels."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def close(a, b, atol, rtol=0.0, what=""):
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol) | torch.isnan(a)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err {float(err.max()):.3e}, " \
                          f"first bad index {bad.nonzero()[0].tolist()}"


# ------------------------------------------------------------------ GEMV (decode projections)
@pytest.mark.parametrize("B", [1, 2, 3, 5])
def test_gemv_fc1_ln_relu(B):
    from edgerunner_amd import kernels as K
    w, b = rnd(6144, 1536, seed=1, scale=0.02), rnd(6144, seed=2, scale=0.02)
    x = rnd(B, 1536, seed=3) * 3 + 0.5
    lw, lb = 1 + 0.1 * rnd(1536, seed=4), 0.05 * rnd(1536, seed=5)
    y, xn = K.gemv(w, x, b, lw, lb, relu=True, return_xnorm=True)
    xr = torch.nn.functional.layer_norm(x.double(), (1536,), lw.double(), lb.double(), 1e-5)
    close(xn, xr, 2e-6, 2e-6, "layernorm prologue")
    close(y, torch.relu(xr @ w.double().T + b.double()), 2e-6, 1e-5, "fc1")


@pytest.mark.parametrize("B", [1, 4])
def test_gemv_head_ln_store_ragged_rows(B):
    from edgerunner_amd import kernels as K
    w = rnd(518, 1536, seed=6, scale=0.02)          # vocab rows: not a multiple of the rows per workgroup
    x = rnd(B, 1536, seed=7)
    lw, lb = 1 + 0.1 * rnd(1536, seed=8), 0.05 * rnd(1536, seed=9)
    y = K.gemv(w, x, None, lw, lb)
    xr = torch.nn.functional.layer_norm(x.double(), (1536,), lw.double(), lb.double(), 1e-5)
    close(y, xr @ w.double().T, 2e-6, 1e-5, "lm_head")


@pytest.mark.parametrize("B", [1, 2, 3])
def test_gemv_out_proj_resid(B):
    from edgerunner_amd import kernels as K
    w, b = rnd(1536, 1536, seed=10, scale=0.02), rnd(1536, seed=11, scale=0.02)
    x, r = rnd(B, 1536, seed=12), rnd(B, 1536, seed=13)
    y = K.gemv(w, x, b, resid=r)
    close(y, x.double() @ w.double().T + b.double() + r.double(), 2e-6, 1e-5, "out_proj")


@pytest.mark.parametrize("B", [1, 2, 3])
def test_gemv_fc2_ksplit_resid(B):
    from edgerunner_amd import kernels as K
    w, b = rnd(1536, 6144, seed=14, scale=0.02), rnd(1536, seed=15, scale=0.02)
    x, r = torch.relu(rnd(B, 6144, seed=16)), rnd(B, 1536, seed=17)
    y = K.gemv(w, x, b, resid=r)
    close(y, x.double() @ w.double().T + b.double() + r.double(), 4e-6, 1e-5, "fc2")


def _batched_case(B):
    w1, b1 = rnd(6144, 1536, seed=60, scale=0.02), rnd(6144, seed=61, scale=0.02)
    w2, b2 = rnd(1536, 6144, seed=62, scale=0.02), rnd(1536, seed=63, scale=0.02)
    wh = rnd(518, 1536, seed=64, scale=0.02)
    x = rnd(B, 1536, seed=65) * 2 + 0.3
    lw, lb = 1 + 0.1 * rnd(1536, seed=66), 0.05 * rnd(1536, seed=67)
    r = rnd(B, 1536, seed=68)
    return w1, b1, w2, b2, wh, x, lw, lb, r


def _batched_run(c):
    from edgerunner_amd import kernels as K
    w1, b1, w2, b2, wh, x, lw, lb, r = c
    f, xn = K.gemv(w1, x, b1, lw, lb, relu=True, return_xnorm=True)          # fc1-like: LN + ReLU
    y = K.gemv(w2, f, b2, resid=r)                                             # fc2-like: K = 6144 + residual
    lg = K.gemv(wh, x, None, lw, lb)                                           # lm_head-like: ragged N
    o = K.gemv(w2[:, :1536].contiguous(), x, b2, resid=r)                      # out_proj-like
    return f, xn, y, lg, o


@pytest.mark.parametrize("B", [5, 16, 19, 32, 37])
def test_gemv_batched_matrix_core_rows(B):
    """B > 4: fc1 / qkv-shaped and fc2 projections take the matrix-core kernels (k_gemv_mfma.h: weights streamed once
    per pass of 32 rows, split-K for fc2), the narrow out_proj / lm_head the VALU batched kernel: accuracy against
    float64, and a row's result must not depend on which other rows share its pass (same bits whether it runs in a
    5-row or a B-row batch, first or second pass)."""
    c = _batched_case(B)
    w1, b1, w2, b2, wh, x, lw, lb, r = c
    f, xn, y, lg, o = _batched_run(c)
    xr = torch.nn.functional.layer_norm(x.double(), (1536,), lw.double(), lb.double(), 1e-5)
    close(xn, xr, 2e-6, 2e-6, "LayerNorm rows")
    close(f, torch.relu(xr @ w1.double().T + b1.double()), 2e-6, 1e-5, "batched fc1")
    close(y, f.double() @ w2.double().T + b2.double() + r.double(), 4e-6, 1e-5, "batched fc2")
    close(lg, xr @ wh.double().T, 2e-6, 1e-5, "batched lm_head")
    close(o, x.double() @ w2[:, :1536].double().T + b2.double() + r.double(), 2e-6, 1e-5, "batched out_proj")
    idx = [B - 1, 0, B // 2, 1, 2]                                             # 5 rows -> still the batched path
    sub = (w1, b1, w2, b2, wh, x[idx].contiguous(), lw, lb, r[idx].contiguous())
    f5, xn5, _, lg5, o5 = _batched_run(sub)
    from edgerunner_amd import kernels as K
    y5 = K.gemv(w2, f[idx].contiguous(), b2, resid=r[idx].contiguous())
    for got, full, name in ((f5, f, "fc1"), (xn5, xn, "ln"), (y5, y, "fc2"), (lg5, lg, "head"), (o5, o, "out_proj")):
        assert torch.equal(got, full[idx]), f"{name}: a row's bits depend on its batch neighbours"


@pytest.mark.parametrize("B", [5, 19])
def test_gemv_batched_valu_rows_bit_identical_to_single(B, monkeypatch):
    """ER_BATCHED_VALU=1 keeps the older VALU batched kernels (one pass per 16 rows): every row equals, bit for
    bit, the same row pushed through the B = 1 kernel (same fmaf chains and reduction tree)."""
    from edgerunner_amd import kernels as K
    monkeypatch.setenv("ER_BATCHED_VALU", "1")
    c = _batched_case(B)
    w1, b1, w2, b2, wh, x, lw, lb, r = c
    f, xn, y, lg, o = _batched_run(c)
    for i in range(B):
        f1, xn1 = K.gemv(w1, x[i:i + 1].contiguous(), b1, lw, lb, relu=True, return_xnorm=True)
        assert torch.equal(xn1[0], xn[i]), f"LayerNorm row {i} differs from the fused prologue"
        assert torch.equal(f1[0], f[i]), f"fc1 row {i}"
        y1 = K.gemv(w2, f[i:i + 1].contiguous(), b2, resid=r[i:i + 1].contiguous())
        assert torch.equal(y1[0], y[i]), f"fc2 row {i}"
        assert torch.equal(K.gemv(wh, x[i:i + 1].contiguous(), None, lw, lb)[0], lg[i]), f"head row {i}"
        assert torch.equal(K.gemv(w2[:, :1536].contiguous(), x[i:i + 1].contiguous(), b2, resid=r[i:i + 1].contiguous())[0],
                           o[i]), f"out_proj row {i}"


# ------------------------------------------------------------------ decode attention
# contexts beyond 8192 keys: configs[2] runs the decode attention to 18050 keys, the position table allows 43008
LONG_LENS = [(96, [8193, 12000], 4), (96, [18050, 8192], 4), (96, [43008], 4)]


@pytest.mark.parametrize("variant", ["split2", "split1"])
@pytest.mark.parametrize("D,lens,steps", [(96, [2051], 4), (96, [1, 33], 4), (96, [6049, 4000, 17], 4),
                                          (64, [300], 4), (96, [2050, 129], 2), (96, [2050, 255, 256, 257], 8)] + LONG_LENS)
def test_attn_decode(D, lens, steps, variant):
    """split2 = the single-row fallback (reserved caches > 8192 keys, B = 2..4), split1 = the leaner split kernel of the
    4 < B < 16 batches; both feed the same merge kernel (64 partials per pass: > 8192 keys take several passes)."""
    from edgerunner_amd import kernels as K, native
    variant = native.ER_ATTN_SPLIT2 if variant == "split2" else native.ER_ATTN_SPLIT1
    B, H = len(lens), 16
    Lcap = (max(lens) + 31) // 32 * 32
    q = rnd(B, H * D, seed=20)
    kc, vc = rnd(B, H, Lcap, D, seed=21), rnd(B, H, Lcap, D, seed=22)
    # poison the unused tail: must never be read
    for b, n in enumerate(lens):
        kc[b, :, n:] = float("nan")
        vc[b, :, n:] = float("nan")
    out = K.attn_decode(q, kc, vc, lens, steps, variant)
    for b, n in enumerate(lens):
        qq = q[b].view(H, 1, D).double()
        w = torch.softmax(qq @ kc[b, :, :n].double().transpose(1, 2) / math.sqrt(D), dim=-1)
        ref = (w @ vc[b, :, :n].double()).reshape(H * D)
        close(out[b], ref, 2e-6, 1e-5, f"attn row {b} len {n}")


@pytest.mark.parametrize("variant", ["split2", "split1"])
@pytest.mark.parametrize("D,lens,steps", [(96, [2051, 700], 4), (96, [1, 65, 6049], 4), (64, [300], 2), (96, [129], 8)] + LONG_LENS)
def test_attn_decode_fp16_cache(D, lens, steps, variant):
    from edgerunner_amd import kernels as K, native
    variant = native.ER_ATTN_SPLIT2 if variant == "split2" else native.ER_ATTN_SPLIT1
    B, H = len(lens), 16
    Lcap = (max(lens) + 63) // 64 * 64
    q = rnd(B, H * D, seed=23)
    kc, vc = rnd(B, H, Lcap, D, seed=24).half(), rnd(B, H, Lcap, D, seed=25).half()
    for b, n in enumerate(lens):
        kc[b, :, n:] = float("nan")
        vc[b, :, n:] = float("nan")
    out = K.attn_decode(q, kc, vc, lens, steps, variant)
    for b, n in enumerate(lens):
        w = torch.softmax(q[b].view(H, 1, D).double() @ kc[b, :, :n].double().transpose(1, 2) / math.sqrt(D), dim=-1)
        close(out[b], (w @ vc[b, :, :n].double()).reshape(H * D), 2e-6, 1e-5, f"fp16-KV attn row {b} len {n}")


@pytest.mark.parametrize("B,H,N,M,causal", [(2, 16, 2050, 2050, True), (1, 3, 100, 333, False), (1, 2, 33, 33, True)])
def test_flash_attn_f16s(B, H, N, M, causal):
    """Fast-mode prefill attention on the fp16 matrix cores with hi/lo-split q and p (the default for batches of >= 2 prefixes):
    fp16-valued k / v, fp32 q; must agree with float64 to fp32 round-off like the fp32 kernel."""
    from edgerunner_amd import kernels as K
    D = 96
    q = rnd(B, N, H * D, seed=80)
    k, v = rnd(B, M, H * D, seed=81).half().float(), rnd(B, M, H * D, seed=82).half().float()
    o = K.flash_attn_f16s(q, k, v, H, causal=causal)
    qd, kd, vd = (t.double().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = qd @ kd.transpose(2, 3) / math.sqrt(D)
    if causal:
        i = torch.arange(N, device=q.device)[:, None]
        j = torch.arange(M, device=q.device)[None, :]
        s = s.masked_fill(j > i + (M - N), float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ vd).transpose(1, 2).reshape(B, N, H * D)
    close(o, ref, 3e-6, 1e-5, "split-fp16 flash attention")


@pytest.mark.parametrize("lens", [[2051, 700, 1, 65, 6049, 128, 129, 63], [18050, 8193, 12000, 43008, 16383]])
@pytest.mark.parametrize("half", [False, True])
def test_attn_decode_streaming_kernel(half, lens):
    """The batch attention of B * heads >= 256: one workgroup per (row, head) walks the whole key range with a running softmax per
    wave (double-buffered register tiles, no partials, no merge kernel).  Ragged lengths incl. 1 key, a non-multiple of the tile,
    and the long contexts of configs[2] (18050) up to the position table's 43008."""
    from edgerunner_amd import kernels as K, native
    B, H, D = len(lens), 16, 96
    Lcap = (max(lens) + 31) // 32 * 32
    q = rnd(B, H * D, seed=70)
    kc, vc = rnd(B, H, Lcap, D, seed=71), rnd(B, H, Lcap, D, seed=72)
    if half:
        kc, vc = kc.half(), vc.half()
    for b, n in enumerate(lens):
        kc[b, :, n:] = float("nan")
        vc[b, :, n:] = float("nan")
    out = K.attn_decode(q, kc, vc, lens, 4, native.ER_ATTN_STREAM)
    for b, n in enumerate(lens):
        w = torch.softmax(q[b].view(H, 1, D).double() @ kc[b, :, :n].double().transpose(1, 2) / math.sqrt(D), dim=-1)
        close(out[b], (w @ vc[b, :, :n].double()).reshape(H * D), 2e-6, 1e-5, f"streaming attn row {b} len {n}")


@pytest.mark.parametrize("length,lcap,kv16,w16", [(1, 64, False, False), (17, 64, False, False), (129, 160, False, False),
                                                  (2051, 6080, False, False), (4097, 6080, False, False),
                                                  (6049, 6080, False, False), (8192, 8192, False, False),
                                                  (15, 64, True, True), (2063, 6080, True, True), (8191, 8192, True, False)])
def test_attn_outproj3_balanced_chunks_fused_merge(length, lcap, kv16, w16):
    """Version 3 of the single-row decode attention (ER_DECODE_V=3): 16 balanced chunks per head, the partial merge fused
    into the out_proj GEMV.  y = Wo . softmax(q K^T / sqrt(D)) V + bo + resid vs float64 torch; the unused cache tail is
    NaN-poisoned, chunks may be empty (length < 16) or ragged."""
    from edgerunner_amd import kernels as K
    H, D = 16, 96
    q = rnd(H * D, seed=60)
    kc, vc = rnd(H, lcap, D, seed=61), rnd(H, lcap, D, seed=62)
    wo, bo, resid = rnd(H * D, H * D, seed=63) * 0.05, rnd(H * D, seed=64), rnd(H * D, seed=65)
    if kv16:
        kc, vc = kc.half(), vc.half()
    if w16:
        wo = wo.half()
    kc[:, length:] = float("nan")
    vc[:, length:] = float("nan")
    y = K.attn_outproj3(q, kc, vc, length, wo, bo, resid)
    w = torch.softmax(q.view(H, 1, D).double() @ kc[:, :length].double().transpose(1, 2) / math.sqrt(D), dim=-1)
    o = (w @ vc[:, :length].double()).reshape(H * D)
    ref = wo.double() @ o + bo.double() + resid.double()
    close(y, ref, 2e-5, 1e-5, f"attn+out_proj v3 len {length}")


# ------------------------------------------------------------------ MFMA GEMM (prefill / encoder)
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (2050, 1536, 1536), (2050, 6144, 1536), (300, 64, 1024),
                                   (2048, 1024, 64), (77, 200, 96)])
def test_gemm_nt_asymmetric(M, N, K):
    """A=I-style asymmetric operands catch a transposed C fragment map."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=30), rnd(N, K, seed=31)
    bias, resid = rnd(N, seed=32), rnd(M, N, seed=33)
    tol = 1e-5 + 4e-7 * K          # sequential fp32 accumulation of K unit-variance products
    c = K_.gemm(a, w, bias, resid, relu=False)
    close(c, a.double() @ w.double().T + bias.double() + resid.double(), tol, 1e-5, "gemm nt")
    c2 = K_.gemm(a, w, bias, None, relu=True)
    close(c2, torch.relu(a.double() @ w.double().T + bias.double()), tol, 1e-5, "gemm nt relu")


def test_gemm_identity_layout():
    from edgerunner_amd import kernels as K_
    a = torch.eye(256, device=DEV)
    w = (torch.arange(256 * 256, device=DEV, dtype=torch.float32).view(256, 256) % 1009) / 7.0   # asymmetric
    c = K_.gemm(a, w)
    assert torch.equal(c, w.T.contiguous()), "C = I . W^T must reproduce W^T exactly"


@pytest.mark.parametrize("M,N,K", [(2050, 96, 2064), (2048, 64, 4096), (130, 96, 144)])
def test_gemm_nn(M, N, K):
    from edgerunner_amd import kernels as K_
    a, b = rnd(M, K, seed=34), rnd(K, N, seed=35)
    c = K_.gemm(a, b, b_is_kn=True)
    close(c, a.double() @ b.double(), 1e-5 + 4e-7 * K, 1e-5, "gemm nn")


def test_gemm_scale_div():
    from edgerunner_amd import kernels as K_
    a, w = rnd(200, 96, seed=36), rnd(333, 96, seed=37)
    c = K_.gemm(a, w, div=math.sqrt(96))
    ref = (a @ w.T) / (96 ** 0.5)
    close(c, ref, 1e-5, 1e-5, "scores / sqrt(D)")


@pytest.mark.parametrize("M,N,K,tile", [(2050, 4608, 1536, 0), (2050, 1536, 6144, 0), (2050, 6144, 1536, 1), (2048, 1024, 8192, 2),
                                          (300, 1536, 1536, 3), (77, 200, 64, 0), (1, 96, 32, 0)])
def test_gemm_f32_lds_dma(M, N, K, tile, monkeypatch):
    """Exact-mode prefill / encoder Linears on the LDS-DMA fp32 kernel (round 4) against the register-staged kernel they replaced: the
    same (lane half -> k) assignment and k order per accumulator, so the results must agree BIT FOR BIT (bias, ReLU, residual epilogues;
    ragged M / N; every tile shape), and both must match float64 to the fp32 chain's round-off."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=270), rnd(N, K, seed=271, scale=0.05)
    bias, resid = rnd(N, seed=272), rnd(M, N, seed=273)
    if tile:
        monkeypatch.setenv("ER_GEMM_TILE", str(tile))
    out = {}
    for dma in ("0", "1"):
        monkeypatch.setenv("ER_GEMM_F32_DMA", dma)
        out[dma] = (K_.gemm(a, w, bias, resid), K_.gemm(a, w, bias, None, relu=True), K_.gemm(a, w))
    for x, y in zip(out["0"], out["1"]):
        assert torch.equal(x, y), f"LDS-DMA fp32 GEMM differs from the register-staged one: max {float((x - y).abs().max()):.3e}"
    ref = a.double() @ w.double().T
    close(out["1"][2], ref, 2e-6 + 1e-7 * K, 1e-5, "fp32 LDS-DMA gemm")
    close(out["1"][0], ref + bias.double() + resid.double(), 2e-6 + 1e-7 * K, 1e-5, "fp32 LDS-DMA gemm + bias + resid")


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (4096, 1024, 1024), (514, 3072, 1024), (257, 1280, 5120), (77, 200, 608)])
def test_gemm_f16_input_mfma(M, N, K):
    """fp16-input MFMA GEMM == fp32 math on fp16-rounded operands (asymmetric operands catch a transposed map)."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=70), rnd(N, K, seed=71, scale=0.05)
    bias, resid = rnd(N, seed=72), rnd(M, N, seed=73)
    wh = w.half()
    ref = a.half().double() @ wh.double().T + bias.double()
    c = K_.gemm_f16(a, wh, bias, resid)
    close(c, ref + resid.double(), 1e-5 + 2e-7 * K, 1e-5, "gemm f16")
    c2 = K_.gemm_f16(a, wh, bias, None, relu=True)
    close(c2, torch.relu(ref), 1e-5 + 2e-7 * K, 1e-5, "gemm f16 relu")
    eye = torch.eye(256, device=DEV)
    wq = ((torch.arange(256 * 256, device=DEV, dtype=torch.float32).view(256, 256) % 509) / 8.0).half()
    assert torch.equal(K_.gemm_f16(eye, wq), wq.float().T.contiguous()), "I . W^T must reproduce W^T exactly"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (4096, 3072, 1024), (4096, 1024, 4096), (4096, 64, 1024), (514, 1024, 1024),
                                   (77, 200, 640), (300, 8192, 1024), (129, 65, 128), (3001, 4164, 192), (4096, 8192, 128)])
def test_gemm_hh_lds_dma(M, N, K):
    """fp16 x fp16 GEMM with both operands brought in by LDS-DMA into an XOR-swizzled image (DiT Linears in fp16 mode): must be
    BIT-IDENTICAL to the register-staged fp16 kernel (same fp16-rounded operands, same k order per MFMA, same epilogue), for all
    tile shapes - the three 4-wave ones and the 8-wave 256 x 256 tile the rule picks from 192 tiles up (4096 x 3072, the ragged
    3001 x 4164, 4096 x 8192) -, ragged edges and the XCD-aware tile order; the fp16 copy of the output is the rounded fp32 output."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=84), rnd(N, K, seed=85, scale=0.05)
    bias, resid = rnd(N, seed=86), rnd(M, N, seed=87)
    wh = w.half()
    ref = a.half().double() @ wh.double().T + bias.double()
    c, c16 = K_.gemm_hh(a, wh, bias, resid, return_half=True)
    close(c, ref + resid.double(), 1e-5 + 2e-7 * K, 1e-5, "LDS-DMA fp16 gemm")
    assert torch.equal(c16, c.half()), "fp16 copy of the output"
    if K % 32 == 0:
        assert torch.equal(c, K_.gemm_f16(a, wh, bias, resid)), "must equal the register-staged fp16 kernel bit for bit"
    c2 = K_.gemm_hh(a, wh, bias, None, relu=True)
    close(c2, torch.relu(ref), 1e-5 + 2e-7 * K, 1e-5, "LDS-DMA fp16 gemm relu")
    if M >= 256 and N >= 256 and K >= 256:      # asymmetric operand: catches a transposed fragment map / a wrong swizzle
        eye = torch.eye(256, device=DEV)
        wq = ((torch.arange(256 * 256, device=DEV, dtype=torch.float32).view(256, 256) % 509) / 8.0).half()
        assert torch.equal(K_.gemm_hh(eye, wq), wq.float().T.contiguous()), "I . W^T must reproduce W^T exactly"


@pytest.mark.parametrize("M,F,K", [(4096, 4096, 1024), (300, 128, 192), (2050, 512, 64)])
def test_gemm_hh_geglu_fused_epilogue(M, F, K):
    """The DiT block's feed-forward-in with GEGLU fused into the GEMM epilogue (core/transformer/dit.py FeedForward): fp16 output of
    (x + bx) * gelu_erf(gate + bg) from the permuted-weight product; against float64 torch on the fp16-rounded operands, and
    BIT-EQUAL across the three kernel forms (4 waves 128 x 128 / 64 x 128, 8 waves 256 x 256 - the one the DiT runs at M = 4096)."""
    from edgerunner_amd import kernels as K_
    a, w, bias = rnd(M, K, seed=95), rnd(2 * F, K, seed=96, scale=0.05), rnd(2 * F, seed=97)
    wh = w.half()
    pre = a.half().double() @ wh.double().T + bias.double()
    ref = pre[:, :F] * torch.nn.functional.gelu(pre[:, F:])
    outs = {t: K_.gemm_hh_geglu(a, wh, bias, force_tile=t) for t in ((0, 1, 2, 4) if F % 128 == 0 else (0, 1, 2))}
    close(outs[0].double(), ref, 2e-3, 2e-3, "GEGLU GEMM (fp16 output)")
    for t, o in outs.items():
        assert torch.equal(o, outs[1]), f"tile form {t} differs from the 128 x 128 form"


def test_geglu_epilogue_erf_accuracy():
    """The fused GEGLU epilogue evaluates erf by Abramowitz & Stegun 7.1.26 on the hardware reciprocal / exp2 (k_gemm.h erf_as7126)
    instead of OCML's erff.  Direct check of the DEVICE function: a product whose gate pre-activation is a chosen grid value and whose
    value part is exactly 1 (weights: identity rows for the gate, zero rows + bias 1 for the value), so the fp16 output IS gelu(grid).
    Against float64, for gate values >= -3: never more than one fp16 ulp off the correctly rounded result, and off at all on < 0.5 % of
    the grid (values near a rounding boundary; 0.26 % measured); below -3 (|gelu| < 5e-3): absolute error < 3e-6."""
    from edgerunner_amd import kernels as K_
    F, K = 64, 64
    grid = torch.linspace(-6.0, 6.0, 4096 * 64, device=DEV).half().float().view(4096, 64)      # fp16-exact inputs: the GEMM rounds A to fp16
    w = torch.zeros(2 * F, K, device=DEV)
    w[F:] = torch.eye(F, device=DEV)                       # gate_j = a[:, j]
    bias = torch.cat([torch.ones(F, device=DEV), torch.zeros(F, device=DEV)])
    out = K_.gemm_hh_geglu(grid, w.half(), bias)           # (0 + 1) * gelu(a)
    ref64 = torch.nn.functional.gelu(grid.double())
    want = ref64.half()
    # distance in fp16 ulps: compare the integer encodings of same-sign values
    def enc(t):
        i = t.view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7fff), i)
    d = (enc(out) - enc(want)).abs()
    body = grid >= -3.0            # below -3 the result is 1 + erf ~ 1e-3 and smaller times x: ANY float32 erf (OCML's too) loses its digits to the
                                   # cancellation in 1 + erf there; those outputs are < 5e-3 in magnitude and are held to an ABSOLUTE bound instead
    assert int(d[body].max()) <= 1, f"fused GEGLU is {int(d[body].max())} fp16 ulps off the correctly rounded gelu for a gate value >= -3"
    frac = float((d[body] > 0).float().mean())
    err = (out.double() - ref64).abs()
    print(f"GEGLU epilogue erf: {frac * 100:.3f} % of {int(body.sum())} grid values >= -3 differ from the correctly rounded fp16 gelu (by one ulp); "
          f"max abs error over [-6, 6] {float(err.max()):.2e}, below -3 {float(err[~body].max()):.2e}")
    assert frac < 5e-3
    assert float(err[~body].max()) < 3e-6, "tail: absolute error of x * (1 + erf) / 2 for gate values below -3"
    assert float((err / ref64.abs().clamp_min(1e-3)).max()) < 1.5e-3, "relative error beyond fp16 rounding"


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,rows,heads,K", [(4096, 2048, 16, 1024), (256, 64, 2, 64), (384, 128, 4, 192)])
def test_gemm_hh_qkv_writes_v_transposed(M, rows, heads, K, tile):
    """The q/k/v projection of the DiT self-attention (core/transformer/dit.py:100-126): q and k leave the GEMM as fp16 rows, V as
    V^T per head in the key order flash_attn_hh_kernel reads (replaces a separate transpose pass): every element must be BIT-EQUAL
    to the plain epilogue's fp16 copy, for all tile shapes, and the V columns of the row output must stay untouched."""
    from edgerunner_amd import kernels as K_
    N = 3 * heads * 64
    a, w, bias = rnd(M, K, seed=91), rnd(N, K, seed=92, scale=0.05), rnd(N, seed=93)
    wh = w.half()
    _, c16 = K_.gemm_hh(a, wh, bias, None, return_half=True)
    qk16, vt = K_.gemm_hh_qkv(a, wh, bias, rows, force_tile=tile)
    assert torch.equal(qk16[:, :2 * N // 3], c16[:, :2 * N // 3]), "q / k rows"
    assert float(qk16[:, 2 * N // 3:].abs().max()) == 0.0, "the V third of the row output must not be written"
    v = c16[:, 2 * N // 3:].view(M // rows, rows, heads, 64).permute(0, 2, 3, 1)         # [b][h][d][key]
    key = torch.arange(rows, device=DEV)
    pos = (key & ~15) | (key & 3) | ((key & 8) >> 1) | ((key & 4) << 1)                   # fa_vt_pos
    want = torch.empty_like(vt)
    want[..., pos] = v
    assert torch.equal(vt, want), "V^T per head in MFMA key order"


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 4096), (4096, 1024, 1024), (3001, 4164, 192), (77, 200, 640), (129, 65, 128), (256, 384, 64), (2050, 1536, 6144)])
def test_gemm_hh_streamed_form_is_bit_identical(M, N, K, monkeypatch):
    """The streamed LDS-DMA GEMM (k_gemm_stream.h: loader + matrix waves, five-stage ring, persistent tile stream, eight-wave epilogue;
    the DiT's feed-forward-out by launch_gemm_hh's rule) must give the bits of the 4-wave kernels on every shape - several tiles per
    workgroup (3001 x 4164), fewer tiles than CUs, ragged M / N, one k-tile (K = 64) - with bias + residual + fp16 copy, and with ReLU."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=284), rnd(N, K, seed=285, scale=0.05)
    bias, resid = rnd(N, seed=286), rnd(M, N, seed=287)
    wh = w.half()
    monkeypatch.setenv("ER_GEMM_STREAM", "0")
    c0, h0 = K_.gemm_hh(a, wh, bias, resid, return_half=True)
    r0 = K_.gemm_hh(a, wh, bias, None, relu=True)
    monkeypatch.setenv("ER_GEMM_STREAM", "2")
    c1, h1 = K_.gemm_hh(a, wh, bias, resid, return_half=True)
    r1 = K_.gemm_hh(a, wh, bias, None, relu=True)
    assert torch.equal(c1, c0) and torch.equal(h1, h0) and torch.equal(r1, r0), f"streamed form differs: max {float((c1 - c0).abs().max()):.3e}"
    ref = a.half().double() @ wh.double().T + bias.double() + resid.double()
    close(c1, ref, 1e-5 + 2e-7 * K, 1e-5, "streamed LDS-DMA fp16 gemm")


@pytest.mark.parametrize("M,rows,heads,K", [(4096, 2048, 16, 1024), (384, 128, 4, 192)])
def test_gemm_hh_streamed_form_qkv_v_transposed(M, rows, heads, K, monkeypatch):
    """... and its V^T epilogue (the V third of a fused q/k/v projection leaves as V^T per head in fa_vt_pos order)."""
    from edgerunner_amd import kernels as K_
    N = 3 * heads * 64
    a, w, bias = rnd(M, K, seed=291), rnd(N, K, seed=292, scale=0.05), rnd(N, seed=293)
    wh = w.half()
    monkeypatch.setenv("ER_GEMM_STREAM", "0")
    qk0, vt0 = K_.gemm_hh_qkv(a, wh, bias, rows, force_tile=0)
    monkeypatch.setenv("ER_GEMM_STREAM", "2")
    qk1, vt1 = K_.gemm_hh_qkv(a, wh, bias, rows, force_tile=0)
    assert torch.equal(qk1, qk0) and torch.equal(vt1, vt0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (2050, 4608, 1536), (2050, 1536, 6144), (514, 3072, 1024), (77, 200, 608)])
def test_gemm_f16_split_activations(M, N, K):
    """Split-fp16 GEMM (fast-mode prefill): fp16 weights x (hi + lo)-split fp32 activations must equal the fp32-activation
    product to fp32 round-off - i.e. the arithmetic of the per-token GEMV path - NOT the fp16-rounded-activation product."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=74), rnd(N, K, seed=75, scale=0.05)
    a[0, :8] = torch.tensor([1e-7, -3e-6, 3.0009766, -0.33333334, 12.345678, 1e-3, -1.0, 0.0], device=DEV)   # tiny values, values needing the lo part
    bias, resid = rnd(N, seed=76), rnd(M, N, seed=77)
    wh = w.half()
    ref = a.double() @ wh.double().T + bias.double()
    c = K_.gemm_f16s(a, wh, bias, resid)
    close(c, ref + resid.double(), 2e-6 + 1e-7 * K, 2e-6, "split fp16 gemm")
    rounded = a.half().double() @ wh.double().T + bias.double() + resid.double()
    if K >= 1024:   # the rounded-activation product is measurably further away: the split is doing its job
        assert float((c.double() - ref - resid.double()).abs().max()) < 0.05 * float((rounded - ref - resid.double()).abs().max())
    c2 = K_.gemm_f16s(a, wh, bias, None, relu=True)
    close(c2, torch.relu(ref), 2e-6 + 1e-7 * K, 2e-6, "split fp16 gemm relu")


@pytest.mark.parametrize("M,N,K", [(2050, 4608, 1536), (2050, 1536, 6144), (300, 1536, 1536)])
def test_gemm_f16s_forms_agree(M, N, K, monkeypatch):
    """The fast-mode prefill picks between two forms of the split-fp16 product by row count (er_api.hip linear_hs): split pass + LDS-DMA
    kernel for one or two prefixes, register-staged kernel beyond.  The claim that a row's logits do not depend on how many prefixes
    share the launch rests on the two forms giving the same BITS (round-3 advisor): same hi / lo split, same MFMA order per accumulator."""
    from edgerunner_amd import kernels as K_
    a, w = rnd(M, K, seed=174), rnd(N, K, seed=175, scale=0.05)
    bias, resid = rnd(N, seed=176), rnd(M, N, seed=177)
    wh = w.half()
    out = {}
    for form in ("dma", "reg"):
        monkeypatch.setenv("ER_K_GEMM_F16S_FORM", form)
        out[form] = (K_.gemm_f16s(a, wh, bias, resid), K_.gemm_f16s(a, wh, bias, None, relu=True))
    assert torch.equal(out["dma"][0], out["reg"][0]) and torch.equal(out["dma"][1], out["reg"][1])
    ref = a.double() @ wh.double().T + bias.double() + resid.double()
    close(out["reg"][0], ref, 2e-6 + 1e-7 * K, 2e-6, "register-staged split fp16 gemm")


@pytest.mark.parametrize("B,H,N,M", [(1, 16, 2048, 2048), (2, 16, 2048, 257), (1, 2, 100, 70), (1, 1, 33, 1)])
def test_flash_attn_f16(B, H, N, M):
    from edgerunner_amd import kernels as K_
    q, k, v = rnd(B, N, H * 64, seed=80), rnd(B, M, H * 64, seed=81), rnd(B, M, H * 64, seed=82)
    k[:, M // 2] *= 3.0                                     # a spiky key row forces the running-max rescale
    o = K_.flash_attn_f16(q, k, v, H)
    qh, kh, vh = (t.half().double().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, dim=-1)
    ref = (p @ vh).transpose(1, 2).reshape(B, N, H * 64)
    close(o, ref, 4e-3, 4e-3, "flash attention (fp16 P)")   # P is rounded to fp16 before P.V
    # the LDS-DMA variant works on the same fp16 roundings of q / k / v / p: equal to the register-staged kernel up to the fp16
    # rounding of its OUTPUT (the only difference; it feeds an fp16 Linear)
    o2 = K_.flash_attn_hh(q, k, v, H)
    assert torch.equal(o2, o.half().float()), f"LDS-DMA attention differs: max {float((o2 - o).abs().max()):.3e}"


@pytest.mark.parametrize("B,H,N,M,D,causal", [
    (2, 16, 2050, 2050, 96, True),        # the prefill shape (ragged last query tile and key tile)
    (1, 16, 2048, 4096, 64, False),       # the point encoder's cross-attention
    (2, 3, 130, 1000, 64, False),         # ragged, keys >> queries
    (1, 2, 100, 100, 96, True), (1, 2, 70, 200, 96, True),     # causal with M > N: key j visible iff j <= i + (M - N)
    (1, 1, 33, 1, 64, False), (1, 1, 1, 1, 96, True)])
def test_flash_attn_f32(B, H, N, M, D, causal):
    """Exact-fp32 fused attention (prefill / point encoder) vs an fp64 torch reference of attention()
    (core/transformer/attention.py:47-62: scores / sqrt(D), -inf upper triangle, softmax, P V)."""
    from edgerunner_amd import kernels as K_
    q, k, v = rnd(B, N, H * D, seed=83), rnd(B, M, H * D, seed=84), rnd(B, M, H * D, seed=85)
    k[:, M // 2] *= 3.0                                     # a spiky key row forces the running-max rescale
    o = K_.flash_attn_f32(q, k, v, H, causal=causal)
    qh, kh, vh = (t.double().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    sc = qh @ kh.transpose(-1, -2) / math.sqrt(D)
    if causal:
        i = torch.arange(N, device=DEV)[:, None]
        j = torch.arange(M, device=DEV)[None, :]
        sc = sc.masked_fill(j > i + (M - N), float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ vh).transpose(1, 2).reshape(B, N, H * D)
    close(o, ref, 5e-6, 2e-5, "fp32 flash attention")


@pytest.mark.parametrize("B,H,N,M", [(1, 16, 2050, 2050), (2, 16, 2050, 2050), (3, 4, 300, 300), (1, 2, 100, 100), (1, 2, 70, 200), (1, 1, 1, 1),
                                     (1, 3, 64, 64), (1, 2, 129, 129)])
def test_flash_attn_f32_key_range_split(B, H, N, M, monkeypatch):
    """The causal head_dim-96 prefill attention of a single prefix runs as two workgroups per query tile, each walking one half of
    its key tiles, + flash32_merge_kernel (k_flash_attn_f32.h, KSP; default since round 5, ER_FLASH32_KSPLIT=0 = the unsplit
    kernel).  Same fp64 reference and tolerance as test_flash_attn_f32, and within a few ulp of the unsplit kernel."""
    from edgerunner_amd import kernels as K_
    D = 96
    q, k, v = rnd(B, N, H * D, seed=83), rnd(B, M, H * D, seed=84), rnd(B, M, H * D, seed=85)
    k[:, M // 2] *= 3.0
    monkeypatch.setenv("ER_FLASH32_KSPLIT", "0")
    base = K_.flash_attn_f32(q, k, v, H, causal=True)
    monkeypatch.delenv("ER_FLASH32_KSPLIT")
    o = K_.flash_attn_f32(q, k, v, H, causal=True)
    qh, kh, vh = (t.double().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    sc = qh @ kh.transpose(-1, -2) / math.sqrt(D)
    i = torch.arange(N, device=DEV)[:, None]
    j = torch.arange(M, device=DEV)[None, :]
    sc = sc.masked_fill(j > i + (M - N), float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ vh).transpose(1, 2).reshape(B, N, H * D)
    close(o, ref, 5e-6, 2e-5, "fp32 flash attention, key-range split")
    close(o, base.double(), 2e-6, 2e-6, "split vs unsplit kernel")
    if N >= 2050 and B == 1:
        assert not torch.equal(o, base), "the split did not engage"
    if B > 1:
        assert torch.equal(o, base), "the split is for a single sample only (k_flash_attn_f32.h flash32_ksplit)"


def test_flash_attn_f32_batch_invariance():
    """Exact-mode prefill attention of a prompt does not depend on how many OTHER prompts share its batch: the key-range
    split (different rounding in its merge) is taken by a single sample only, so row 0 of a pair equals row 0 of a batch of three bit
    for bit; the same prompt alone is within the merge's few ulp of them."""
    from edgerunner_amd import kernels as K_
    H, D, N = 16, 96, 2050
    q, k, v = rnd(3, N, H * D, seed=183), rnd(3, N, H * D, seed=184), rnd(3, N, H * D, seed=185)
    o3 = K_.flash_attn_f32(q, k, v, H, causal=True)
    o2 = K_.flash_attn_f32(q[:2].contiguous(), k[:2].contiguous(), v[:2].contiguous(), H, causal=True)
    o1 = K_.flash_attn_f32(q[:1].contiguous(), k[:1].contiguous(), v[:1].contiguous(), H, causal=True)
    assert torch.equal(o2, o3[:2]), "rows of a pair must equal the same rows of a batch of three"
    close(o1[0], o3[0].double(), 2e-6, 2e-6, "single sample (key-range split) vs the same sample in a batch")


# ------------------------------------------------------------------ row ops
@pytest.mark.parametrize("cols", [1536, 1024])
def test_layernorm_rows(cols):
    from edgerunner_amd import kernels as K
    x = rnd(2051, cols, seed=40) * 2 + 1
    w, b = 1 + 0.1 * rnd(cols, seed=41), 0.1 * rnd(cols, seed=42)
    y = K.layernorm(x, w, b)
    close(y, torch.nn.functional.layer_norm(x.double(), (cols,), w.double(), b.double(), 1e-5), 3e-6, 3e-6, "layernorm")


@pytest.mark.parametrize("causal", [False, True])
def test_softmax_rows(causal):
    from edgerunner_amd import kernels as K
    rows, cols, ld = 300, 300 if causal else 1000, 1008
    s = rnd(rows, ld, seed=43) * 3
    ref = s[:, :cols].double().clone()
    if causal:
        ref = ref + torch.triu(torch.full((rows, cols), float("-inf"), device=DEV, dtype=torch.float64), diagonal=1)
    ref = torch.softmax(ref, dim=-1)
    K.softmax_(s, cols, causal)
    close(s[:, :cols], ref, 1e-7, 1e-5, "softmax")
    assert float(s[:, cols:].abs().max()) == 0.0, "padding columns must be zero"


# ------------------------------------------------------------------ sampling head + grammar automaton
def _oracle_allowed(grammar, t, counter, last, V, eos=2):
    """Integer restatement via the oracle's closure semantics (core/models.py:237-268)."""
    if grammar == 0:
        return list(range(V)), counter
    if grammar == 1:
        return list(range(3, V)) + ([eos] if t % 9 == 1 else []), counter
    if t == 0:
        return [5], counter
    if last == 5:
        counter = 9
    elif last in (3, 4):
        counter = 3
    elif last >= 6:
        counter -= 1
    return (list(range(6, V)) if counter > 0 else [3, 4, 5, eos]), counter


@pytest.mark.parametrize("grammar", [0, 1, 2])
def test_sample_head_greedy_matches_oracle_processors(grammar):
    import arae_oracle as O
    from edgerunner_amd import kernels as K
    V, B = 518, 7
    cases = [(0, 0, 0), (1, 5, 0), (2, 100, 9), (10, 300, 1), (10, 3, 0), (11, 4, 2), (19, 6, 1)]
    logits = rnd(B, V, seed=50) * 2
    logits[3, 2] = 50.0          # EOS is the arg max: must win only where the grammar allows it
    logits[4, 2] = 50.0
    for min_new in (0, 64):
        for row, (t, last, counter) in enumerate(cases):
            nt, co, uo = K.sample_head(logits[row:row + 1].contiguous(), 0, grammar, t, [last], [counter], [1],
                                       min_new=min_new)
            allowed, c_ref = _oracle_allowed(grammar, t, counter, last, V)
            s = logits[row].cpu().clone()
            if t < min_new:
                s[2] = -math.inf
            mask = torch.full_like(s, -math.inf)
            mask[allowed] = 0
            ref = int(torch.argmax(s + mask))
            assert nt[0] == ref, (grammar, t, last, counter, min_new, nt, ref)
            assert co[0] == c_ref
            assert uo[0] == (0 if ref == 2 else 1)


def test_sample_head_wide_vocabulary_topk():
    """V = 1030 (the LR tokenizer backend): ids above 1024 must be reachable by argmax and by the top-k sampler."""
    import arae_oracle as O
    from edgerunner_amd import kernels as K
    V = 1030
    logits = rnd(1, V, seed=53)
    logits[0, 1029] = 9.0
    nt, _, _ = K.sample_head(logits, 0, 0, 3, [7], [0], [1])
    assert nt[0] == 1029
    logits[0, 1027] = 9.0
    logits[0, 1025] = 8.5
    seen = set()
    for step in range(150):
        nt, _, _ = K.sample_head(logits, 1, 0, step, [7], [0], [1], top_k=3, seed=99)
        u = K.philox_uniform(99, step, 0)
        ref = O.sample_from_uniform(O.top_k_filter(logits.cpu(), 3)[0], u)
        assert nt[0] in (1025, 1027, 1029)
        if nt[0] != ref:   # only legal when u sits on a CDF boundary (fp32 summation order), as in the V = 518 test below
            filt = O.top_k_filter(logits.cpu(), 3)[0]
            cdf = torch.cumsum(torch.softmax(filt.double(), -1), 0)
            assert min(abs(float(cdf[nt[0]]) - u), abs(float(cdf[ref]) - u)) < 1e-5, (step, nt, ref, u)
        seen.add(nt[0])
    assert seen == {1025, 1027, 1029}


def test_sample_head_finished_rows_emit_pad():
    from edgerunner_amd import kernels as K
    logits = rnd(3, 518, seed=51)
    nt, co, uo = K.sample_head(logits, 0, 2, 12, [7, 0, 9], [2, 0, 1], [1, 0, 1])
    assert nt[1] == 0 and uo[1] == 0 and co[1] == 0
    assert uo[0] == 1 and uo[2] == 1


def test_sample_head_topk_categorical():
    """Device draw == inverse-CDF over the HF top-k/softmax distribution with the same uniform."""
    import arae_oracle as O
    from edgerunner_amd import kernels as K
    V = 518
    logits = rnd(1, V, seed=52) * 3
    logits[0, 100] = logits[0, 101]          # a tie inside the candidate set
    hits = np.zeros(V)
    for step in range(200):
        nt, _, _ = K.sample_head(logits, 1, 0, step, [7], [0], [1], top_k=10, seed=1234)
        u = K.philox_uniform(1234, step, 0)
        filt = O.top_k_filter(logits.cpu(), 10)[0]
        ref = O.sample_from_uniform(filt, u)
        if nt[0] != ref:   # only legal when u sits on a CDF boundary (fp32 summation order)
            p = torch.softmax(filt.double(), -1)
            cdf = torch.cumsum(p, 0)
            assert min(abs(float(cdf[nt[0]]) - u), abs(float(cdf[ref]) - u)) < 1e-5, (step, nt, ref, u)
        assert filt[nt[0]] > -math.inf, "sampled a token outside the top-k set"
        hits[nt[0]] += 1
    assert (hits > 0).sum() >= 3, "sampler is not exploring the candidate set"
    # grammar with 4 legal ids < top_k: every legal id stays a candidate
    seen = set()
    for step in range(1, 120):
        nt, _, _ = K.sample_head(torch.zeros(1, V, device=DEV), 1, 2, step, [3], [0], [1], top_k=10, seed=7)
        # last=3 sets counter=3 -> coordinates only
        assert nt[0] >= 6
    for step in range(1, 200):
        nt, _, _ = K.sample_head(torch.zeros(1, V, device=DEV), 1, 2, step, [6], [1], [1], top_k=10, seed=7)
        seen.add(nt[0])   # counter 1 -> 0: control tokens {3,4,5,2}, uniform logits
    assert seen == {2, 3, 4, 5}
