// The fast-mode prefill attention on the fp16 matrix cores with fp32-grade operands: the default for every batch size (round 3,
// encode + prefill per sample against the fp32-matrix-core kernel: 21.9 -> 19.3 ms at B = 1, 15.5 -> 12.8 ms at B = 8, 14.65 -> 12.2 ms
// at B = 32; end-to-end fast-mode logits move by 2.9e-6; profiles/r03_f16s_prefix_attention.log,
// profiles/r03_dit_vt_epilogue_prefill_prefetch.log).
//
// In fast mode the K / V rows the prefix attention reads are already fp16 values (kv_scatter_half_kernel rounds them to
// the cache dtype, er_api.hip), but the fused attention still runs on the fp32 matrix cores (flash_attn_f32_kernel: 7.4 of
// the 24.4 ms of a fast-mode prefill at B = 1, 29 % MFMA-busy at 1/16 of the fp16 rate).  Here both products run on
// v_mfma_f32_32x32x16_f16 with the fp32 operand split into two fp16 numbers, exactly like the split-fp16 GEMM (k_gemm.h):
//   S^T = K (Q_hi + Q_lo)^T        Q scaled by 2^4 before the split (keeps the lo part out of the fp16 subnormals), the
//                                  factor folded into the sqrt(D) division (a power of two: same rounding)
//   O^T = V^T (P_hi + P_lo)^T      P scaled by 2^10 (P <= 1), the factor folded into the final 1/l normalisation
// fp16 x fp16 products are exact in the fp32 accumulator, so the result is the fp16-K/V x fp32-Q/P product to fp32 round-off:
// the arithmetic of core/transformer/attention.py:27-62 on fp16-rounded storage, which is what the fast-mode parity tests
// check (oracle on fp16-rounded storage).  Same transposed formulation, fragment maps and online softmax as
// k_flash_attn.h / k_flash_attn_f32.h (a query's statistics stay in one lane column); causal tiles beyond a wave's last key
// are skipped.
#pragma once
#include "er_common.h"
#include "k_flash_attn.h"
#include "k_flash_attn_f32.h"

namespace er {

constexpr float FAS_QSCALE = 16.0f, FAS_PSCALE = 1024.0f;

// grid (ceil(N / (32 NWV)), H, B), 64 NWV threads; arguments as flash_attn_f32_kernel (Flash32Args), K / V must hold
// fp16-representable values (they are converted, not rounded, on their way into LDS).
// The K / V rows of tile t + 1 are loaded into registers while tile t is multiplied (96 more registers at NWV = 2).  The kernel holds
// more than 256 registers either way (q hi / lo 48, O^T 48, S^T 32, one tile's staging set: one wave per SIMD), so nothing but this
// prefetch hides the load -> convert -> LDS latency of a tile: ONE 2050-token prefix is 33 x 16 = 528 two-wave workgroups whose longest
// walks 33 tiles (without it: 22.6 instead of 19.3 ms of encode + prefill at B = 1, 12.98 instead of 12.79 ms per sample at B = 8).
template <int D, bool CAUSAL, int NWV>
__global__ __launch_bounds__(64 * NWV) void flash_attn_f16s_kernel(Flash32Args a) {
    constexpr int THREADS = 64 * NWV;
    constexpr int KT = 64, KLD = D + 8, VLD = KT + 8;     // keys per tile; LDS row strides in halves (KLD / 8 and VLD / 8 odd)
    constexpr int KS = D / 16, NDB = D / 32, F4 = D / 4;
    static_assert(D % 32 == 0 && ((KLD / 8) & 1) == 1 && ((VLD / 8) & 1) == 1, "head_dim 64 or 96");
    __shared__ __attribute__((aligned(16))) _Float16 Ks[KT * KLD];   // [key][d]
    __shared__ __attribute__((aligned(16))) _Float16 Vt[D * VLD];    // [d][key]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qt = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int q0 = qt * (NWV * 32) + wid * 32;
    const float* Q = a.Q + b * a.qs_b + h * a.qs_h;
    const float* K = a.K + b * a.ks_b + h * a.ks_h;
    const float* V = a.V + b * a.vs_b + h * a.vs_h;
    float* O = a.O + b * a.os_b + h * a.os_h;

    // Q fragment (B operand of S^T): query q0 + li, dims ks*16 + half*8 + e, as hi + lo fp16 parts of 16 q
    fa_h8 qh[KS], ql[KS];
    const int qi = q0 + li;
    {
        const float* qr = Q + (long long)min(qi, a.N - 1) * a.ldq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(qr + ks * 16 + half * 8);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(qr + ks * 16 + half * 8 + 4);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = x[e] * FAS_QSCALE;
                const _Float16 hi = (_Float16)s;
                qh[ks][e] = hi;
                ql[ks][e] = (_Float16)(s - (float)hi);
            }
        }
    }
    fa_f16v ot[NDB];                 // O^T: rows d = db*32 + (r&3) + 8*(r>>2) + 4*half, column q = li (scaled by FAS_PSCALE)
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;     // l_run covers this lane half's keys only

    const int wg_last_q = min(a.N - 1, qt * (NWV * 32) + NWV * 32 - 1);
    const int kmax = CAUSAL ? min(a.M, wg_last_q + a.causal_off + 1) : a.M;
    const int ntiles = (kmax + KT - 1) / KT;
    const int wave_last_key = CAUSAL ? (q0 + 31 + a.causal_off) : (a.M - 1);
    const float sdiv = a.sqrt_d * FAS_QSCALE;
    const float inv_sdiv = 1.0f / sdiv;

    constexpr int NST = (KT * F4) / THREADS;
    static_assert((KT * F4) % THREADS == 0, "staging loop shape");
    f32x4 kpre[NST], vpre[NST];
    auto fetch = [&](int kbase, int u, f32x4& kv, f32x4& vv) {
        const int idx = tid + THREADS * u, key = idx / F4, c4 = idx - key * F4;
        const int gk = min(kbase + key, a.M - 1);
        kv = *reinterpret_cast<const f32x4*>(K + (long long)gk * a.ldk + 4 * c4);
        vv = *reinterpret_cast<const f32x4*>(V + (long long)gk * a.ldv + 4 * c4);
    };
    auto stage = [&](int u, const f32x4& kv, const f32x4& vv) {
        const int idx = tid + THREADS * u, key = idx / F4, c4 = idx - key * F4;
        *reinterpret_cast<fa_h4*>(&Ks[key * KLD + 4 * c4]) = (fa_h4){(_Float16)kv.x, (_Float16)kv.y, (_Float16)kv.z, (_Float16)kv.w};
        Vt[(4 * c4 + 0) * VLD + key] = (_Float16)vv.x;
        Vt[(4 * c4 + 1) * VLD + key] = (_Float16)vv.y;
        Vt[(4 * c4 + 2) * VLD + key] = (_Float16)vv.z;
        Vt[(4 * c4 + 3) * VLD + key] = (_Float16)vv.w;
    };
#pragma unroll
    for (int u = 0; u < NST; ++u) fetch(0, u, kpre[u], vpre[u]);
    for (int t = 0; t < ntiles; ++t) {
        const int kbase = t * KT;
        __syncthreads();                     // previous tile fully consumed
        // stage K -> Ks[key][d], V -> Vt[d][key] as fp16 (exact: the values are fp16 already)
#pragma unroll
        for (int u = 0; u < NST; ++u) stage(u, kpre[u], vpre[u]);
        __syncthreads();
        if (t + 1 < ntiles) {
#pragma unroll
            for (int u = 0; u < NST; ++u) fetch(kbase + KT, u, kpre[u], vpre[u]);
        }
        if (kbase > wave_last_key) continue;   // wave-uniform: the barriers above are still hit by every wave

        // S^T = K Q^T: two 32-key blocks; lane (li, half) holds keys kbase + kb*32 + (r&3) + 8*(r>>2) + 4*half
        fa_f16v st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const fa_h8 ka = *reinterpret_cast<const fa_h8*>(&Ks[(kb * 32 + li) * KLD + ks * 16 + half * 8]);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, ql[ks], st[kb], 0, 0, 0);     // small parts first
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qh[ks], st[kb], 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                bool ok = key < a.M;
                if (CAUSAL) ok = ok && key <= qi + a.causal_off;
                // the correctly rounded quotient in three instructions (see k_flash_attn_f32.h; sdiv is sqrt(D) times a power of two,
                // so the exhaustive check of scripts/probes/div_const_probe.hip carries over)
                const float q0 = st[kb][r] * inv_sdiv;
                const float s = ok ? fmaf(fmaf(-q0, sdiv, st[kb][r]), inv_sdiv, q0) : -INFINITY;
                st[kb][r] = s;
                mloc = fmaxf(mloc, s);
            }
        mloc = xor_max<32>(mloc);
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = (m_new == -INFINITY) ? 1.0f : expf(m_run - m_new);   // m_run = -inf -> exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (st[kb][r] == -INFINITY) ? 0.f : expf(st[kb][r] - m_new);
                st[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;

        // O^T += V^T P^T: per 32-key block two 16-key steps; B = P registers 8*step .. +8 of this lane (keys
        // 16*step + 4*half + {0..3} and + 8 + {0..3}) as hi + lo parts of 1024 p; A = the same keys of row d from Vt
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int stp = 0; stp < 2; ++stp) {
                fa_h8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float s = st[kb][8 * stp + e] * FAS_PSCALE;
                    const _Float16 hi = (_Float16)s;
                    ph[e] = hi;
                    pl[e] = (_Float16)(s - (float)hi);
                }
                const int kcol = kb * 32 + 16 * stp + 4 * half;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const _Float16* vr = &Vt[(db * 32 + li) * VLD + kcol];
                    const fa_h4 v0 = *reinterpret_cast<const fa_h4*>(vr);
                    const fa_h4 v1 = *reinterpret_cast<const fa_h4*>(vr + 8);
                    const fa_h8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pl, ot[db], 0, 0, 0);
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, ph, ot[db], 0, 0, 0);
                }
            }
    }
    const float l_tot = (xor_sum<32>(l_run)) * FAS_PSCALE;
    if (qi < a.N) {
        float* orow = O + (long long)qi * a.ldo;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) orow[db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = ot[db][r] / l_tot;
    }
}

inline hipError_t launch_flash_attn_f16s(const Flash32Args& a, int D, bool causal, int H, int B, hipStream_t st) {
    const long long wg4 = (long long)((a.N + 127) / 128) * H * B;
    const int nwv = wg4 < 768 ? 2 : 4;          // same rule as launch_flash_attn_f32
    dim3 grid((a.N + nwv * 32 - 1) / (nwv * 32), H, B), blk(64 * nwv);
    if (D != 96) return hipErrorInvalidValue;
    if (nwv == 4) {
        if (causal) hipLaunchKernelGGL((flash_attn_f16s_kernel<96, true, 4>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((flash_attn_f16s_kernel<96, false, 4>), grid, blk, 0, st, a);
    } else {
        if (causal) hipLaunchKernelGGL((flash_attn_f16s_kernel<96, true, 2>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((flash_attn_f16s_kernel<96, false, 2>), grid, blk, 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace er
