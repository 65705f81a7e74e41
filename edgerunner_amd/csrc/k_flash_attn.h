// Flash-style attention on the fp16-input matrix cores for the compute-bound front-end (DiT self/cross attention in
// fast mode): softmax(Q K^T / sqrt(D)) V without materialising the scores (the fp32 path writes and re-reads a
// [H, N, M] matrix per sample: 60 % of a DiT forward).  Non-causal, head_dim 64, fp32 in / fp32 out,
// fp16 operands, fp32 accumulation and fp32 online softmax.  Mirrors attention(q, k, v) of
// core/transformer/attention.py:27-62 as used by SelfAttention / CrossAttention (:98-153).
//
// Both products are computed TRANSPOSED so that everything a query row needs stays in one lane column:
//   S^T = K Q^T   (A = K tile rows from LDS, B = Q fragment in registers)  -> lane (q = lane&31) holds 16 keys x 2 blocks
//   O^T = V^T P^T (A = V^T rows from LDS,    B = P^T = the S^T registers repacked to fp16, no data movement)
// so the row max / sum need one xor-32 exchange and the O rescale factor is lane-local.  The MFMA sums over k in
// whatever internal order it likes: A and B are always fed the same (lane-half, element) -> k assignment.
#pragma once
#include "er_common.h"

namespace er {

struct FlashArgs {
    const float* Q; const float* K; const float* V; float* O;
    int N, M;                       // queries, keys
    int ldq, ldk, ldv, ldo;         // row strides (floats)
    long long qs_b, ks_b, vs_b, os_b;   // batch strides
    int head_stride;                // offset between heads inside a row (= D)
    float scale;                    // 1/sqrt(D)
    _Float16* O16;                  // optional: the output rounded to fp16 INSTEAD of O (same strides): it only feeds an fp16 Linear
};

constexpr int FA_D = 64, FA_QW = 32, FA_KT = 64, FA_LD = 72;   // head dim, q rows per wave, keys per tile, LDS row stride (halves)
typedef _Float16 fa_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 fa_h4 __attribute__((ext_vector_type(4)));
typedef float fa_f16v __attribute__((ext_vector_type(16)));

// grid (ceil(N / 128), H, B), 256 threads: wave w owns q rows [128*bx + 32*w, +32)
__global__ __launch_bounds__(ER_WG) void flash_attn_f16_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 Ks[FA_KT * FA_LD];   // [key][d]
    __shared__ __attribute__((aligned(16))) _Float16 Vt[FA_D * FA_LD];    // [d][key]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (ER_NWAVES * FA_QW) + wid * FA_QW;
    const float* Q = a.Q + b * a.qs_b + h * a.head_stride;
    const float* K = a.K + b * a.ks_b + h * a.head_stride;
    const float* V = a.V + b * a.vs_b + h * a.head_stride;
    float* O = a.O16 ? nullptr : a.O + b * a.os_b + h * a.head_stride;
    _Float16* O16 = a.O16 ? a.O16 + b * a.os_b + h * a.head_stride : nullptr;

    // Q fragment (B operand of S^T): q = q0 + li, d = ks*16 + half*8 + e
    fa_h8 qb[4];
    {
        const int q = min(q0 + li, a.N - 1);
        const float* qr = Q + (long long)q * a.ldq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(qr + ks * 16 + half * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(qr + ks * 16 + half * 8 + 4);
            qb[ks] = (fa_h8){(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w,
                             (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
        }
    }
    fa_f16v ot[2];                    // O^T accumulators: rows d = db*32 + (r&3)+8*(r>>2)+4*half, column q = li
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;     // per query (lane column); l_run covers this lane-half's keys only

    const int ntiles = (a.M + FA_KT - 1) / FA_KT;
    for (int t = 0; t < ntiles; ++t) {
        const int kbase = t * FA_KT;
        __syncthreads();              // previous tile fully consumed
        // stage K -> Ks[key][d], V -> Vt[d][key] (fp16); 64x64 floats each = 4 float4 per thread
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + ER_WG * u, key = idx >> 4, c4 = idx & 15;
            const int gk = min(kbase + key, a.M - 1);
            const f32x4 kv = *reinterpret_cast<const f32x4*>(K + (long long)gk * a.ldk + 4 * c4);
            *reinterpret_cast<fa_h4*>(&Ks[key * FA_LD + 4 * c4]) = (fa_h4){(_Float16)kv.x, (_Float16)kv.y, (_Float16)kv.z, (_Float16)kv.w};
            const f32x4 vv = *reinterpret_cast<const f32x4*>(V + (long long)gk * a.ldv + 4 * c4);
            Vt[(4 * c4 + 0) * FA_LD + key] = (_Float16)vv.x;
            Vt[(4 * c4 + 1) * FA_LD + key] = (_Float16)vv.y;
            Vt[(4 * c4 + 2) * FA_LD + key] = (_Float16)vv.z;
            Vt[(4 * c4 + 3) * FA_LD + key] = (_Float16)vv.w;
        }
        __syncthreads();

        // S^T = K Q^T: two 32-key blocks, keys kbase + kb*32 + (r&3)+8*(r>>2)+4*half
        fa_f16v st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const fa_h8 ka = *reinterpret_cast<const fa_h8*>(&Ks[(kb * 32 + li) * FA_LD + ks * 16 + half * 8]);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qb[ks], st[kb], 0, 0, 0);
            }
        }
        // online softmax for query li
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float s = key < a.M ? st[kb][r] * a.scale : -INFINITY;
                st[kb][r] = s;
                mloc = fmaxf(mloc, s);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = expf(m_run - m_new);          // m_run = -inf on the first tile -> 0
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(st[kb][r] - m_new);  // masked keys: exp(-inf) = 0
                st[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;

        // O^T += V^T P^T: per 32-key block two 16-key steps; B = P registers 8*step .. +8 of this lane (keys
        // 16*step + 4*half + {0..3} and + 8 + {0..3}); A = the same keys of row d from Vt
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int stp = 0; stp < 2; ++stp) {
                const fa_h8 pb = {(_Float16)st[kb][8 * stp + 0], (_Float16)st[kb][8 * stp + 1], (_Float16)st[kb][8 * stp + 2],
                                  (_Float16)st[kb][8 * stp + 3], (_Float16)st[kb][8 * stp + 4], (_Float16)st[kb][8 * stp + 5],
                                  (_Float16)st[kb][8 * stp + 6], (_Float16)st[kb][8 * stp + 7]};
                const int kcol = kb * 32 + 16 * stp + 4 * half;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const _Float16* vr = &Vt[(db * 32 + li) * FA_LD + kcol];
                    const fa_h4 v0 = *reinterpret_cast<const fa_h4*>(vr);
                    const fa_h4 v1 = *reinterpret_cast<const fa_h4*>(vr + 8);
                    const fa_h8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, ot[db], 0, 0, 0);
                }
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int q = q0 + li;
    if (q < a.N) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long o = (long long)q * a.ldo + db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = ot[db][r] / l_tot;
                if (O16) O16[o] = (_Float16)v;
                else O[o] = v;
            }
    }
}

inline hipError_t launch_flash_attn_f16(const FlashArgs& a, int H, int B, hipStream_t st) {
    dim3 grid((a.N + ER_NWAVES * FA_QW - 1) / (ER_NWAVES * FA_QW), H, B);
    hipLaunchKernelGGL(flash_attn_f16_kernel, grid, dim3(ER_WG), 0, st, a);
    return hipGetLastError();
}

}  // namespace er
