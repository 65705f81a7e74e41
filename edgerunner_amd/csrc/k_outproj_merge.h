// Decode step, single row (ER_DECODE_V=3): the merge of the split-attention partials FUSED into the out_proj GEMV.
//
// Replaces, per layer and token, attn_combine2_kernel + the out_proj gemv_kernel (two dependent launches, 3.0 + 3.8 us when
// replayed alone, 4.8 + 4.7 us in situ) with one launch:
//   o_h   = sum_s exp(m_s - M) o_{h,s} / sum_s exp(m_s - M) l_s      the softmax merge of attention.py:44-62 cut into chunks
//   y     = Wo o + bo + resid                                        modeling_opt.py:232 out_proj, :273 residual add
// Every workgroup needs the whole merged vector o (1536 values), so every workgroup redoes the merge: with the balanced
// attention kernel (k_attn_decode.h, version 3) a head always has NCH = 16 partials, i.e. 16 x 16 x 384 B = 98 KB of
// L2-resident partials per workgroup, read as 16-byte columns in ONE round trip (16 loads per lane).
//
// Geometry: 256 workgroups (one per CU) x 8 waves.  Merge: wave w owns heads 2w and 2w+1, one per 32-lane half; lane
// (half, c) with c < 24 owns float4 column c of that head and lane (half, s) with s < NCH also fetches {m_s, l_s}, so the
// maximum / weights / denominators are 5-step half-wave butterflies and a weight reaches its column lanes by one bpermute.
// GEMV: waves 0..5 own one output row each (the same per-lane fmaf chain over float4 #(j*64+lane) and the same 64-lane
// butterfly as gemv_kernel); waves 6 and 7 re-load the rows of waves 4 and 5 so that every wave has the same number of loads
// in flight (a wave-dependent load count would make the compiler wait for the WEIGHTS before the merge arithmetic).
// Loads are issued in the order their consumers run: bias / residual, {m, l}, the partial columns, then the weight stream.
#pragma once
#include "er_common.h"
#include "k_gemv.h"

namespace er {

struct OutMergeArgs {
    const void* W;           // out_proj weight [N][N] in the kernel's weight type
    const float* bias;       // [N]
    const float* resid;      // [N]
    float* out;              // [N]
    const float* part_o;     // [H][NCH][D]
    const float* part_ml;    // [H][NCH][2]
    int N;                   // hidden = H * D = 1536
};

constexpr int OM_WAVES = 8, OM_ROWS = 6, OM_THREADS = OM_WAVES * 64;

// All seven arguments are scalars (13 dwords): preloaded into SGPRs at wave launch (k_gemv.h gemv_kernel), so the partial / weight
// loads do not wait for an s_load of the argument block.
template <typename WT, int D, int NCH>
__global__ __launch_bounds__(OM_THREADS) void outproj_merge_kernel(const void* pW, const float* pbias, const float* presid, const float* ppart_o,
                                                                   const float* ppart_ml, float* pout, int pN) {
    OutMergeArgs a;
    a.W = pW; a.bias = pbias; a.resid = presid; a.part_o = ppart_o; a.part_ml = ppart_ml; a.out = pout; a.N = pN;
    constexpr int EPL = WTraits<WT>::EPL, XV = EPL / 4, K = 1536, J = K / (64 * EPL);
    constexpr int C4 = D / 4;                 // float4 columns per head (24)
    static_assert(D == 96 && NCH <= 32 && C4 <= 32 && K == 16 * D, "two 96-wide heads per wave, 16 heads");
    __shared__ __attribute__((aligned(16))) float xs[K];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    const int head = 2 * wid + half;
    const int row = blockIdx.x * OM_ROWS + (wid < OM_ROWS ? wid : wid - 2);

    // ---- loads, in consumer order
    const float bias = a.bias ? a.bias[row] : 0.f;
    const float resid = a.resid[row];
    const int sl = min(li, NCH - 1);
    const float2 ml = *reinterpret_cast<const float2*>(a.part_ml + ((long long)head * NCH + sl) * 2);
    const int c4 = min(li, C4 - 1);           // lanes 24..31 of a half re-read column 23 (never stored)
    const f32x4* pcol = reinterpret_cast<const f32x4*>(a.part_o + (long long)head * NCH * D) + c4;
    f32x4 pv[NCH];
    f32x4 w[J];
    const f32x4* wr = reinterpret_cast<const f32x4*>(reinterpret_cast<const WT*>(a.W) + (long long)row * K);
#pragma unroll
    for (int s = 0; s < NCH; ++s) pv[s] = pcol[s * C4];
    // (the weight row issued FIRST - its HBM round trip started before the address unit has worked through the 8 x 17 partial loads -
    // measured slower, 4.80 vs 4.55 us fp32 and 4.17 vs 4.07 us fp16: the merge arithmetic then waits for it, loads return in order;
    // profiles/r05_ab_om_wfirst.log)
#pragma unroll
    for (int j = 0; j < J; ++j) w[j] = __builtin_nontemporal_load(wr + j * 64 + lane);
    __builtin_amdgcn_sched_barrier(0);

    // ---- merge of this half-wave's head
    const bool act = li < NCH;
    float m = act ? ml.x : -INFINITY;
    m = xor_max<16>(m); m = xor_max<8>(m); m = xor_max<4>(m); m = xor_max<2>(m); m = xor_max<1>(m);     // stays inside the 32-lane half
    const float wgt = (act && ml.x != -INFINITY) ? expf(ml.x - m) : 0.f;           // m is finite: chunk 0 always holds a key
    float l = act ? ml.y * wgt : 0.f;
    l = xor_sum<16>(l); l = xor_sum<8>(l); l = xor_sum<4>(l); l = xor_sum<2>(l); l = xor_sum<1>(l);
    f32x4 oe = {0.f, 0.f, 0.f, 0.f}, oo = {0.f, 0.f, 0.f, 0.f};                    // even / odd partials: two independent chains
#pragma unroll
    for (int s = 0; s < NCH; s += 2) {
        const float w0 = __shfl(wgt, (lane & 32) | s, 64);
        oe.x = fmaf(pv[s].x, w0, oe.x); oe.y = fmaf(pv[s].y, w0, oe.y); oe.z = fmaf(pv[s].z, w0, oe.z); oe.w = fmaf(pv[s].w, w0, oe.w);
        if (s + 1 < NCH) {
            const float w1 = __shfl(wgt, (lane & 32) | (s + 1), 64);
            oo.x = fmaf(pv[s + 1].x, w1, oo.x); oo.y = fmaf(pv[s + 1].y, w1, oo.y);
            oo.z = fmaf(pv[s + 1].z, w1, oo.z); oo.w = fmaf(pv[s + 1].w, w1, oo.w);
        }
    }
    if (li < C4) {
        f32x4 r;
        r.x = (oe.x + oo.x) / l; r.y = (oe.y + oo.y) / l; r.z = (oe.z + oo.z) / l; r.w = (oe.w + oo.w) / l;
        reinterpret_cast<f32x4*>(xs + head * D)[li] = r;
    }
    __syncthreads();

    // ---- out_proj row: dot the (already in flight) weight row with the merged vector
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        f32x4 x[XV];
#pragma unroll
        for (int u = 0; u < XV; ++u) x[u] = reinterpret_cast<const f32x4*>(xs)[(j * 64 + lane) * XV + u];
        s = dot_w<WT>(w[j], x, s);
    }
    s = wave_sum(s);
    if (lane == 0 && wid < OM_ROWS) a.out[row] = (s + bias) + resid;
}

template <typename WT, int D>
inline hipError_t launch_outproj_merge(const OutMergeArgs& a, int nch, hipStream_t st) {
    const int grid = a.N / OM_ROWS;            // 1536 / 6 = 256: one workgroup per CU
    if (nch == 16) hipLaunchKernelGGL((outproj_merge_kernel<WT, D, 16>), dim3(grid), dim3(OM_THREADS), 0, st, a.W, a.bias, a.resid, a.part_o, a.part_ml,
                                      a.out, a.N);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace er
