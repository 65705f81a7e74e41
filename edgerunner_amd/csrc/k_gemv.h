// Weight-streaming GEMV for the decode step (batch 1..4 rows per pass), fp32.
//
// Replaces the per-token nn.Linear calls of the reference decoder
// (core/transformer/modeling_opt.py:185,189-190 q/k/v, :232 out_proj, :281 fc1,
// :284 fc2, :497 lm_head) and fuses what surrounds them:
//   prologue  LayerNorm of the previous sub-block's residual sum (post-LN decoder,
//             modeling_opt.py:273-274 / :287-288), or token+position embedding
//             (modeling_opt.py:340-342, 355-357) for layer 0;
//   epilogue  +bias, ReLU (:282), +residual (:273 / :287), or the KV-cache append
//             that replaces torch.cat (:191-192).
//
// HBM-bound: every weight element is read exactly once per token.  Layout: W is the
// nn.Linear weight [N][K] row-major, so one output row is a contiguous K-vector.
// A wave owns RW rows; lane l reads float4 #(j*64+l) of the row slice (1 KiB per
// wave-instruction, fully coalesced), keeps J*RW loads in flight, FMAs against the
// LayerNorm'd input held in registers (staged once per workgroup through LDS) and
// finishes with a 64-lane shuffle reduction.  For K = 4 slices (fc2) the 4 waves of
// a workgroup split K and combine through LDS.
#pragma once
#include "er_common.h"

namespace er {

enum { PRO_NONE = 0, PRO_LN = 1, PRO_EMBED = 2 };
enum { EPI_STORE = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_QKV = 3 };

struct GemvArgs {
    const float* W;        // [N][K]
    const float* bias;     // [N] or nullptr
    int N;
    // prologue
    const float* xin;      // PRO_NONE: input [NB][K]; PRO_LN: pre-LN vector [NB][K]
    const float* ln_w;     // PRO_LN
    const float* ln_b;
    float eps;
    float* hout;           // PRO_LN / PRO_EMBED: block 0 stores the prologue result here ([NB][K]); may be null
    const float* embd;     // PRO_EMBED: token table [V][K]
    const float* posemb;   //            position table [P][K]
    const int* tok;        //            current token per row  (device)
    const int* pos;        // PRO_EMBED / EPI_QKV: position of the token being fed, per row (device)
    // epilogue
    float* out;            // [NB][N]
    const float* resid;    // EPI_RESID: [NB][N]
    float* q;              // EPI_QKV: [NB][hidden]
    float* kcache;         //          [B][H][Lcap][D] (this layer)
    float* vcache;
    int hidden, head_dim, l_cap;
    long long kv_bstride;  // H*Lcap*D
};

// bias / residual / cache position are fetched at kernel entry (EpiPre) so that the epilogue after the
// reduction is pure arithmetic + one store instead of a chain of dependent L2 round trips.
struct EpiPre { float bias; float resid; int pos; };

template <int EPI>
__device__ __forceinline__ EpiPre gemv_epi_prefetch(const GemvArgs& a, int n, int b) {
    EpiPre e{0.f, 0.f, 0};
    n = min(n, a.N - 1);
    if (a.bias) e.bias = a.bias[n];
    if (EPI == EPI_RESID) e.resid = a.resid[(long long)b * a.N + n];
    if (EPI == EPI_QKV) e.pos = a.pos[b];
    return e;
}

template <int EPI>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, int n, int b, float v, const EpiPre& e) {
    v += e.bias;
    if (EPI == EPI_STORE) {
        a.out[(long long)b * a.N + n] = v;
    } else if (EPI == EPI_RELU) {
        a.out[(long long)b * a.N + n] = fmaxf(v, 0.0f);
    } else if (EPI == EPI_RESID) {
        a.out[(long long)b * a.N + n] = v + e.resid;
    } else {  // EPI_QKV: rows [0,hidden) = q, [hidden,2h) = k, [2h,3h) = v
        const int which = n / a.hidden;
        const int c = n - which * a.hidden;
        if (which == 0) {
            a.q[(long long)b * a.hidden + c] = v;
        } else {
            const int h = c / a.head_dim, d = c - h * a.head_dim;
            float* cache = (which == 1) ? a.kcache : a.vcache;
            cache[(long long)b * a.kv_bstride + ((long long)h * a.l_cap + e.pos) * a.head_dim + d] = v;
        }
    }
}

// K = KS * J * 256.  Dynamic LDS: NB*K floats (input) + 64 floats scratch.
template <int J, int KS, int NB, int RW, int PRO, int EPI>
__global__ __launch_bounds__(ER_WG) void gemv_f32_kernel(GemvArgs a) {
    constexpr int K = KS * J * 256;
    constexpr int PT = K / ER_WG;  // elements per thread in the prologue
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;              // [NB][K]
    float* red = smem + NB * K;    // 64 floats
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int slice = (KS == 1) ? 0 : wid;                 // K-slice this wave reduces
    const int row0 = (KS == 1) ? (blockIdx.x * ER_NWAVES + wid) * RW : blockIdx.x * RW;

    // ---------------- issue this wave's weight loads first: they do not depend on the prologue, so the
    // HBM round trip (~1-2 us) overlaps the LayerNorm reductions instead of following them
    f32x4 w[RW][J];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = min(row0 + r, a.N - 1);            // clamp: out-of-range rows are loaded but never stored
        const f32x4* wr = reinterpret_cast<const f32x4*>(a.W + (long long)row * K + slice * (J * 256));
#pragma unroll
        for (int j = 0; j < J; ++j) w[r][j] = __builtin_nontemporal_load(wr + j * 64 + lane);
    }

    // epilogue operands of the (row, batch) pairs this thread will finish
    EpiPre pre[RW][NB];
    if (KS == 1) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) pre[r][b] = gemv_epi_prefetch<EPI>(a, row0 + r, b);
    } else {
        const int t = min(tid, RW * NB - 1);
        pre[0][0] = gemv_epi_prefetch<EPI>(a, row0 + t / NB, t % NB);
    }

    // ---------------- prologue: build the input vector(s) in LDS
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v[PT];
        if (PRO == PRO_EMBED) {
            const float* e = a.embd + (long long)a.tok[b] * K;
            const float* p = a.posemb + (long long)a.pos[b] * K;
#pragma unroll
            for (int i = 0; i < PT; ++i) v[i] = e[tid + i * ER_WG] + p[tid + i * ER_WG];
        } else {
            const float* x = a.xin + (long long)b * K;
#pragma unroll
            for (int i = 0; i < PT; ++i) v[i] = x[tid + i * ER_WG];
        }
        if (PRO == PRO_LN) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < PT; ++i) s += v[i];
            const float mean = block_sum(s, red) / (float)K;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < PT; ++i) { const float d = v[i] - mean; s2 = fmaf(d, d, s2); }
            const float var = block_sum(s2, red) / (float)K;
            const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int c = tid + i * ER_WG;
                v[i] = (v[i] - mean) * rstd * a.ln_w[c] + a.ln_b[c];
            }
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) xs[b * K + tid + i * ER_WG] = v[i];
        if (PRO != PRO_NONE && a.hout != nullptr && blockIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < PT; ++i) a.hout[(long long)b * K + tid + i * ER_WG] = v[i];
        }
    }
    __syncthreads();

    // ---------------- main: dot the (already in flight) weight rows with the input
    f32x4 xr[NB][J];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < J; ++j)
            xr[b][j] = reinterpret_cast<const f32x4*>(xs + b * K + slice * (J * 256))[j * 64 + lane];

    float acc[RW][NB];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) s = dot4(w[r][j], xr[b][j], s);
            acc[r][b] = wave_sum(s);
        }

    if (KS == 1) {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (row0 + r < a.N) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) gemv_epilogue<EPI>(a, row0 + r, b, acc[r][b], pre[r][b]);
                }
        }
    } else {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int b = 0; b < NB; ++b) red[(r * NB + b) * KS + wid] = acc[r][b];
        }
        __syncthreads();
        if (tid < RW * NB) {
            const int r = tid / NB, b = tid - r * NB;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k) s += red[tid * KS + k];
            if (row0 + r < a.N) gemv_epilogue<EPI>(a, row0 + r, b, s, pre[0][0]);
        }
    }
}

template <int J, int KS, int NB, int RW, int PRO, int EPI>
inline hipError_t launch_gemv(const GemvArgs& a, hipStream_t st) {
    static_assert(KS == 1 || KS == ER_NWAVES, "K is reduced by one wave or by all four");
    constexpr int K = KS * J * 256;
    const int rows_per_block = (KS == 1) ? ER_NWAVES * RW : RW;
    const int grid = (a.N + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)(NB * K + 64) * sizeof(float);
    hipLaunchKernelGGL((gemv_f32_kernel<J, KS, NB, RW, PRO, EPI>), dim3(grid), dim3(ER_WG), lds, st, a);
    return hipGetLastError();
}

}  // namespace er
