// Single-token causal self-attention over the incremental KV cache (decode step).
//
// Replaces attention(q, K, V, causal=True) with N == 1
// (core/transformer/attention.py:27-62, called from
// core/transformer/modeling_opt.py:229) on the cache that
// modeling_opt.py:191-192 grows with torch.cat.  Here the cache is preallocated
// [B][H][Lcap][D] (the reference's own [B,16,L,96] layout with a capacity stride)
// and the step's K/V row has already been appended by the QKV GEMV epilogue.
//
// HBM-bound: reads 2*L*D floats per (batch, head).  With only B*16 (b,h) pairs the
// key range is split over S workgroups per pair (flash-decoding): each workgroup
// computes scores for its chunk (8 lanes x 3 float4 cover one 96-float key row: a
// wave-instruction reads 8 complete 128-byte lines), a chunk-local softmax
// (max, exp, sum) and the weighted V sum, and writes {m, l, o[D]}; a second tiny
// kernel merges the S partials exactly as a single softmax would.
#pragma once
#include "er_common.h"

namespace er {

struct AttnDecArgs {
    const float* q;        // [B][hidden]
    const float* kcache;   // [B][H][Lcap][D]
    const float* vcache;
    const int* pos;        // device, per row: index of the newest key (len = pos+1); or
    int fixed_len;         // >0: use this length for every row instead of pos (unit tests)
    const int* len_dev;    // optional per-row lengths (overrides pos when non-null)
    float* part;           // [B][H][S][D+2] = {m, l, o[0..D)}
    float* out;            // combine: [B][hidden]
    int H, l_cap, S, hidden;
    long long kv_bstride;
    float sqrt_d;          // sqrt(D): scores are divided by it, as the reference does
};

__device__ __forceinline__ int attn_len(const AttnDecArgs& a, int b) {
    if (a.len_dev) return a.len_dev[b];
    if (a.fixed_len > 0) return a.fixed_len;
    return a.pos[b] + 1;
}

// grid (S, H, B), 256 threads.  Dynamic LDS: chunk_max floats (scores) + 4*D + 8.
template <int D>
__global__ __launch_bounds__(ER_WG) void attn_decode_f32_kernel(AttnDecArgs a, int chunk_max) {
    constexpr int NV = D / 32;   // float4 per lane per key (8 lanes per key)
    static_assert(D % 32 == 0, "head_dim must be a multiple of 32");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                         // [chunk_max]
    float* ored = smem + chunk_max;           // [4][D]
    float* red = ored + ER_NWAVES * D;        // [8]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & 7, g = lane >> 3;
    const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = attn_len(a, b);
    int chunk = (len + a.S - 1) / a.S;
    chunk = (chunk + 31) & ~31;
    const int k0 = s * chunk;
    const int k1 = min(len, k0 + chunk);
    float* pout = a.part + (((long long)b * a.H + h) * a.S + s) * (D + 2);
    if (k0 >= len) {                          // empty split: neutral element of the merge
        if (tid < D + 2) pout[tid] = (tid == 0) ? -INFINITY : 0.0f;
        return;
    }
    f32x4 qv[NV];
    const f32x4* qp = reinterpret_cast<const f32x4*>(a.q + (long long)b * a.hidden + h * D);
#pragma unroll
    for (int j = 0; j < NV; ++j) qv[j] = qp[j * 8 + p];
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;

    // ---- pass 1: scores of this chunk
    const float* kb = a.kcache + head_off;
    float mloc = -INFINITY;
#pragma unroll 2
    for (int kk = k0 + wid * 8 + g; kk < k1; kk += 32) {
        const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)kk * D);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) acc = dot4(qv[j], kr[j * 8 + p], acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        const float sv = acc / a.sqrt_d;
        if (p == 0) sc[kk - k0] = sv;
        mloc = fmaxf(mloc, sv);
    }
    const float m = block_max(mloc, red);     // barriers inside also publish sc[]
    const int n = k1 - k0;
    float lloc = 0.f;
    for (int i = tid; i < n; i += ER_WG) {
        const float e = expf(sc[i] - m);
        sc[i] = e;
        lloc += e;
    }
    const float l = block_sum(lloc, red);

    // ---- pass 2: o = sum_k p_k V_k
    const float* vb = a.vcache + head_off;
    f32x4 o[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kk = k0 + wid * 8 + g; kk < k1; kk += 32) {
        const float pw = sc[kk - k0];
        const f32x4* vr = reinterpret_cast<const f32x4*>(vb + (long long)kk * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f32x4 v = vr[j * 8 + p];
            o[j].x = fmaf(pw, v.x, o[j].x);
            o[j].y = fmaf(pw, v.y, o[j].y);
            o[j].z = fmaf(pw, v.z, o[j].z);
            o[j].w = fmaf(pw, v.w, o[j].w);
        }
    }
    // sum over the 8 key groups of the wave (lanes with equal p), then over the 4 waves
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            o[j].x += __shfl_xor(o[j].x, off, 64);
            o[j].y += __shfl_xor(o[j].y, off, 64);
            o[j].z += __shfl_xor(o[j].z, off, 64);
            o[j].w += __shfl_xor(o[j].w, off, 64);
        }
    }
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) reinterpret_cast<f32x4*>(ored + wid * D)[j * 8 + p] = o[j];
    }
    __syncthreads();
    if (tid < D) pout[2 + tid] = (ored[tid] + ored[D + tid]) + (ored[2 * D + tid] + ored[3 * D + tid]);
    if (tid == 0) { pout[0] = m; pout[1] = l; }
}

// grid (H, B), 128 threads (>= D): merge the S partial softmaxes of one (b, h).
template <int D>
__global__ __launch_bounds__(128) void attn_combine_f32_kernel(AttnDecArgs a) {
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* pb = a.part + ((long long)b * a.H + h) * a.S * (D + 2);
    float M = -INFINITY;
    for (int s = 0; s < a.S; ++s) M = fmaxf(M, pb[s * (D + 2)]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < a.S; ++s) {
        const float ms = pb[s * (D + 2)];
        if (ms == -INFINITY) continue;        // empty split
        const float w = expf(ms - M);
        l = fmaf(pb[s * (D + 2) + 1], w, l);
        if (tid < D) o = fmaf(pb[s * (D + 2) + 2 + tid], w, o);
    }
    if (tid < D) a.out[(long long)b * a.hidden + h * D + tid] = o / l;
}

inline int attn_chunk_max(int l_cap, int S) { return (((l_cap + S - 1) / S) + 31) & ~31; }

template <int D>
inline hipError_t launch_attn_decode(const AttnDecArgs& a, int B, hipStream_t st) {
    const int chunk_max = attn_chunk_max(a.l_cap, a.S);
    const size_t lds = (size_t)(chunk_max + ER_NWAVES * D + 8) * sizeof(float);
    hipLaunchKernelGGL((attn_decode_f32_kernel<D>), dim3(a.S, a.H, B), dim3(ER_WG), lds, st, a, chunk_max);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_combine_f32_kernel<D>), dim3(a.H, B), dim3(128), 0, st, a);
    return hipGetLastError();
}

}  // namespace er
