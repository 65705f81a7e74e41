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
// key range is cut into fixed chunks of 32*STEPS keys, one workgroup per chunk
// (flash-decoding).  8 lanes x NV float4 cover one key row, so a wave-instruction
// reads 8 complete 128-byte lines.  Every wave issues ALL of its K and V loads
// (2*STEPS*NV float4 per lane) before touching any of them: one HBM round trip per
// workgroup instead of one per loop iteration.  Scores, the chunk-local softmax and
// the weighted V sum then run out of registers; the workgroup writes {m, l, o[D]} and
// a second tiny kernel merges the partials of a head exactly as one softmax would.
#pragma once
#include "er_common.h"

namespace er {

struct AttnDecArgs {
    const float* q;        // [B][hidden]
    const float* kcache;   // [B][H][Lcap][D]
    const float* vcache;
    const int* pos;        // device, per row: index of the newest key (len = pos+1); or
    int fixed_len;         // >0: use this length for every row instead of pos
    const int* len_dev;    // optional per-row lengths (overrides pos when non-null)
    float* part;           // [B][H][S][D+2] = {m, l, o[0..D)}
    float* out;            // combine: [B][hidden]
    int H, l_cap, S, hidden;   // S = number of chunks the grid covers = ceil(l_cap / chunk)
    long long kv_bstride;
    float sqrt_d;          // sqrt(D): scores are divided by it, as the reference does
};

__device__ __forceinline__ int attn_len(const AttnDecArgs& a, int b) {
    if (a.len_dev) return a.len_dev[b];
    if (a.fixed_len > 0) return a.fixed_len;
    return a.pos[b] + 1;
}

// grid (S, H, B), 256 threads; chunk = 32*STEPS keys: wave w, step i, lane group g -> key k0 + 32*i + 8*w + g.
template <int D, int STEPS>
__global__ __launch_bounds__(ER_WG) void attn_decode_f32_kernel(AttnDecArgs a) {
    constexpr int NV = D / 32;   // float4 per lane per key (8 lanes per key)
    constexpr int CHUNK = 32 * STEPS;
    static_assert(D % 32 == 0, "head_dim must be a multiple of 32");
    __shared__ __attribute__((aligned(16))) float ored[ER_NWAVES * D];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & 7, g = lane >> 3;
    const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = attn_len(a, b);
    const int k0 = s * CHUNK;
    if (k0 >= len) return;                    // inactive chunk: the merge only visits ceil(len/CHUNK) partials
    const int k1 = min(len, k0 + CHUNK);
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;
    const float* kb = a.kcache + head_off;
    const float* vb = a.vcache + head_off;

    // ---- all loads first: K rows, then V rows (K returns first, V lands while the scores are reduced)
    f32x4 kreg[STEPS][NV], vreg[STEPS][NV];
    bool valid[STEPS];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int kk = k0 + 32 * i + 8 * wid + g;
        valid[i] = kk < k1;
        const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) kreg[i][j] = kr[j * 8 + p];
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int kk = k0 + 32 * i + 8 * wid + g;
        const f32x4* vr = reinterpret_cast<const f32x4*>(vb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) vreg[i][j] = vr[j * 8 + p];
    }
    f32x4 qv[NV];
    const f32x4* qp = reinterpret_cast<const f32x4*>(a.q + (long long)b * a.hidden + h * D);
#pragma unroll
    for (int j = 0; j < NV; ++j) qv[j] = qp[j * 8 + p];

    // ---- scores q.k / sqrt(D) (every lane of an 8-lane group ends up holding its key's score)
    float sc[STEPS];
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) acc = dot4(qv[j], kreg[i][j], acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        sc[i] = valid[i] ? acc / a.sqrt_d : -INFINITY;
        mloc = fmaxf(mloc, sc[i]);
    }
    const float m = block_max(mloc, red);     // finite: the chunk holds at least one key

    // ---- chunk-local softmax weights and their sum (each key counted once: lanes with p == 0)
    float pw[STEPS];
    float lloc = 0.f;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        pw[i] = valid[i] ? expf(sc[i] - m) : 0.f;
        if (p == 0) lloc += pw[i];
    }
    const float l = block_sum(lloc, red);

    // ---- o = sum_k p_k V_k
    f32x4 o[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            o[j].x = fmaf(pw[i], vreg[i][j].x, o[j].x);
            o[j].y = fmaf(pw[i], vreg[i][j].y, o[j].y);
            o[j].z = fmaf(pw[i], vreg[i][j].z, o[j].z);
            o[j].w = fmaf(pw[i], vreg[i][j].w, o[j].w);
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
    float* pout = a.part + (((long long)b * a.H + h) * a.S + s) * (D + 2);
    if (tid < D) pout[2 + tid] = (ored[tid] + ored[D + tid]) + (ored[2 * D + tid] + ored[3 * D + tid]);
    if (tid == 0) { pout[0] = m; pout[1] = l; }
}

// grid (H, B), 256 threads: merge the active partial softmaxes of one (b, h).  Latency-bound (a few KB
// from L2), so it is organised as two rounds of independent loads instead of a serial walk over the
// partials: round 1 fetches every {m_s, l_s} at once (one per thread) and turns them into the merge
// weights w_s = exp(m_s - M) in LDS; round 2 has thread (c, half) accumulate column c over the
// partials of its half with unrolled, independent loads.
template <int D, int STEPS>
__global__ __launch_bounds__(ER_WG) void attn_combine_f32_kernel(AttnDecArgs a) {
    constexpr int CHUNK = 32 * STEPS;
    constexpr int W = D + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [S] merge weights + [128] half sums + [8]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int n_act = (attn_len(a, b) + CHUNK - 1) / CHUNK;
    float* wts = smem;
    float* half1 = smem + a.S;
    float* red = half1 + 128;
    const float* pb = a.part + ((long long)b * a.H + h) * a.S * W;
    // round 1
    float mloc = -INFINITY;
    for (int s = tid; s < n_act; s += ER_WG) mloc = fmaxf(mloc, pb[s * W]);
    const float M = block_max(mloc, red);
    float lloc = 0.f;
    for (int s = tid; s < n_act; s += ER_WG) {
        const float w = expf(pb[s * W] - M);
        wts[s] = w;
        lloc = fmaf(pb[s * W + 1], w, lloc);
    }
    const float l = block_sum(lloc, red);            // barriers inside publish wts[]
    // round 2
    const int c = tid & 127, half = tid >> 7;
    float o = 0.f;
    if (c < D) {
        const float* col = pb + 2 + c;
        int s = half;
        for (; s + 14 < n_act; s += 16) {            // 8 independent loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(s + 2 * u) * W];
#pragma unroll
            for (int u = 0; u < 8; ++u) o = fmaf(v[u], wts[s + 2 * u], o);
        }
        for (; s < n_act; s += 2) o = fmaf(col[s * W], wts[s], o);
    }
    if (half == 1) half1[c] = o;
    __syncthreads();
    if (half == 0 && c < D) a.out[(long long)b * a.hidden + h * D + c] = (o + half1[c]) / l;
}

constexpr int ATTN_STEPS_DEFAULT = 4;          // 128 keys per workgroup
inline int attn_num_chunks(int l_cap, int steps) { return (l_cap + 32 * steps - 1) / (32 * steps); }

template <int D>
inline hipError_t launch_attn_partial_d(const AttnDecArgs& a, int steps, int B, hipStream_t st) {
    const dim3 grid(a.S, a.H, B), blk(ER_WG);
    if (steps == 2) hipLaunchKernelGGL((attn_decode_f32_kernel<D, 2>), grid, blk, 0, st, a);
    else if (steps == 8) hipLaunchKernelGGL((attn_decode_f32_kernel<D, 8>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((attn_decode_f32_kernel<D, 4>), grid, blk, 0, st, a);
    return hipGetLastError();
}
template <int D>
inline hipError_t launch_attn_combine_d(const AttnDecArgs& a, int steps, int B, hipStream_t st) {
    const dim3 grid(a.H, B), blk(ER_WG);
    const size_t lds = (size_t)(a.S + 128 + 8) * sizeof(float);
    if (steps == 2) hipLaunchKernelGGL((attn_combine_f32_kernel<D, 2>), grid, blk, lds, st, a);
    else if (steps == 8) hipLaunchKernelGGL((attn_combine_f32_kernel<D, 8>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((attn_combine_f32_kernel<D, 4>), grid, blk, lds, st, a);
    return hipGetLastError();
}

}  // namespace er
