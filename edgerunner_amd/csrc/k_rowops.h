// Row-wise / element-wise kernels of the once-per-sample stages (prefill, point
// encoder).  All HBM-bound streaming kernels: coalesced rows, one pass where the
// row fits in registers.
#pragma once
#include "er_common.h"

namespace er {

// y[r,:] = LayerNorm(x[r,:]) * w + b   (nn.LayerNorm, biased variance, eps inside sqrt;
// core/transformer/modeling_opt.py:274,288, core/transformer/point.py:137,112-114, core/models.py:65).
// One wave per row, the row lives in registers (CPL = cols/64 values per lane).
template <int CPL>
__global__ __launch_bounds__(ER_WG) void layernorm_rows_kernel(const float* x, const float* w, const float* b,
                                                               float* y, int rows, long long ldx, long long ldy,
                                                               float eps) {
    constexpr int COLS = CPL * 64;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * ER_NWAVES + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (long long)r * ldx;
    // every load up front: y may alias x and w / b are not restrict-qualified, so left in the store loop the affine parameters
    // would be fetched one column group at a time BEHIND the previous group's stores (18 us per launch for 25 MB in round 2)
    float v[CPL], wv[CPL], bv[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = xr[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < CPL; ++i) { wv[i] = w[lane + 64 * i]; bv[i] = b[lane + 64 * i]; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)COLS;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { const float d = v[i] - mean; s2 = fmaf(d, d, s2); }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)COLS + eps);
    float* yr = y + (long long)r * ldy;
#pragma unroll
    for (int i = 0; i < CPL; ++i) yr[lane + 64 * i] = (v[i] - mean) * rstd * wv[i] + bv[i];
}

inline hipError_t launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int cols,
                                   long long ldx, long long ldy, float eps, hipStream_t st) {
    const int grid = (rows + ER_NWAVES - 1) / ER_NWAVES;
    if (rows == 0) return hipSuccess;
    switch (cols) {
        case 1536: hipLaunchKernelGGL((layernorm_rows_kernel<24>), dim3(grid), dim3(ER_WG), 0, st, x, w, b, y, rows, ldx, ldy, eps); break;
        case 1280: hipLaunchKernelGGL((layernorm_rows_kernel<20>), dim3(grid), dim3(ER_WG), 0, st, x, w, b, y, rows, ldx, ldy, eps); break;
        case 1024: hipLaunchKernelGGL((layernorm_rows_kernel<16>), dim3(grid), dim3(ER_WG), 0, st, x, w, b, y, rows, ldx, ldy, eps); break;
        case 512:  hipLaunchKernelGGL((layernorm_rows_kernel<8>), dim3(grid), dim3(ER_WG), 0, st, x, w, b, y, rows, ldx, ldy, eps); break;
        case 256:  hipLaunchKernelGGL((layernorm_rows_kernel<4>), dim3(grid), dim3(ER_WG), 0, st, x, w, b, y, rows, ldx, ldy, eps); break;
        case 64:   hipLaunchKernelGGL((layernorm_rows_kernel<1>), dim3(grid), dim3(ER_WG), 0, st, x, w, b, y, rows, ldx, ldy, eps); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// In-place row softmax of attention scores (core/transformer/attention.py:53-57):
// row r keeps columns [0, n_valid) with n_valid = causal ? min(cols, r+1+causal_off) : cols
// (the reference adds a -inf upper triangle, whose softmax weight is exactly 0) and
// columns [n_valid, ld_pad) are written as 0 so the following P.V GEMM may run over a
// 16-padded K.  grid (rows, batch), one workgroup per row.
__global__ __launch_bounds__(ER_WG) void softmax_rows_kernel(float* s, int rows, int cols, long long ld, int ld_pad,
                                                             long long batch_stride, int causal, int causal_off, int use_lds) {
    __shared__ float red[8];
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];   // use_lds: the row is staged here -> 1 read + 1 write of HBM
    const int r = blockIdx.x, tid = threadIdx.x;
    float* row = s + blockIdx.y * batch_stride + (long long)r * ld;
    const int nv = causal ? min(cols, r + 1 + causal_off) : cols;
    float* w = use_lds ? rowbuf : row;          // same per-thread element order either way: identical results
    float m = -INFINITY;
    for (int c = tid; c < nv; c += ER_WG) {
        const float v = row[c];
        if (use_lds) rowbuf[c] = v;
        m = fmaxf(m, v);
    }
    m = block_max(m, red);
    float l = 0.f;
    for (int c = tid; c < nv; c += ER_WG) {
        const float e = expf(w[c] - m);
        w[c] = e;
        l += e;
    }
    l = block_sum(l, red);
    for (int c = tid; c < nv; c += ER_WG) row[c] = w[c] / l;
    for (int c = nv + tid; c < ld_pad; c += ER_WG) row[c] = 0.f;
}

inline hipError_t launch_softmax_rows(float* s, int rows, int cols, long long ld, int ld_pad, long long batch_stride, int batch,
                                      int causal, int causal_off, hipStream_t st) {
    const int use_lds = cols <= 15 * 1024;       // <= 60 KiB of LDS; longer rows fall back to in-place passes
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows, batch), dim3(ER_WG), use_lds ? (size_t)cols * sizeof(float) : 0, st, s, rows,
                       cols, ld, ld_pad, batch_stride, causal, causal_off, use_lds);
    return hipGetLastError();
}

// GEGLU (core/transformer/point.py:68-71): out[m, j] = u[m, j] * gelu_erf(u[m, F + j]).
__global__ __launch_bounds__(ER_WG) void geglu_kernel(const float* u, float* out, long long rows, int F) {
    const long long total = rows * F;
    for (long long i = (long long)blockIdx.x * ER_WG + threadIdx.x; i < total; i += (long long)gridDim.x * ER_WG) {
        const long long m = i / F;
        const int j = (int)(i - m * F);
        const float x = u[m * 2 * F + j];
        const float gt = u[m * 2 * F + F + j];
        const float gelu = gt * 0.5f * (1.0f + erff(gt * 0.70710678118654752440f));
        out[i] = x * gelu;
    }
}

// hidden = inputs_embeds + embed_positions(arange(S))   (core/transformer/modeling_opt.py:355-357)
__global__ __launch_bounds__(ER_WG) void add_pos_kernel(const float* emb, const float* pos, float* out, int B, int S,
                                                        int C, int pos0) {
    const long long total = (long long)B * S * C / 4;
    const int c4 = C / 4;
    for (long long i = (long long)blockIdx.x * ER_WG + threadIdx.x; i < total; i += (long long)gridDim.x * ER_WG) {
        const long long row = i / c4;
        const int c = (int)(i - row * c4);
        const int s = (int)(row % S);
        const f32x4 a = reinterpret_cast<const f32x4*>(emb)[i];
        const f32x4 p = reinterpret_cast<const f32x4*>(pos)[(long long)(pos0 + s) * c4 + c];
        reinterpret_cast<f32x4*>(out)[i] = a + p;
    }
}

// PointEmbed features (core/transformer/point.py:53-63): for point m,
// out[m, 0:F) = sin(x.basis), out[m, F:2F) = cos(x.basis), out[m, 2F:2F+3) = xyz, zero-padded to ldo
// (the 51-wide Linear input padded to a multiple of 16 for the GEMM).
__global__ __launch_bounds__(ER_WG) void point_embed_kernel(const float* pts, const float* basis, float* out,
                                                            long long M, int F, int ldo) {
    const long long total = M * ldo;
    for (long long i = (long long)blockIdx.x * ER_WG + threadIdx.x; i < total; i += (long long)gridDim.x * ER_WG) {
        const long long m = i / ldo;
        const int c = (int)(i - m * ldo);
        const float x = pts[m * 3 + 0], y = pts[m * 3 + 1], z = pts[m * 3 + 2];
        float v = 0.f;
        if (c < 2 * F) {
            const int e = (c < F) ? c : c - F;
            float proj = x * basis[e];
            proj = fmaf(y, basis[F + e], proj);
            proj = fmaf(z, basis[2 * F + e], proj);
            v = (c < F) ? sinf(proj) : cosf(proj);
        } else if (c < 2 * F + 3) {
            v = pts[m * 3 + (c - 2 * F)];
        }
        out[i] = v;
    }
}

// out[r, :] = table[ids[r], :]   (nn.Embedding lookups: core/models.py:137,228)
__global__ __launch_bounds__(ER_WG) void gather_rows_kernel(const float* table, const int* ids, float* out, int rows,
                                                            int C, long long ldo) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const float* src = table + (long long)ids[r] * C;
    float* dst = out + (long long)r * ldo;
    for (int c = threadIdx.x; c < C; c += ER_WG) dst[c] = src[c];
}

// Fast mode prefill: move the K/V columns of the fused projection output qkv[M][3*hidden] into the fp16
// cache [B][H][Lcap][D] (what modeling_opt.py:189-192 stores, in the reference's GPU dtype) and round the
// scratch copy through fp16 in place, so the prefix attention sees exactly the values later steps will read.
// grid (ceil(2 * hidden / 4 / 256), rows): a token row per blockIdx.y (strided), four consecutive columns per thread (head_dim % 4 == 0:
// they stay inside one head).  Round 2's element-per-thread form with a 64-bit and two 32-bit index divisions per ELEMENT was
// bound by its index arithmetic: 26 us per layer for 63 MB (profiles/r02_prefill_fp16_kernel_stats.csv).
__global__ __launch_bounds__(ER_WG) void kv_scatter_half_kernel(float* qkv, _Float16* kcache, _Float16* vcache, int M,
                                                                int S, int hidden, int head_dim, int l_cap,
                                                                long long kv_bstride) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const int c2 = (blockIdx.x * ER_WG + threadIdx.x) * 4;
    if (c2 >= 2 * hidden) return;
    const int which = c2 >= hidden ? 1 : 0, c = c2 - which * hidden;       // 0 = K, 1 = V
    const int h = c / head_dim, d = c - h * head_dim;
    _Float16* cache = which == 0 ? kcache : vcache;
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        const int b = m / S, s = m - b * S;
        f32x4* src = reinterpret_cast<f32x4*>(qkv + (long long)m * 3 * hidden + hidden + c2);
        const f32x4 v = *src;
        const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *src = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        *reinterpret_cast<h4*>(cache + (long long)b * kv_bstride + ((long long)h * l_cap + s) * head_dim + d) = hv;
    }
}
inline dim3 kv_scatter_grid(int M, int hidden) {
    return dim3((unsigned)((2 * hidden / 4 + ER_WG - 1) / ER_WG), (unsigned)(M < 65535 ? (M > 0 ? M : 1) : 65535));
}

inline int ew_grid(long long total) {
    long long g = (total + ER_WG - 1) / ER_WG;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace er
