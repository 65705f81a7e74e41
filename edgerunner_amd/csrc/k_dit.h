// Element-wise / row kernels of the DiT denoiser (scope row f3): adaLN-single modulation, timestep
// embedding, SiLU, and the DDIM + classifier-free-guidance update.  HBM-bound streaming kernels.
#pragma once
#include "er_common.h"

namespace er {

// y = LayerNorm_noaffine(x, eps) * (1 + scale_b) + shift_b          (core/transformer/dit.py:131-132,137-138,192-193)
// with  scale_b[c] = table[scale_idx][c] + tvec[b*t_bstride + scale_idx*t_cstride + c]   (same for shift):
//   DiTLayer: table = scale_shift_table [6][C], tvec = t_adaln [B][6][C]  (t_bstride 6C, t_cstride C)
//   DiT out:  table = scale_shift_table [2][C], tvec = t_emb   [B][C]     (t_bstride C,  t_cstride 0)
// One wave per row (row in registers).  In-place allowed.
template <int CPL>
__global__ __launch_bounds__(ER_WG) void ln_modulate_rows_kernel(const float* x, float* y, int rows, int rows_per_batch,
                                                                 const float* table, const float* tvec, long long t_bstride,
                                                                 long long t_cstride, int shift_idx, int scale_idx, float eps) {
    constexpr int C = CPL * 64;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * ER_NWAVES + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int b = r / rows_per_batch;
    const float* xr = x + (long long)r * C;
    float v[CPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { v[i] = xr[lane + 64 * i]; s += v[i]; }
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { const float d = v[i] - mean; s2 = fmaf(d, d, s2); }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)C + eps);
    const float* tb = tvec + (long long)b * t_bstride;
    float* yr = y + (long long)r * C;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        const float scale = table[scale_idx * C + c] + tb[scale_idx * t_cstride + c];
        const float shift = table[shift_idx * C + c] + tb[shift_idx * t_cstride + c];
        yr[c] = (v[i] - mean) * rstd * (1.0f + scale) + shift;
    }
}

// gate_b[c] = table[idx][c] + t_adaln[b][idx][c]  -> out [B][C]   (the adaLN gates fed to the GEMM epilogue)
__global__ void adaln_gate_kernel(const float* table, const float* tada, float* out, int B, int C, int idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    out[i] = table[idx * C + c] + tada[((long long)b * 6 + idx) * C + c];
}

// Timesteps(256) (dit.py:45-77): emb[b] = [sin(t*w_k), cos(t*w_k)], w_k = exp(-ln(10000) * k / 128)
__global__ void timestep_embed_kernel(const float* t, float* out, int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i - b * half;
    const float w = expf((-9.210340371976184f * (float)k) / (float)half);
    const float a = t[b] * w;
    out[(long long)b * 2 * half + k] = sinf(a);
    out[(long long)b * 2 * half + half + k] = cosf(a);
}

__global__ void silu_kernel(const float* x, float* y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v / (1.0f + expf(-v)); }
}

// One DDIM (eta = 0, v-prediction) step with classifier-free guidance (core/models_dit.py:222-227):
//   v = v_uncond + g * (v_cond - v_uncond);  x0 = sa_t*x - sb_t*v;  eps = sa_t*v + sb_t*x;  x <- sa_p*x0 + sb_p*eps
// pred holds [uncond rows | cond rows], each n elements.
__global__ void ddim_cfg_step_kernel(float* latents, const float* pred, long long n, float guidance, float sa_t, float sb_t,
                                     float sa_p, float sb_p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = pred[i], c = pred[n + i], x = latents[i];
    const float v = u + guidance * (c - u);
    const float x0 = sa_t * x - sb_t * v;
    const float eps = sa_t * v + sb_t * x;
    latents[i] = sa_p * x0 + sb_p * eps;
}

}  // namespace er
