// Element-wise / row kernels of the DiT denoiser (scope row f3): adaLN-single modulation, timestep
// embedding, SiLU, and the DDIM + classifier-free-guidance update.  HBM-bound streaming kernels.
#pragma once
#include "er_common.h"

namespace er {

// y = LayerNorm_noaffine(x, eps) * (1 + scale_b) + shift_b          (core/transformer/dit.py:131-132,137-138,192-193)
// with  scale_b[c] = table[scale_idx][c] + tvec[b*t_bstride + scale_idx*t_cstride + c]   (same for shift):
//   DiTLayer: table = scale_shift_table [6][C], tvec = t_adaln [B][6][C]  (t_bstride 6C, t_cstride C)
//   DiT out:  table = scale_shift_table [2][C], tvec = t_emb   [B][C]     (t_bstride C,  t_cstride 0)
// One wave per row (row in registers).  In-place allowed.  y16 (optional): the same row rounded to fp16 - the A operand of the
// Linear that follows (gemm_hh_mfma_kernel).
// RPW rows per wave (all of ONE batch element - the launcher guarantees RPW | rows_per_batch): the four modulation vectors of a batch
// element are 16 KB that every row used to re-read from L2 next to its own 4 KB (round 3: 15.3 us per launch); a wave now loads them
// once for RPW rows.  Per-row arithmetic unchanged.
template <int CPL, int RPW = 1>
__global__ __launch_bounds__(ER_WG) void ln_modulate_rows_kernel(const float* x, float* y, int rows, int rows_per_batch,
                                                                 const float* table, const float* tvec, long long t_bstride,
                                                                 long long t_cstride, int shift_idx, int scale_idx, float eps,
                                                                 _Float16* y16) {
    constexpr int C = CPL * 64, V = CPL / 4;        // a lane holds V float4: columns 4 * (lane + 64 i) .. + 3 (16-byte accesses throughout)
    static_assert(CPL % 4 == 0, "row width must be a multiple of 256");
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * ER_NWAVES + (threadIdx.x >> 6)) * RPW;
    if (r0 >= rows) return;
    const int b = r0 / rows_per_batch;
    const float* tb = tvec + (long long)b * t_bstride;
    // every load goes out before the first use: the rows, then the four modulation operands (y may alias x and the tables are
    // not restrict-qualified, so left in the store loop their loads would queue behind the stores of the previous column group)
    f32x4 v[RPW][V], sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long long)min(r0 + j, rows - 1) * C);
#pragma unroll
        for (int i = 0; i < V; ++i) v[j][i] = xr[lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c4 = lane + 64 * i;
        sc[i] = reinterpret_cast<const f32x4*>(table + scale_idx * C)[c4] + reinterpret_cast<const f32x4*>(tb + scale_idx * t_cstride)[c4];
        sh[i] = reinterpret_cast<const f32x4*>(table + shift_idx * C)[c4] + reinterpret_cast<const f32x4*>(tb + shift_idx * t_cstride)[c4];
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = r0 + j;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) s += (v[j][i].x + v[j][i].y) + (v[j][i].z + v[j][i].w);
        const float mean = wave_sum(s) / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float d0 = v[j][i].x - mean, d1 = v[j][i].y - mean, d2 = v[j][i].z - mean, d3 = v[j][i].w - mean;
            s2 = fmaf(d0, d0, s2); s2 = fmaf(d1, d1, s2); s2 = fmaf(d2, d2, s2); s2 = fmaf(d3, d3, s2);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)C + eps);
        if (r >= rows) continue;                     // wave-uniform (only when RPW does not divide rows: the launcher avoids it)
        f32x4* yr = reinterpret_cast<f32x4*>(y + (long long)r * C);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c4 = lane + 64 * i;
            f32x4 o;
            o.x = (v[j][i].x - mean) * rstd * (1.0f + sc[i].x) + sh[i].x;
            o.y = (v[j][i].y - mean) * rstd * (1.0f + sc[i].y) + sh[i].y;
            o.z = (v[j][i].z - mean) * rstd * (1.0f + sc[i].z) + sh[i].z;
            o.w = (v[j][i].w - mean) * rstd * (1.0f + sc[i].w) + sh[i].w;
            yr[c4] = o;
            if (y16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                reinterpret_cast<h4*>(y16 + (long long)r * C)[c4] = (h4){(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
            }
        }
    }
}

// gate_b[c] = table[idx][c] + t_adaln[b][idx][c]  -> out [B][C]   (the adaLN gates fed to the GEMM epilogue)
__global__ void adaln_gate_kernel(const float* table, const float* tada, float* out, int B, int C, int idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    out[i] = table[idx * C + c] + tada[((long long)b * 6 + idx) * C + c];
}

// the gates of EVERY layer in one launch: out[(layer * 2 + which) * B * C + b * C + c], which 0 -> chunk 2 (gate_msa), 1 -> chunk 5
// (gate_mlp); tables = device array of the layers' scale_shift_table pointers.  (One launch instead of two ~4 us launches per layer.)
__global__ void adaln_gate_all_kernel(const float* const* tables, const float* tada, float* out, int layers, int B, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)B * C;
    if (i >= per * 2 * layers) return;
    const int lw = (int)(i / per), l = lw >> 1, idx = (lw & 1) ? 5 : 2;
    const int r = (int)(i - lw * per), b = r / C, c = r - b * C;
    out[i] = tables[l][idx * C + c] + tada[((long long)b * 6 + idx) * C + c];
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

// ---------------------------------------------------------------------------------------------------
// CLIP ViT image encoder front (core/models_dit.py:104-111 -> HF CLIPVisionModel)
namespace er {

// TF.normalize(mean, std) then F.interpolate(bilinear, align_corners=False) to OUT x OUT   (models_dit.py:108-109)
__global__ void clip_preprocess_kernel(const float* img, float* out, int B, int H, int W, int OUT) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * 3 * OUT * OUT;
    if (i >= total) return;
    const int ox = (int)(i % OUT), oy = (int)((i / OUT) % OUT), c = (int)((i / ((long long)OUT * OUT)) % 3);
    const long long b = i / ((long long)3 * OUT * OUT);
    const float mean = c == 0 ? 0.48145466f : (c == 1 ? 0.4578275f : 0.40821073f);
    const float stdv = c == 0 ? 0.26862954f : (c == 1 ? 0.26130258f : 0.27577711f);
    const float sy = fmaxf(((float)oy + 0.5f) * ((float)H / (float)OUT) - 0.5f, 0.f);
    const float sx = fmaxf(((float)ox + 0.5f) * ((float)W / (float)OUT) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* p = img + (b * 3 + c) * (long long)H * W;
    const float v00 = (p[(long long)y0 * W + x0] - mean) / stdv, v01 = (p[(long long)y0 * W + x1] - mean) / stdv;
    const float v10 = (p[(long long)y1 * W + x0] - mean) / stdv, v11 = (p[(long long)y1 * W + x1] - mean) / stdv;
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// Conv2d(3, width, kernel P, stride P, no bias) as a GEMM: A[b*G*G + py*G + px][c*P*P + ky*P + kx] (zero-padded to lda)
__global__ void clip_im2col_kernel(const float* px, float* A, int B, int S, int P, int lda) {
    const int G = S / P;
    const long long total = (long long)B * G * G * lda;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % lda);
    const long long row = i / lda;
    const int gx = (int)(row % G), gy = (int)((row / G) % G);
    const long long b = row / ((long long)G * G);
    float v = 0.f;
    if (col < 3 * P * P) {
        const int c = col / (P * P), r = col - c * P * P, ky = r / P, kx = r - ky * P;
        v = px[((b * 3 + c) * S + (gy * P + ky)) * (long long)S + gx * P + kx];
    }
    A[i] = v;
}

// x[b][0] = class_embedding + pos[0]; x[b][1+p] = patches[b][p] + pos[1+p]     (CLIPVisionEmbeddings.forward)
__global__ void clip_assemble_kernel(const float* patches, const float* cls, const float* pos, float* x, int B, int NP, int C) {
    const long long total = (long long)B * (NP + 1) * C;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int tkn = (int)((i / C) % (NP + 1));
    const long long b = i / ((long long)(NP + 1) * C);
    const float base = tkn == 0 ? cls[c] : patches[(b * NP + (tkn - 1)) * C + c];
    x[i] = base + pos[(long long)tkn * C + c];
}

__global__ void gelu_kernel(const float* x, float* y, long long n) {       // exact (erf) GELU, in place allowed
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
}

}  // namespace er
