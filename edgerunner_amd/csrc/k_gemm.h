// fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: an exact,
// k-ordered fmaf chain at the fp32 vector rate) for the once-per-sample work:
// the 2050-token prefill (core/transformer/modeling_opt.py:321-426 with
// inputs_embeds) and the point-cloud encoder (core/transformer/point.py:186-206).
// Not on the per-token path - the decode step uses k_gemv.h.
//
//   C[M,N] = epilogue( A[M,K] . op(B) )       op(B) = B^T for B [N,K] (nn.Linear weight / K-cache rows)
//                                              op(B) = B   for B [K,N] (V rows, "NN")
// Batched over blockIdx.z with two-level strides (batch, head).  Tile (64 TM) x (64 TN) x 32 in two LDS stages,
// 4 waves as 2x2, each wave TM x TN MFMA tiles of 32x32 (TM, TN in {1, 2}: the launcher picks the largest tile that still
// gives every CU several workgroups - the 2050-row prefill and the 4096-row DiT GEMMs have only 200-800 tiles of 128x128,
// i.e. ONE 4-wave workgroup per CU with nothing to hide its barriers and LDS reads behind).  Operands are staged k-major
// in LDS so that an MFMA operand fetch (lane l needs element [k = l>>5][i = l&31])
// is a conflict-free ds_read_b32; the next k-tile is prefetched into registers while
// the current one is multiplied.
#pragma once
#include "er_common.h"

namespace er {

enum { GEPI_PLAIN = 0, GEPI_QKV = 1 };

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;        // [N] or null
    const float* resid;       // [M][ldr] or null (added last)
    int resid_mod;            // > 0: the residual has resid_mod rows and row m reads row m % resid_mod (one table shared by a batch)
    const float* gate;        // or null: per-(batch,column) multiplier applied before the residual add:
    int gate_rows;            //   v = resid + gate[(m / gate_rows) * gate_bstride + n] * (acc + bias)   (adaLN gates)
    long long gate_bstride;
    int M, N, K;              // K % 16 == 0; A must be readable (and zero-padded) up to K
    int lda, ldb, ldc, ldr;
    long long sA1, sA2, sB1, sB2, sC1, sC2;   // offsets = (z / Z2) * s?1 + (z % Z2) * s?2
    int Z2;
    int b_is_kn;              // 0: B is [N][K]; 1: B is [K][N]
    int kb_valid;             // NN: rows k >= kb_valid of B are treated as zero
    float div;                // != 0: acc / div first (attention scale, as the reference divides)
    int relu;
    int causal;               // 1: skip tiles entirely above the diagonal (n0 > m0+127+causal_off); for NN: k beyond it
    int causal_off;
    int epi;                  // GEPI_*
    // GEPI_QKV: rows m = b*S + s; cols [0,h) -> q[m][c], [h,2h) -> K cache, [2h,3h) -> V cache
    float* q; float* kcache; float* vcache;
    int S, hidden, head_dim, l_cap;
    long long kv_bstride;
};

constexpr int GBK = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// epilogue shared by the fp32 and fp16-input kernels: C/D fragment map of the 32x32 MFMA (dtype independent):
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, float* C, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                              int wn, int kh, int li) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int gn = n0 + wn * 32 * TN + j * 32 + li;
                if (gm >= g.M || gn >= g.N) continue;
                float v = acc[i][j][r];
                if (g.div != 0.f) v = v / g.div;
                if (g.bias) v += g.bias[gn];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.gate) v *= g.gate[(long long)(gm / g.gate_rows) * g.gate_bstride + gn];
                if (g.resid) v += g.resid[(long long)(g.resid_mod > 0 ? gm % g.resid_mod : gm) * g.ldr + gn];
                if (g.epi == GEPI_PLAIN) {
                    C[(long long)gm * g.ldc + gn] = v;
                } else {
                    const int which = gn / g.hidden, c = gn - which * g.hidden;
                    if (which == 0) {
                        g.q[(long long)gm * g.hidden + c] = v;
                    } else {
                        const int b = gm / g.S, s = gm - b * g.S;
                        const int h = c / g.head_dim, d = c - h * g.head_dim;
                        float* cache = (which == 1) ? g.kcache : g.vcache;
                        cache[(long long)b * g.kv_bstride + ((long long)h * g.l_cap + s) * g.head_dim + d] = v;
                    }
                }
            }
}

template <int TM, int TN>
__global__ __launch_bounds__(ER_WG) void gemm_f32_mfma_kernel(GemmArgs g) {
    constexpr int GBM = 64 * TM, GBN = 64 * TN, GLD = GBN + 4;
    // two LDS stages: tile kt+1 is written while tile kt is multiplied -> ONE workgroup barrier per k-tile (round 1: two
    // barriers per 16-wide tile kept the matrix pipe 44 % busy)
    // LDS row strides: the transposing (k-major) stores of a thread's float4 go to 4 rows, and the 32 lanes of a store group
    // hold (k-quad 0..7, row 0..3): with a stride = 1 (mod 8) those 32 addresses fall into 32 different banks (stride 132
    // put them into 8 banks: PMC showed half of all LDS cycles were bank conflicts).  The NN B tile is stored row-wise with
    // 16-byte stores and keeps the 16-byte aligned stride.
    constexpr int GLT = GBM + 1, GLTB = GBN + 1;
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLT];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    const int ldbs = g.b_is_kn ? GLD : GLTB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    if (g.causal && !g.b_is_kn && n0 > m0 + GBM - 1 + g.causal_off) return;   // fully masked score tile
    const int z = blockIdx.z, z1 = z / g.Z2, z2 = z - z1 * g.Z2;
    const float* A = g.A + z1 * g.sA1 + z2 * g.sA2;
    const float* B = g.B + z1 * g.sB1 + z2 * g.sB2;
    float* C = g.C + z1 * g.sC1 + z2 * g.sC2;

    int K = g.K;
    if (g.causal && g.b_is_kn) {              // P.V: probabilities beyond the diagonal are exactly zero
        const int klim = (m0 + GBM + g.causal_off + 15) / 16 * 16;
        K = min(K, klim);
    }
    const int nk = (K + GBK - 1) / GBK;       // K % 16 == 0; a trailing half tile is zero-filled

    // global -> register staging maps (a k-tile is GBK = 32 wide: 8 float4 per row)
    constexpr int NRA = GBM / 32, NRB = GBN / 32;          // 32-row passes of the A / B(NT) tile
    constexpr int QN = GBN / 4, RPN = ER_WG / QN;          // B(NN): n-quads per k row, k rows per pass (NRB passes cover the 32 k rows)
    const int ar = tid >> 3, akq = tid & 7;   // A / B(NT): row (0..31, +32 per pass), k-quad
    const int bkr = tid / QN, bnq = tid % QN; // B(NN): k row (+RPN per pass), n-quad
    f32x4 ra[NRA], rb[NRB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kt) {
        const int k0 = kt * GBK;
        const bool kin = k0 + 4 * akq < K;
#pragma unroll
        for (int hh = 0; hh < NRA; ++hh) {
            const int gm = m0 + ar + 32 * hh;
            ra[hh] = (gm < g.M && kin) ? *reinterpret_cast<const f32x4*>(A + (long long)gm * g.lda + k0 + 4 * akq) : zero4;
        }
        if (!g.b_is_kn) {
#pragma unroll
            for (int hh = 0; hh < NRB; ++hh) {
                const int gn = n0 + ar + 32 * hh;
                rb[hh] = (gn < g.N && kin) ? *reinterpret_cast<const f32x4*>(B + (long long)gn * g.ldb + k0 + 4 * akq) : zero4;
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < NRB; ++hh) {
                const int gk = k0 + bkr + RPN * hh;
                const int gn = n0 + 4 * bnq;
                rb[hh] = (gk < g.kb_valid && gk < K && gn < g.N) ? *reinterpret_cast<const f32x4*>(B + (long long)gk * g.ldb + gn) : zero4;
            }
        }
    };
    auto store_tile = [&](int s) {
        float* as = As[s];
        float* bs = Bs[s];
#pragma unroll
        for (int hh = 0; hh < NRA; ++hh) {
            const int m = ar + 32 * hh;
            as[(4 * akq + 0) * GLT + m] = ra[hh].x;
            as[(4 * akq + 1) * GLT + m] = ra[hh].y;
            as[(4 * akq + 2) * GLT + m] = ra[hh].z;
            as[(4 * akq + 3) * GLT + m] = ra[hh].w;
        }
        if (!g.b_is_kn) {
#pragma unroll
            for (int hh = 0; hh < NRB; ++hh) {
                const int n = ar + 32 * hh;
                bs[(4 * akq + 0) * GLTB + n] = rb[hh].x;
                bs[(4 * akq + 1) * GLTB + n] = rb[hh].y;
                bs[(4 * akq + 2) * GLTB + n] = rb[hh].z;
                bs[(4 * akq + 3) * GLTB + n] = rb[hh].w;
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < NRB; ++hh)
                *reinterpret_cast<f32x4*>(&bs[(bkr + RPN * hh) * GLD + 4 * bnq]) = rb[hh];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int kh = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            const float* ap = As[cur] + (2 * kk + kh) * GLT + wm * 32 * TM + li;
            const float* bp = Bs[cur] + (2 * kk + kh) * ldbs + wn * 32 * TN + li;
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = ap[32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = bp[32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        // the other stage was last read in iteration kt-1, and every wave passed that iteration's barrier since
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<TM, TN>(g, C, acc, m0, n0, wm, wn, kh, li);
}

// ---------------------------------------------------------------------------------------------------
// fp16-input variant for the compute-bound front-end GEMMs (DiT / CLIP linears in fast mode):
//   C[M,N] = epilogue( fp16(A[M,K] fp32) . fp16 W[N,K]^T ), fp32 accumulate on v_mfma_f32_32x32x16_f16
// (16x the fp32 matrix rate).  Tile 128x128x32; A is rounded to fp16 on its way into LDS, W is stored fp16.
// LDS rows are [row][32 k] halves padded to 80 bytes so the 16-byte operand reads (lane l: row l&31,
// k-block (l>>5)*8) hit 16 distinct 4-bank slots per 16-lane group.  Both operands use the same
// (lane-half, element) -> k assignment, so the product sum is independent of the instruction's internal k order.
constexpr int HBK = 32, HLD = 40;            // halves per LDS row (32 + 8 pad)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN>
__global__ __launch_bounds__(ER_WG) void gemm_f16_mfma_kernel(GemmArgs g) {
    constexpr int GBM = 64 * TM, GBN = 64 * TN;
    // ONE LDS stage (20 KB): a k-tile is only 8 MFMAs (256 cycles) per wave, far shorter than a global-load round trip, so
    // what hides that latency is the number of workgroups per CU, not a second LDS stage (a two-stage version measured
    // slower: 21.4 vs 20.4 ms per DiT forward; PMC: the split kernel below sat at 16 % MFMA busy with two stages)
    __shared__ __attribute__((aligned(16))) _Float16 As[1][GBM * HLD];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[1][GBN * HLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const float* A = g.A;
    const _Float16* B = reinterpret_cast<const _Float16*>(g.B);
    const int nk = g.K / HBK;
    constexpr int NA = GBM / 32, NB = (GBN + 63) / 64;      // float4 of A / 16-byte pieces of B per thread and k-tile
    f32x4 ra[NA];
    f32x4 rb[NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int kt) {
        const int k0 = kt * HBK;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + ER_WG * u, row = idx >> 3, c4 = idx & 7;
            const int gm = m0 + row;
            ra[u] = (gm < g.M) ? *reinterpret_cast<const f32x4*>(A + (long long)gm * g.lda + k0 + 4 * c4) : zero4;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + ER_WG * u, row = idx >> 2, c8 = idx & 3;
            const int gn = n0 + row;
            rb[u] = (gn < g.N && row < GBN) ? *reinterpret_cast<const f32x4*>(B + (long long)gn * g.ldb + k0 + 8 * c8) : zero4;
        }
    };
    auto store_tile = [&](int s) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + ER_WG * u, row = idx >> 3, c4 = idx & 7;
            h16x4 hv = {(_Float16)ra[u].x, (_Float16)ra[u].y, (_Float16)ra[u].z, (_Float16)ra[u].w};
            *reinterpret_cast<h16x4*>(&As[s][row * HLD + 4 * c4]) = hv;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + ER_WG * u, row = idx >> 2, c8 = idx & 3;
            if (row < GBN) *reinterpret_cast<f32x4*>(&Bs[s][row * HLD + 8 * c8]) = rb[u];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int kh = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        constexpr int cur = 0;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            const int ko = ks * 16 + kh * 8;
            h16x8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const h16x8*>(&As[cur][(wm * 32 * TM + 32 * i + li) * HLD + ko]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const h16x8*>(&Bs[cur][(wn * 32 * TN + 32 * j + li) * HLD + ko]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) store_tile(0);
        __syncthreads();
    }
    gemm_epilogue<TM, TN>(g, g.C, acc, m0, n0, wm, wn, kh, li);
}

// ---------------------------------------------------------------------------------------------------
// "Split" fp16 variant for the FAST-mode prefill (fp16-stored decoder weights, fp32 activations, fp32 accumulate - the
// arithmetic the per-token GEMV path performs): the fp32 activation is split on its way into LDS into two fp16 numbers,
//   a = hi + lo,  hi = fp16(a),  lo = fp16(a - hi)      (|a - hi - lo| <= 2^-22 |a|: fp32-grade operand),
// and every weight fragment is multiplied by both (two v_mfma_f32_32x32x16_f16 per fragment pair, fp16 x fp16 products
// are exact in the fp32 accumulator).  16x the fp32 matrix rate at twice the instruction count = 8x, with results within
// fp32 round-off of the fp32-activation product - so prefill and decode keep seeing one model and the fast-mode parity
// tests (ids exact / logits vs the fp16-STORAGE emulation) hold unchanged.  Tile 128x128x32, two LDS stages.
template <int BK, int TM, int TN>
__global__ __launch_bounds__(ER_WG) void gemm_f16s_mfma_kernel(GemmArgs g) {
    constexpr int GBM = 64 * TM, GBN = 64 * TN;
    constexpr int LDH = BK + 8;            // halves per LDS row (16-byte reads of a 16-lane group hit 16 distinct 4-bank slots for 40 and 72)
    // one LDS stage (30 KB -> 5 workgroups per CU): see gemm_f16_mfma_kernel
    __shared__ __attribute__((aligned(16))) _Float16 Ah[1][GBM * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Al[1][GBM * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[1][GBN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const float* A = g.A;
    const _Float16* B = reinterpret_cast<const _Float16*>(g.B);
    const int nk = g.K / BK;
    constexpr int C4 = BK / 4, C8 = BK / 8;                                // float4 of A / 16-byte pieces of B per tile row
    constexpr int NA = GBM * C4 / ER_WG, NB = (GBN * C8 + ER_WG - 1) / ER_WG;   // ... per thread and k-tile
    f32x4 ra[NA];
    f32x4 rb[NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + ER_WG * u, row = idx / C4, c4 = idx % C4;
            const int gm = m0 + row;
            ra[u] = (gm < g.M) ? *reinterpret_cast<const f32x4*>(A + (long long)gm * g.lda + k0 + 4 * c4) : zero4;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + ER_WG * u, row = idx / C8, c8 = idx % C8;
            const int gn = n0 + row;
            rb[u] = (gn < g.N && row < GBN) ? *reinterpret_cast<const f32x4*>(B + (long long)gn * g.ldb + k0 + 8 * c8) : zero4;
        }
    };
    auto store_tile = [&](int s) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + ER_WG * u, row = idx / C4, c4 = idx % C4;
            const h16x4 hv = {(_Float16)ra[u].x, (_Float16)ra[u].y, (_Float16)ra[u].z, (_Float16)ra[u].w};
            const h16x4 lv = {(_Float16)(ra[u].x - (float)hv[0]), (_Float16)(ra[u].y - (float)hv[1]),
                              (_Float16)(ra[u].z - (float)hv[2]), (_Float16)(ra[u].w - (float)hv[3])};
            *reinterpret_cast<h16x4*>(&Ah[s][row * LDH + 4 * c4]) = hv;
            *reinterpret_cast<h16x4*>(&Al[s][row * LDH + 4 * c4]) = lv;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + ER_WG * u, row = idx / C8, c8 = idx % C8;
            if (row < GBN) *reinterpret_cast<f32x4*>(&Bs[s][row * LDH + 8 * c8]) = rb[u];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int kh = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        constexpr int cur = 0;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ko = ks * 16 + kh * 8;
            h16x8 bv[TN], ah[TM], al[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const h16x8*>(&Bs[cur][(wn * 32 * TN + 32 * j + li) * LDH + ko]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const h16x8*>(&Ah[cur][(wm * 32 * TM + 32 * i + li) * LDH + ko]);
                al[i] = *reinterpret_cast<const h16x8*>(&Al[cur][(wm * 32 * TM + 32 * i + li) * LDH + ko]);
            }
            // the small (lo) products first, then the large ones: the accumulator sees them in increasing magnitude
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bv[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) store_tile(0);
        __syncthreads();
    }
    gemm_epilogue<TM, TN>(g, g.C, acc, m0, n0, wm, wn, kh, li);
}

inline GemmArgs gemm_args_default() {
    GemmArgs g{};
    g.Z2 = 1;
    g.epi = GEPI_PLAIN;
    return g;
}

// ---- tile choice.  A 128x128 tile reads the fewest operand bytes per flop, but the once-per-sample GEMMs of this path are
// small: the 2050-row prefill has 17 x 12..48 tiles of 128x128 and the 4096-row DiT Linears 32 x 8..64, i.e. about ONE 4-wave
// workgroup per CU, whose barriers, LDS reads and global-load latency nothing hides.  Halve the tile (first along M, then
// along N) until every CU gets at least GEMM_MIN_WGS_PER_CU workgroups.  ER_GEMM_TILE = 1 (128x128), 2 (64x128), 3 (64x64)
// forces a shape for A/B runs.
constexpr int GEMM_MIN_WGS_PER_CU = 3;
inline int gemm_pick_tile(int M, int N, int batch) {
    static const int forced = [] { const char* v = getenv("ER_GEMM_TILE"); return v ? atoi(v) : 0; }();
    if (forced >= 1 && forced <= 3) return forced;
    auto wgs = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
    const long long want = 256LL * GEMM_MIN_WGS_PER_CU;
    if (wgs(128, 128) >= want) return 1;
    if (wgs(64, 128) >= want) return 2;
    return 3;
}
#define ER_GEMM_DISPATCH(KERNEL_T, g, batch, st)                                                                       \
    do {                                                                                                               \
        const int tile_ = gemm_pick_tile((g).M, (g).N, (batch));                                                       \
        const int bm_ = tile_ == 1 ? 128 : 64, bn_ = tile_ == 3 ? 64 : 128;                                            \
        const dim3 grid_(((g).N + bn_ - 1) / bn_, ((g).M + bm_ - 1) / bm_, (batch));                                   \
        if (tile_ == 1) hipLaunchKernelGGL((KERNEL_T(2, 2)), grid_, dim3(ER_WG), 0, (st), (g));                        \
        else if (tile_ == 2) hipLaunchKernelGGL((KERNEL_T(1, 2)), grid_, dim3(ER_WG), 0, (st), (g));                   \
        else hipLaunchKernelGGL((KERNEL_T(1, 1)), grid_, dim3(ER_WG), 0, (st), (g));                                   \
    } while (0)
#define ER_K_F16(TM, TN) gemm_f16_mfma_kernel<TM, TN>
#define ER_K_F16S32(TM, TN) gemm_f16s_mfma_kernel<32, TM, TN>
#define ER_K_F16S64(TM, TN) gemm_f16s_mfma_kernel<64, TM, TN>
#define ER_K_F32(TM, TN) gemm_f32_mfma_kernel<TM, TN>

inline hipError_t launch_gemm_f16(const GemmArgs& g, hipStream_t st) {   // NT only, K % 32 == 0, B = fp16 weights
    ER_GEMM_DISPATCH(ER_K_F16, g, 1, st);
    return hipGetLastError();
}

inline hipError_t launch_gemm_f16s(const GemmArgs& g, hipStream_t st) {  // NT only, K % 32 == 0, B = fp16 weights, A split hi/lo
    static const bool bk64 = [] { const char* v = getenv("ER_F16S_BK"); return v && atoi(v) == 64; }();
    if (bk64 && g.K % 64 == 0) ER_GEMM_DISPATCH(ER_K_F16S64, g, 1, st);
    else ER_GEMM_DISPATCH(ER_K_F16S32, g, 1, st);
    return hipGetLastError();
}

inline hipError_t launch_gemm(const GemmArgs& g, int batch, hipStream_t st) {
    ER_GEMM_DISPATCH(ER_K_F32, g, batch, st);
    return hipGetLastError();
}

}  // namespace er
