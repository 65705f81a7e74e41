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
    // GEPI_PLAIN, optional: the same value rounded to fp16 into c16 [M][ldc16] - the A operand of the NEXT fp16 GEMM
    // (gemm_hh_mfma_kernel reads fp16 activations straight into LDS), written by the kernel that produces it
    _Float16* c16;
    int ldc16;
    // split-activation form of the LDS-DMA kernel (fast-mode prefill): A = a_hi (through g.A) + a_lo, both fp16 [M][lda]
    const _Float16* a_lo;
    // LDS-DMA kernel, optional: columns >= vt_col0 (the V third of a fused q/k/v projection, 64 columns per head) are NOT written to
    // c16 but TRANSPOSED per head into vt16 [M / vt_rows][heads][64][vt_ld], keys in fa_vt_pos order - the V^T operand of
    // flash_attn_hh_kernel, written by the GEMM that produces V instead of a transpose pass over it.  M % 64 == 0, vt_rows % 64 == 0.
    _Float16* vt16;
    int vt_col0, vt_rows, vt_ld;
};

constexpr int GBK = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// epilogue shared by the fp32 and fp16-input kernels: C/D fragment map of the 32x32 MFMA (dtype independent):
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Everything that depends on the column only (bias, the q/k/v split of GEPI_QKV) is resolved once per 32-column block, and the
// row -> batch maps (adaLN gate row, shared residual table, cache row of GEPI_QKV) without a division per element: a wave's 32
// rows cross at most one batch boundary when the batch has >= 32 rows (otherwise the generic division path runs).  Round 2
// evaluated five integer divisions and every option per ELEMENT in a 64-fold unrolled body: ~100 KB of code per kernel, which the
// instruction cache could not hold - a fixed ~19 us per workgroup, i.e. the whole time of a K = 1024 GEMM
// (profiles/r03_gemm_hh_probe.log).
struct RowBatch {          // batch index (row / period) and row inside the batch for the rows [row0, row0 + span) (span 32: one MFMA block)
    int q0, next, period;
    bool fast;
    __device__ __forceinline__ RowBatch(int row0, int period_, int span = 32) : period(period_) {
        fast = period_ >= span;
        q0 = period_ > 0 ? row0 / period_ : 0;
        next = (q0 + 1) * period_;
    }
    __device__ __forceinline__ int batch(int gm) const { return fast ? q0 + (gm >= next ? 1 : 0) : gm / period; }
    __device__ __forceinline__ int inner(int gm) const { return gm - batch(gm) * period; }
};

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, float* C, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                              int wn, int kh, int li) {
    const bool has_div = g.div != 0.f, has_gate = g.gate != nullptr, has_resid = g.resid != nullptr, qkv = g.epi != GEPI_PLAIN;
#ifdef ER_GEMM_PROBE_NO_EPILOGUE      // scripts/probes/gemm_hh_probe.hip: what the k-loop costs without the stores
    if (acc[0][0][0] != 12345.678f) return;
#endif
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wn * 32 * TN + j * 32 + li;
        if (gn >= g.N) continue;
        const float bias = g.bias ? g.bias[gn] : 0.f;
        // GEPI_QKV: column -> (q | k | v, head, dim)
        int which = 0, qc = 0;
        long long kv_col = 0;
        if (qkv) {
            which = gn / g.hidden;
            qc = gn - which * g.hidden;
            const int h = qc / g.head_dim, d = qc - h * g.head_dim;
            kv_col = (long long)h * g.l_cap * g.head_dim + d;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = m0 + wm * 32 * TM + i * 32;
            const RowBatch gb(row0, has_gate ? g.gate_rows : 0), rb(row0, (has_resid && g.resid_mod > 0) ? g.resid_mod : 0),
                           sb(row0, qkv ? g.S : 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = row0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (gm >= g.M) continue;
                float v = acc[i][j][r];
                if (has_div) v = v / g.div;
                v += bias;
                if (g.relu) v = fmaxf(v, 0.f);
                if (has_gate) v *= g.gate[(long long)gb.batch(gm) * g.gate_bstride + gn];
                if (has_resid) v += g.resid[(long long)(g.resid_mod > 0 ? rb.inner(gm) : gm) * g.ldr + gn];
                if (!qkv) {
                    C[(long long)gm * g.ldc + gn] = v;
                    if (g.c16) g.c16[(long long)gm * g.ldc16 + gn] = (_Float16)v;
                } else if (which == 0) {
                    g.q[(long long)gm * g.hidden + qc] = v;
                } else {
                    float* cache = (which == 1) ? g.kcache : g.vcache;
                    cache[(long long)sb.batch(gm) * g.kv_bstride + kv_col + (long long)sb.inner(gm) * g.head_dim] = v;
                }
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
    // gfx950 only (the build targets nothing else, edgerunner_amd/build.py): the 128 x 128 tile's two stages are 66.8 KB of static LDS -
    // above the 64 KB a workgroup may hold on gfx90a / gfx942, inside the 160 KB of a gfx950 CU, where they leave room for TWO
    // workgroups per CU (gemm_pick_tile's "three per CU" counts workgroups per CU of the GRID, not resident ones)
    static_assert(sizeof(float) * 2 * GBK * (GLT + GLD) <= 80 * 1024, "two workgroups of this tile must fit the 160 KB LDS of a gfx950 CU");
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

// ---------------------------------------------------------------------------------------------------
// fp16 x fp16 variant with BOTH operands brought in by LDS-DMA (global_load_lds_dwordx4): for the compute-bound front-end GEMMs
// whose activations already exist in fp16 (the producing kernel - adaLN-modulated LayerNorm, attention, GEGLU, a GEMM epilogue -
// writes an fp16 copy of what it hands to the next Linear, rounding exactly where gemm_f16_mfma_kernel rounds on its way in, so
// the results are bit-identical to that kernel's).  Round 2's kernel staged both operands through registers (fp32 A converted in
// the loop, one LDS stage, two barriers per 32-deep k-tile): 19-25 % MFMA-busy with a third of the LDS cycles lost to bank conflicts
// (profiles/r02_pmc_sq_prefill_fp16.json).  Here:
//   * k-tile 64, two LDS stages, ONE barrier per k-tile; no staging registers, no conversion, no ds_write at all;
//   * the LDS image of a tile row is its 128 bytes (8 chunks of 16 B) with chunk c stored at slot c ^ ((row >> 1) & 7).  LDS-DMA
//     writes lane-linearly (wave-uniform base + 16 B x lane), so the permutation is applied to the per-lane GLOBAL address (the 8
//     lanes of a row still read one 128-byte line); the fragment reads (lane l: row l & 31, chunk 2 ks + (l >> 5)) then hit 16
//     different 16-byte bank groups per 16-lane group: conflict-free ds_read_b128;
//   * XCD-aware tile order: the 1-D grid is remapped so that each XCD's L2 sees a contiguous run of tiles (neighbouring tiles
//     share their A row panel) instead of every eighth one (profiles/r02_pmc_hbm_summary.json: 6.4x operand re-fetch).
// Requires K % 64 == 0, lda / ldb multiples of 8 halves, 16-byte aligned operands.  Rows beyond M / N are clamped on load and
// dropped by the epilogue.
constexpr int XBK = 64;
typedef __attribute__((address_space(1))) const void* er_gptr;
typedef __attribute__((address_space(3))) void* er_lptr;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// Epilogue of the LDS-DMA kernel.  The MFMA accumulator holds a COLUMN per lane (4 consecutive rows per register quad), so storing
// it directly costs 64 four-byte stores per lane: measured store-issue-bound - 58 of the 68 us of a 4096 x 8192 x 64 product, i.e.
// 134 MB at 2.3 TB/s, and ~25 us of fixed cost per workgroup at every K (profiles/r03_gemm_hh_probe.log).  Here every wave
// transposes its (32 TM) x (32 TN) block through its own slice of the (now idle) stage buffers and finishes ROW-wise: 16 bytes per
// lane, 16 lanes per 256-byte row segment, bias / gate / residual as float4 loads.
//   HEPI_PLAIN: C = f(acc) (+ optional fp16 copy c16).
//   HEPI_GEGLU: the B rows were permuted (gemm_hh_geglu_row) so that the wave's columns [0,32) are the GEGLU value part and
//               [32,64) the gate part of the SAME 32 outputs: out16[m][j] = fp16((x + bx) * gelu_erf(gt + bg))
//               (core/transformer/dit.py FeedForward / point.py:68-71), the [M][2F] pre-activation never reaches HBM.
enum { HEPI_PLAIN = 0, HEPI_GEGLU = 1 };

// row of the ORIGINAL [2F][K] GEGLU weight that sits at row p of the permuted copy (tile = 128 columns = 4 blocks of 32:
// {value j0..j0+31, gate j0..j0+31, value j0+32.., gate j0+32..}); host and device agree through this one function
__host__ __device__ inline int gemm_hh_geglu_row(int p, int F) {
    const int tile = p >> 7, local = p & 127, wn = local >> 6, blk = (local >> 5) & 1, l = local & 31;
    const int j = tile * 64 + wn * 32 + l;
    return blk == 0 ? j : F + j;
}

// Operands of the plain row-wise epilogue a caller already holds in registers (k_gemm_stream.h requests them when a tile STARTS, so
// they arrive under the k-loop): bias at the lane's four columns, the gate rows of the two batches the wave's rows can lie in, the
// residual float4 of every pass of the call.  Loaded with exactly the lane map and the `vec` condition of hh_epi_rows (hh_epi_vec).
struct HhEpiPre {
    f32x4 bias, gate0, gate1;
    int gb0, gb1;
    f32x4 res[8];          // res[t]: the t-th row pass the call handles (at most eight)
};
__device__ __forceinline__ bool hh_epi_vec(const GemmArgs& g, int gn) {
    return gn + 3 < g.N && (!g.C || !(g.ldc & 3)) && (!g.resid || !(g.ldr & 3)) && !(g.gate_bstride & 3) && (!g.c16 || !(g.ldc16 & 3));
}

// erf for the FUSED GEGLU epilogue of the fp16 front-end mode (round 6): Abramowitz & Stegun 7.1.26 - one reciprocal, one hardware exp2,
// five FMAs - instead of OCML's erff (~42 instructions per element: the epilogue is VALU-bound on it, 64 calls per lane and 256 x 256
// tile).  |error| <= 1.5e-7 for the formula (+ ~1 ulp each from v_rcp_f32 / v_exp_f32): three orders of magnitude below the fp16
// rounding of the value it produces, so the fp16 output differs from the erff form only where a value sits on a rounding boundary
// (tests/test_gpu_kernels.py::test_geglu_epilogue_erf_accuracy: for gate values >= -3 at most one fp16 ulp off the correctly rounded
// gelu, on 0.26 % of a dense grid; below -3, where any float32 erf loses its digits in 1 + erf, absolute error < 3e-6).  Guided DiT forward
// 7.60 -> 7.46 ms in situ (profiles/r06_dit_fast_erf_in_situ.log).  The exact (fp32) front-end keeps erff (k_rowops.h geglu_kernel).
__device__ __forceinline__ float erf_as7126(float v) {
    const float a = fabsf(v), t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    const float p = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    return copysignf(1.0f - p * __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f), v);
}

// accumulators -> the wave's (32 TM) x (32 TN) float slice of LDS, row-major
template <int TM, int TN>
__device__ __forceinline__ void hh_epi_stage(float* sw, const f32x16 (&acc)[TM][TN], int lane) {
    constexpr int WC = 32 * TN;
    const int kh = lane >> 5, li = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sw[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh) * WC + 32 * j + li] = acc[i][j][r];
}

// the row-wise part: LDS slice -> bias / gate / residual / stores.  PARTS = 2: this call handles half `part` (0 / 1) of the slice's row
// passes (keys for the V^T form) - the streamed kernel's loader waves take the second half of every matrix wave's block.
// `pre_` + use_pre (optional): the operands of THIS call's passes, already in registers (res[t] = the t-th pass of this part); the flag
// is separate from the pointer so that the struct stays in registers (a pointer that may be null at run time sends it to scratch).
template <int TM, int TN, int HEPI, int PARTS = 1>
__device__ __forceinline__ void hh_epi_rows(const GemmArgs& g, const float* sw, int mw, int nw, int lane, int part = 0, const HhEpiPre* pre_ = nullptr,
                                            bool use_pre = false) {
    constexpr int WR = 32 * TM, WC = 32 * TN;
    static_assert(HEPI == HEPI_PLAIN || TN == 2, "GEGLU pairs the wave's two 32-column blocks");
    if constexpr (HEPI == HEPI_PLAIN) {
        if (nw >= g.N) return;                // a wave whose whole column block lies beyond N (256-wide tiles on N % 256 != 0)
        if (g.vt16 && nw >= g.vt_col0) {      // wave-uniform: the V columns of this wave's block go out as V^T rows
            // lane -> (column d of the block, a run of WR / LPD keys): 16 keys per step = two 16-byte chunks of the V^T row in
            // fa_vt_pos order ({0-3, 8-11}, {4-7, 12-15}); the column walk down the LDS rows is conflict-free up to the two lanes
            // that share a bank when the block is 64 columns wide
            constexpr int LPD = 64 / WC, KPL = WR / LPD, NQ = KPL / 16;
            static_assert(KPL % 16 == 0 && NQ % PARTS == 0, "whole 16-key groups per lane and part");
            const int d = lane % WC, kk0 = (lane / WC) * KPL;
            const int col = nw - g.vt_col0 + d, hh = col >> 6, dd = col & 63;
            const int heads = (g.N - g.vt_col0) >> 6;
            const float bias = g.bias ? g.bias[nw + d] : 0.f;
            const int gm0 = mw + kk0, b = gm0 / g.vt_rows, key0 = gm0 - b * g.vt_rows;      // KPL | 64 | vt_rows: the run stays in one batch
            if (gm0 >= g.M) return;
            _Float16* dst = g.vt16 + (((long long)b * heads + hh) * 64 + dd) * g.vt_ld + key0;
#pragma unroll
            for (int q_ = 0; q_ < NQ / PARTS; ++q_) {
                const int q = q_ + part * (NQ / PARTS);
                float e[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) e[t] = sw[(kk0 + 16 * q + t) * WC + d] + bias;
                const h16x8 c0 = {(_Float16)e[0], (_Float16)e[1], (_Float16)e[2], (_Float16)e[3], (_Float16)e[8], (_Float16)e[9], (_Float16)e[10], (_Float16)e[11]};
                const h16x8 c1 = {(_Float16)e[4], (_Float16)e[5], (_Float16)e[6], (_Float16)e[7], (_Float16)e[12], (_Float16)e[13], (_Float16)e[14], (_Float16)e[15]};
                *reinterpret_cast<h16x8*>(dst + 16 * q) = c0;
                *reinterpret_cast<h16x8*>(dst + 16 * q + 8) = c1;
            }
            return;
        }
        constexpr int LPR = WC / 4, RPP = 64 / LPR;              // lanes per row, rows per pass
        const int c4 = lane % LPR, rr0 = lane / LPR;
        const int gn = nw + 4 * c4;
        if (gn >= g.N) return;
        const bool vec = hh_epi_vec(g, gn);
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (use_pre) bias = pre_->bias;
        else if (g.bias) {
            bias.x = g.bias[gn];
            if (gn + 1 < g.N) bias.y = g.bias[gn + 1];
            if (gn + 2 < g.N) bias.z = g.bias[gn + 2];
            if (gn + 3 < g.N) bias.w = g.bias[gn + 3];
        }
        const RowBatch gb(mw, g.gate ? g.gate_rows : 0, WR), rb(mw, (g.resid && g.resid_mod > 0) ? g.resid_mod : 0, WR);
        // 32-row blocks (TM = 1: the N = 1024 residual products of the DiT): the residual operands of all eight passes are requested before
        // the first is used (round 6) - with the loads inside a 4-fold unrolled pass loop a wave had 4 x 1 KB in flight, 16-32 KB per CU; in
        // situ 32.8 -> 32.0 us mean over the four products (profiles/r06_dit_trace_ab.log).  64-row blocks (TM = 2: q/k/v, the 256 x 256 tile)
        // keep the 4-fold unrolled loop: sixteen unrolled passes cost the q/k/v product 41.7 -> 47.5 us (same log; the round-3 lesson about
        // unrolled per-element option code and the instruction cache).  The gate row depends on the batch only: the wave's rows span at
        // most two batches (RowBatch fast path), one float4 each.
        constexpr int NPASS = WR / RPP / PARTS, CH = NPASS < 8 ? NPASS : 8;
        static_assert((WR / RPP) % PARTS == 0, "whole passes per part");
        const int t0 = part * NPASS;
        f32x4 gate0 = {1.f, 1.f, 1.f, 1.f}, gate1 = gate0;
        int gb0 = 0, gb1 = 0;
        if (use_pre) { gate0 = pre_->gate0; gate1 = pre_->gate1; gb0 = pre_->gb0; gb1 = pre_->gb1; }
        else if (g.gate && vec) {
            gb0 = gb.batch(mw);
            gb1 = gb.batch(min(mw + WR - 1, g.M - 1));
            gate0 = *reinterpret_cast<const f32x4*>(g.gate + (long long)gb0 * g.gate_bstride + gn);
            gate1 = *reinterpret_cast<const f32x4*>(g.gate + (long long)gb1 * g.gate_bstride + gn);
        }
        const bool gate_pair = g.gate && vec && gb.fast;       // every row of the wave is in batch gb0 or gb1
        if constexpr (NPASS > 8) {
#pragma unroll 4
            for (int t_ = 0; t_ < NPASS; ++t_) {
                const int rr = rr0 + RPP * (t0 + t_), gm = mw + rr;
                if (gm >= g.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(sw + rr * WC + 4 * c4);
                if (g.div != 0.f) { v.x = v.x / g.div; v.y = v.y / g.div; v.z = v.z / g.div; v.w = v.w / g.div; }
                v += bias;
                if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                const long long rrow = g.resid ? (long long)(g.resid_mod > 0 ? rb.inner(gm) : gm) * g.ldr + gn : 0;
                const long long grow = g.gate ? (long long)gb.batch(gm) * g.gate_bstride + gn : 0;
                if (vec) {
                    if (g.gate) v *= gate_pair ? (gb.batch(gm) == gb0 ? gate0 : gate1) : *reinterpret_cast<const f32x4*>(g.gate + grow);
                    if (g.resid) v += *reinterpret_cast<const f32x4*>(g.resid + rrow);
                    if (g.C) *reinterpret_cast<f32x4*>(g.C + (long long)gm * g.ldc + gn) = v;
                    if (g.c16) *reinterpret_cast<h16x4*>(g.c16 + (long long)gm * g.ldc16 + gn) = (h16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (gn + u >= g.N) break;
                        float w = e[u];
                        if (g.gate) w *= g.gate[grow + u];
                        if (g.resid) w += g.resid[rrow + u];
                        if (g.C) g.C[(long long)gm * g.ldc + gn + u] = w;
                        if (g.c16) g.c16[(long long)gm * g.ldc16 + gn + u] = (_Float16)w;
                    }
                }
            }
        } else {
#pragma unroll
            for (int c0 = 0; c0 < NPASS; c0 += CH) {
                f32x4 rres[CH];
                if (NPASS <= 8 && use_pre) {                       // (HhEpiPre holds at most eight passes: 32-row blocks only)
#pragma unroll
                    for (int t = 0; t < CH; ++t) rres[t] = pre_->res[(c0 + t) & 7];
                } else if (g.resid && vec) {
#pragma unroll
                    for (int t = 0; t < CH; ++t) {
                        const int gm = mw + rr0 + RPP * (t0 + c0 + t);
                        const int rrow_ = g.resid_mod > 0 ? rb.inner(min(gm, g.M - 1)) : min(gm, g.M - 1);
                        rres[t] = *reinterpret_cast<const f32x4*>(g.resid + (long long)rrow_ * g.ldr + gn);
                    }
                }
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    const int rr = rr0 + RPP * (t0 + c0 + t), gm = mw + rr;
                    if (gm >= g.M) continue;
                    f32x4 v = *reinterpret_cast<const f32x4*>(sw + rr * WC + 4 * c4);
                    if (g.div != 0.f) { v.x = v.x / g.div; v.y = v.y / g.div; v.z = v.z / g.div; v.w = v.w / g.div; }
                    v += bias;
                    if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    const long long rrow = g.resid ? (long long)(g.resid_mod > 0 ? rb.inner(gm) : gm) * g.ldr + gn : 0;
                    const long long grow = g.gate ? (long long)gb.batch(gm) * g.gate_bstride + gn : 0;
                    if (vec) {
                        if (g.gate) v *= gate_pair ? (gb.batch(gm) == gb0 ? gate0 : gate1) : *reinterpret_cast<const f32x4*>(g.gate + grow);
                        if (g.resid) v += rres[t];
                        if (g.C) *reinterpret_cast<f32x4*>(g.C + (long long)gm * g.ldc + gn) = v;     // null: only the fp16 copy is wanted
                        if (g.c16) *reinterpret_cast<h16x4*>(g.c16 + (long long)gm * g.ldc16 + gn) = (h16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (gn + u >= g.N) break;
                            float w = e[u];
                            if (g.gate) w *= g.gate[grow + u];
                            if (g.resid) w += g.resid[rrow + u];
                            if (g.C) g.C[(long long)gm * g.ldc + gn + u] = w;
                            if (g.c16) g.c16[(long long)gm * g.ldc16 + gn + u] = (_Float16)w;
                        }
                    }
                }
            }
        }
    } else {
        // GEGLU: lane -> (row, 4 outputs): value part at columns 4 c .. +3, gate part at 32 + 4 c .. +3 of the wave's block
        constexpr int LPR = 8, RPP = 8, NPASS = WR / RPP / PARTS;
        static_assert((WR / RPP) % PARTS == 0, "whole passes per part");
        const int c = lane % LPR, rr0 = lane / LPR;
        const int F = g.N >> 1;                                   // outputs per row
        const int jo = (nw >> 1) + 4 * c;                         // output column: permuted column nw + l <-> output nw / 2 + l
        const f32x4 bx = *reinterpret_cast<const f32x4*>(g.bias + nw + 4 * c), bg = *reinterpret_cast<const f32x4*>(g.bias + nw + 32 + 4 * c);
#pragma unroll 4
        for (int t_ = 0; t_ < NPASS; ++t_) {
            const int rr = rr0 + RPP * (t_ + part * NPASS), gm = mw + rr;
            if (gm >= g.M) continue;
            f32x4 x = *reinterpret_cast<const f32x4*>(sw + rr * WC + 4 * c);
            f32x4 gt = *reinterpret_cast<const f32x4*>(sw + rr * WC + 32 + 4 * c);
            x += bx;
            gt += bg;
            const float o0 = x.x * (gt.x * 0.5f * (1.0f + erf_as7126(gt.x * 0.70710678118654752440f)));
            const float o1 = x.y * (gt.y * 0.5f * (1.0f + erf_as7126(gt.y * 0.70710678118654752440f)));
            const float o2 = x.z * (gt.z * 0.5f * (1.0f + erf_as7126(gt.z * 0.70710678118654752440f)));
            const float o3 = x.w * (gt.w * 0.5f * (1.0f + erf_as7126(gt.w * 0.70710678118654752440f)));
            *reinterpret_cast<h16x4*>(g.c16 + (long long)gm * F + jo) = (h16x4){(_Float16)o0, (_Float16)o1, (_Float16)o2, (_Float16)o3};
        }
    }
}

template <int TM, int TN, int HEPI>
__device__ __forceinline__ void gemm_hh_epilogue(const GemmArgs& g, float* sw, const f32x16 (&acc)[TM][TN], int mw, int nw, int lane) {
#ifdef ER_GEMM_PROBE_NO_EPILOGUE
    if (acc[0][0][0] != 12345.678f) return;
#endif
    hh_epi_stage<TM, TN>(sw, acc, lane);
    // (same wave writes and reads: the LDS queue is in order, no barrier)
    hh_epi_rows<TM, TN, HEPI>(g, sw, mw, nw, lane);
}

// SPLIT: the A operand is hi + lo (two fp16 arrays: a = fp16(x), lo = fp16(x - hi), |x - hi - lo| <= 2^-22 |x|), every weight fragment
// is multiplied by both (the lo products first), i.e. the fp16-weight x fp32-activation product to fp32 round-off - what
// gemm_f16s_mfma_kernel computes from an fp32 A with a register-staged split; same MFMA order per accumulator -> same bits.
template <int TM, int TN, int HEPI = HEPI_PLAIN, bool SPLIT = false>
__global__ __launch_bounds__(ER_WG) void gemm_hh_mfma_kernel(GemmArgs g, int ntx) {
    constexpr int GBM = 64 * TM, GBN = 64 * TN, AROWS = SPLIT ? 2 * GBM : GBM;
    constexpr int STAGE = (AROWS + GBN) * XBK;                                   // halves per stage: A rows (hi, then lo), then B rows
    constexpr int NAI = GBM / 32, NBI = GBN / 32;                                // 8-row LDS-DMA pieces per wave and operand
    static_assert(2 * STAGE * 2 <= 65536, "two stages fit the static LDS limit");
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];             // the ONLY LDS object (a second one makes hipcc drain vmcnt per k-step)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware tile order (bijective for any grid size)
    const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + within;
    const int ty = lin / ntx, tx = lin - ty * ntx;
    const int m0 = ty * GBM, n0 = tx * GBN;
    const _Float16* A = reinterpret_cast<const _Float16*>(g.A);
    const _Float16* B = reinterpret_cast<const _Float16*>(g.B);
    const int nk = g.K / XBK;

    // per-lane source pointers of this wave's LDS-DMA pieces: piece i covers tile rows 8i .. 8i+7, lane -> (row 8i + lane/8, slot lane%8)
    const int lrow = lane >> 3, lslot = lane & 7;
    const _Float16* pa[NAI];
    const _Float16* pl[SPLIT ? NAI : 1];
    const _Float16* pb[NBI];
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
        const int r = 8 * (wid * NAI + j) + lrow;
        const long long off = (long long)min(m0 + r, g.M - 1) * g.lda + ((lslot ^ ((r >> 1) & 7)) << 3);
        pa[j] = A + off;
        if (SPLIT) pl[j] = g.a_lo + off;
    }
#pragma unroll
    for (int j = 0; j < NBI; ++j) {
        const int r = 8 * (wid * NBI + j) + lrow;
        pb[j] = B + (long long)min(n0 + r, g.N - 1) * g.ldb + ((lslot ^ ((r >> 1) & 7)) << 3);
    }
    auto issue = [&](int kt, int s) {
        _Float16* as = lds + s * STAGE;
        _Float16* bs = as + AROWS * XBK;
#pragma unroll
        for (int j = 0; j < NAI; ++j)
            __builtin_amdgcn_global_load_lds((er_gptr)(pa[j] + kt * XBK), (er_lptr)(as + 8 * (wid * NAI + j) * XBK), 16, 0, 0);
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < NAI; ++j)
                __builtin_amdgcn_global_load_lds((er_gptr)(pl[j] + kt * XBK), (er_lptr)(as + (GBM + 8 * (wid * NAI + j)) * XBK), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NBI; ++j)
            __builtin_amdgcn_global_load_lds((er_gptr)(pb[j] + kt * XBK), (er_lptr)(bs + 8 * (wid * NBI + j) * XBK), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kh = lane >> 5, li = lane & 31, swz = (li >> 1) & 7;
    if (nk > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);      // lands while this tile is multiplied
        const _Float16* as = lds + cur * STAGE + (wm * 32 * TM + li) * XBK;
        const _Float16* bs = lds + cur * STAGE + AROWS * XBK + (wn * 32 * TN + li) * XBK;
#pragma unroll
        for (int ks = 0; ks < XBK / 16; ++ks) {
            const int co = ((2 * ks + kh) ^ swz) << 3;
            h16x8 av[TM], bv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const h16x8*>(bs + 32 * j * XBK + co);
            if (SPLIT) {      // the small (lo) products first, then the large ones: the accumulator sees them in increasing magnitude
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const h16x8*>(as + (GBM + 32 * i) * XBK + co);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const h16x8*>(as + 32 * i * XBK + co);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile have landed ...
        __syncthreads();                                    // ... and everybody's; everybody is done reading `cur`
    }
    // every wave is past the last barrier: the stage buffers are free.  Wave w's (32 TM) x (32 TN) floats fit its quarter of them
    static_assert(4 * 32 * TM * 32 * TN * 4 <= 2 * STAGE * 2, "epilogue staging fits the stage buffers");
    static_assert(!SPLIT || TM == 1, "split form: 64-row tiles (two A images per stage)");
    float* sw = reinterpret_cast<float*>(lds) + wid * (32 * TM * 32 * TN);
    gemm_hh_epilogue<TM, TN, HEPI>(g, sw, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, lane);
}

// ---------------------------------------------------------------------------------------------------
// The same product on a 256 x 256 workgroup tile: 8 waves (2 x 4), each owning a 128 x 64 block = a 4 x 2 grid of 32 x 32 MFMA
// accumulators (128 registers).  Round 4 measured why the 4-wave kernel above stops at ~850 TFLOP/s: a 64 x 64
// block per wave reads 16 KB of fragments out of LDS per 64-deep k-step for 16 MFMAs, and deeper stages only trade occupancy away;
// a 128 x 64 block reads 24 KB for 32 MFMAs - a third fewer LDS bytes per flop - and a 256 x 256 tile brings half the operand
// bytes per flop through LDS-DMA (cdna_hip_programming.md section 5: the 256^2 tile with two LDS buffers at BK = 64).
// Same 128-byte row images, same XOR swizzle, same fragment reads and the SAME MFMA sequence per accumulator element (k ascending
// in steps of 16) as gemm_hh_mfma_kernel: bit-identical results.  Two stages of 64 KB = 128 KB of LDS, one workgroup per CU; the
// epilogue runs the 64 x 64 row-wise epilogue of the 4-wave kernel twice per wave (rows 0..63, 64..127 of its block) through
// its 16 KB slice of the idle stage buffers.  For the wide products only (launch_gemm_hh / launch_gemm_hh_geglu pick it when it
// fills the chip: N = 8192 feed-forward-in and N = 3072 qkv of the DiT front-end).
// LDS-DMA pieces issued from inline asm (hipcc treats the builtin as an LDS write every later ds_read may alias and parks a
// vmcnt(0) in front of the fragment reads / inside __syncthreads; an asm statement is invisible to that bookkeeping, so the waits are
// the counted ones written in the loop - same helper as k_flash_attn.h fa_glds16).  M0 (the LDS base) is saved and restored.
__device__ __forceinline__ void gm_glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
constexpr int X256_BM = 256, X256_BN = 256, X256_THREADS = 512;
template <int HEPI = HEPI_PLAIN>
__global__ __launch_bounds__(X256_THREADS) void gemm_hh256_kernel(GemmArgs g, int ntx) {
    constexpr int STAGE = (X256_BM + X256_BN) * XBK;                              // halves per stage: 256 A rows, then 256 B rows
    constexpr int NPI = X256_BM / 8 / 8;                                          // 8-row LDS-DMA pieces per wave and operand (4)
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];              // 128 KB: the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + within;
    const int ty = lin / ntx, tx = lin - ty * ntx;
    const int m0 = ty * X256_BM, n0 = tx * X256_BN;
    const _Float16* A = reinterpret_cast<const _Float16*>(g.A);
    const _Float16* B = reinterpret_cast<const _Float16*>(g.B);
    const int nk = g.K / XBK;

    const int lrow = lane >> 3, lslot = lane & 7;
    const _Float16* pa[NPI];
    const _Float16* pb[NPI];
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int r = 8 * (wid * NPI + j) + lrow;
        pa[j] = A + (long long)min(m0 + r, g.M - 1) * g.lda + ((lslot ^ ((r >> 1) & 7)) << 3);
        pb[j] = B + (long long)min(n0 + r, g.N - 1) * g.ldb + ((lslot ^ ((r >> 1) & 7)) << 3);
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(er_lptr)lds;              // LDS byte address of the array
    auto issue = [&](int kt, int s) {
        const unsigned as = lds0 + (unsigned)(s * STAGE) * 2u, bs = as + X256_BM * XBK * 2u;
#pragma unroll
        for (int j = 0; j < NPI; ++j) gm_glds16(pa[j] + kt * XBK, as + 8u * (wid * NPI + j) * XBK * 2u);
#pragma unroll
        for (int j = 0; j < NPI; ++j) gm_glds16(pb[j] + kt * XBK, bs + 8u * (wid * NPI + j) * XBK * 2u);
    };
    // (an L2 prefetch of the k-tile two steps ahead - one 4-byte LDS-DMA per lane and line into a dump area, left out of the counted
    // wait - measured SLOWER: 942 vs 990 TFLOP/s at 4096^3, 910 vs 1045 at 8192^3; profiles/r05_gemm_hh256_probe_with_l2_prefetch.log)
    constexpr int TM = 4, TN = 2;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kh = lane >> 5, li = lane & 31, swz = (li >> 1) & 7;
    if (nk > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const _Float16* as = lds + cur * STAGE + (wm * 32 * TM + li) * XBK;
        const _Float16* bs = lds + cur * STAGE + X256_BM * XBK + (wn * 32 * TN + li) * XBK;
        // fragments double-buffered in registers: the six reads of k-step ks + 1 are in flight under the eight MFMAs of k-step ks
        // (left to itself hipcc re-used one register set and put an lgkmcnt(0) in front of every group of four MFMAs)
        h16x8 av[2][TM], bv[2][TN];
        auto frags = [&](int ks, int buf) {
            const int co = ((2 * ks + kh) ^ swz) << 3;
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[buf][j] = *reinterpret_cast<const h16x8*>(bs + 32 * j * XBK + co);
#pragma unroll
            for (int i = 0; i < TM; ++i) av[buf][i] = *reinterpret_cast<const h16x8*>(as + 32 * i * XBK + co);
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < XBK / 16; ++ks) {
            if (ks + 1 < XBK / 16) frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // this wave's pieces of the next tile have landed, every fragment read of `cur` has returned (its MFMAs are issued) - then
        // the barrier: everybody's pieces are in, everybody is done reading `cur`
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // every wave is past the last barrier: the stage buffers are free; wave w's 64 x 64 floats (16 KB) fit its eighth of them
    static_assert(8 * 64 * 64 * 4 <= 2 * STAGE * 2, "epilogue staging fits the stage buffers");
    float* sw = reinterpret_cast<float*>(lds) + wid * (64 * 64);
    const int mw = m0 + wm * 32 * TM, nw = n0 + wn * 32 * TN;
    gemm_hh_epilogue<2, 2, HEPI>(g, sw, *reinterpret_cast<const f32x16(*)[2][TN]>(&acc[0]), mw, nw, lane);
    gemm_hh_epilogue<2, 2, HEPI>(g, sw, *reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2]), mw + 64, nw, lane);
}

// fp32 -> fp16 copy of a row-major matrix (the A operand of gemm_hh_mfma_kernel when no producer wrote one)
__global__ __launch_bounds__(ER_WG) void cvt_rows_f16_kernel(const float* x, _Float16* y, long long rows, int cols, int ldx, int ldy) {
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * ER_WG + threadIdx.x; i < total; i += (long long)gridDim.x * ER_WG) {
        const long long r = i / cols;
        const int c = (int)(i - r * cols);
        y[r * ldy + c] = (_Float16)x[r * ldx + c];
    }
}

__global__ __launch_bounds__(ER_WG) void cvt_f16_rows_f32_kernel(const _Float16* x, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * ER_WG + threadIdx.x; i < n; i += (long long)gridDim.x * ER_WG) y[i] = (float)x[i];
}

// ---------------------------------------------------------------------------------------------------
// Exact fp32 GEMM with both operands by LDS-DMA (round 4): C = epilogue(A[M,K] . W[N,K]^T), v_mfma_f32_32x32x2_f32, for the plain
// (NT, unbatched, K % 32 == 0) Linears of the exact-mode prefill and point encoder.  gemm_f32_mfma_kernel above stages a k-tile through
// registers and scatters it k-major into LDS (8 + 8 four-byte ds_write per thread and tile, one wave per SIMD busy with that while the
// matrix pipe idles): 0.53 of the fp32 matrix peak on the prefill shapes.  A 32-float row of a k-tile is 128 bytes - exactly the row
// image of the fp16 LDS-DMA kernel (gemm_hh_mfma_kernel): 8 chunks of 16 B, chunk c of row r stored at slot c ^ ((r >> 1) & 7), written
// by `global_load_lds_dwordx4` with no registers and no ds_write.  The fp32 matrix core is 16x slower than the fp16 one, so a 32-deep
// k-tile is 4096 MFMA cycles per wave and the next tile's DMA lands long before it is needed: two stages, one barrier per k-tile.
// Operand fetch: ONE ds_read_b32 per MFMA and operand block, lane (row l & 31, k = 2 kk + (l >> 5)) - the same (lane half -> k) assignment
// and the same k order per accumulator as gemm_f32_mfma_kernel: BIT-IDENTICAL results (tests/test_gpu_kernels.py::test_gemm_f32_lds_dma).
template <int TM, int TN>
__global__ __launch_bounds__(ER_WG) void gemm_f32d_mfma_kernel(GemmArgs g, int ntx) {
    constexpr int GBM = 64 * TM, GBN = 64 * TN, FBK = 32;                        // k-tile: 32 floats = one 128-byte row image
    constexpr int STAGE = (GBM + GBN) * FBK;                                     // floats per stage
    constexpr int NAI = GBM / 32, NBI = GBN / 32;                                // 8-row LDS-DMA pieces per wave and operand
    static_assert(2 * STAGE * 4 <= 65536, "two stages fit the static LDS limit");
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];                // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + within;      // XCD-aware tile order (bijective)
    const int ty = lin / ntx, tx = lin - ty * ntx;
    const int m0 = ty * GBM, n0 = tx * GBN;
    const int nk = g.K / FBK;
    // (Measured and not kept, profiles/r04_gemm_f32d_ab.log: skipping the MFMAs of 32-row blocks that lie outside the matrix and running
    // the ragged last row tile first - the prefill's 2050 rows are 16 full 128-row tiles + 2 rows, i.e. 3.19 tiles per CU - made encode +
    // prefill SLOWER on the same box, 42.7 -> 44.9 ms.)
    const int lrow = lane >> 3, lslot = lane & 7;
    const float* pa[NAI];
    const float* pb[NBI];
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
        const int r = 8 * (wid * NAI + j) + lrow;
        pa[j] = g.A + (long long)min(m0 + r, g.M - 1) * g.lda + ((lslot ^ ((r >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int j = 0; j < NBI; ++j) {
        const int r = 8 * (wid * NBI + j) + lrow;
        pb[j] = g.B + (long long)min(n0 + r, g.N - 1) * g.ldb + ((lslot ^ ((r >> 1) & 7)) << 2);
    }
    auto issue = [&](int kt, int s) {
        float* as = lds + s * STAGE;
        float* bs = as + GBM * FBK;
#pragma unroll
        for (int j = 0; j < NAI; ++j)
            __builtin_amdgcn_global_load_lds((er_gptr)(pa[j] + kt * FBK), (er_lptr)(as + 8 * (wid * NAI + j) * FBK), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NBI; ++j)
            __builtin_amdgcn_global_load_lds((er_gptr)(pb[j] + kt * FBK), (er_lptr)(bs + 8 * (wid * NBI + j) * FBK), 16, 0, 0);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kh = lane >> 5, li = lane & 31;
    // float index of (row, k) inside a tile image: row * 32 + ((k >> 2) ^ ((row >> 1) & 7)) * 4 + (k & 3); the rows this lane reads
    // are wm * 32 TM + 32 i + li: (row >> 1) & 7 = (li >> 1) & 7 for every i (32 i and wm * 32 TM are multiples of 16 rows)
    const int swz = (li >> 1) & 7;
    if (nk > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);      // lands while this tile is multiplied (4096 MFMA cycles per wave)
        const float* as = lds + cur * STAGE + (wm * 32 * TM + li) * FBK;
        const float* bs = lds + cur * STAGE + GBM * FBK + (wn * 32 * TN + li) * FBK;
        // (These ds_read_b32 fetches are 4-way bank-conflicted - SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.75, profiles/r05_pmc_sq_prefill.json -
        // but off the critical path: whole-chunk ds_read_b128 fetches, conflict-free and bit-identical, measured SLOWER, encode + prefill
        // 38.3 -> 39.2 ms same box, profiles/r05_gemm_f32d_b128.log: twice the LDS bytes and two selects per operand beside 64-cycle MFMAs.)
#pragma unroll
        for (int kk = 0; kk < FBK / 2; ++kk) {
            const int k = 2 * kk + kh;                                          // kh is 0 / 1: the chunk index is uniform per lane half
            const int off = (((k >> 2) ^ swz) << 2) + (k & 3);
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = as[32 * i * FBK + off];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = bs[32 * j * FBK + off];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    gemm_epilogue<TM, TN>(g, g.C, acc, m0, n0, wm, wn, kh, li);
}

inline int gemm_pick_tile(int M, int N, int batch);
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
    const char* fv = getenv("ER_GEMM_TILE");            // per call: the unit tests force each shape in one process
    const int forced = fv ? atoi(fv) : 0;
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

// tile choice of the LDS-DMA kernel (see gemm_pick_tile): placeholder rule, tuned by scripts/probes/gemm_hh_probe.hip
// (profiles/r03_gemm_hh_probe.log): 128x128 wants two workgroups per CU (its 64 KB of stages allow no more); below that 64x128
// at >= 1 per CU beats both 128x128 at < 2 per CU and 64x64 (4096 x 1024 x 4096: 51 vs 54 vs 62 us)
// 256 x 256 (8 waves, one workgroup per CU) when its tiles fill the chip (one workgroup per CU and more; at 192 tiles - the DiT's
// qkv product - it measured slower than 128 x 128: 47.6 vs 41.5 us, profiles/r05_gemm_hh256_probe.log):
// ER_GEMM256=0 restores the round-4 rule
inline bool gemm_hh_use_256(int M, int N) {
    const char* v = getenv("ER_GEMM256");                  // read per launch, like ER_GEMM_TILE / ER_FLASH32_KSPLIT: a test or an A/B script may flip it in-process
    const bool off = v && atoi(v) == 0;
    const long long t = (long long)((M + 255) / 256) * ((N + 255) / 256);
    return !off && t >= 256;
}
inline int gemm_hh_pick_tile(int M, int N) {
    auto wgs = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (gemm_hh_use_256(M, N)) return 4;
    if (wgs(128, 128) >= 512) return 1;
    if (wgs(64, 128) >= 256) return 2;
    return 3;
}

// A = fp16 [M][lda], B = fp16 weights [N][ldb] (both through g.A / g.B), K % 64 == 0
// k_gemm_stream.h: the streamed form (loader + matrix waves, five-stage ring, persistent tile stream), bit-identical to the kernels above.
// Rule (round 6, measured IN SITU, profiles/r06_dit_stream_in_situ*.log): the DEEP products whose 128 x 128 tiles fill the chip about once -
// the DiT's feed-forward-out, 4096 x 1024 x 4096: guided forward 8.38 -> 8.13 ms - where its three k-tiles in flight hide the HBM latency
// of operands that are cold in situ (24 layers of weights do not fit the Infinity Cache); on the K = 1024 products and on q/k/v it measured
// slower in situ (8.22 / 8.34 ms), as in the replay probe, and so did its GEGLU form on the feed-forward-in (8.21 vs 7.94 ms,
// r06_dit_stream_geglu_in_situ.log).  ER_GEMM_STREAM=0: never; =2: wherever it is legal (unit tests); read per launch.
inline hipError_t launch_gemm_hh_stream(const GemmArgs& g, hipStream_t st, bool geglu);
inline bool gemm_hh_use_stream(int M, int N, int K) {
    const char* v = getenv("ER_GEMM_STREAM");
    const int mode = v ? atoi(v) : 1;
    if (mode == 0) return false;
    if (mode == 2) return true;
    const long long t = (long long)((M + 127) / 128) * ((N + 127) / 128);
    return K >= 2048 && t >= 128 && t <= 512;
}
inline hipError_t launch_gemm_hh(const GemmArgs& g, hipStream_t st, int force_tile = 0) {
    if (g.K % XBK != 0 || (g.lda & 7) || (g.ldb & 7)) return hipErrorInvalidValue;
    if (!force_tile && gemm_hh_use_stream(g.M, g.N, g.K)) return launch_gemm_hh_stream(g, st, false);
    if (g.vt16 && ((g.M & 63) || (g.vt_rows & 63) || (g.vt_col0 & 127) || ((g.N - g.vt_col0) & 63) || (g.vt_ld & 7) || g.div != 0.f || g.relu ||
                   g.gate || g.resid))
        return hipErrorInvalidValue;
    const int tile = force_tile ? force_tile : gemm_hh_pick_tile(g.M, g.N);
    if (tile == 4) {          // 256 x 256, 8 waves (the V^T epilogue walks 64-row runs: fine, 64 | 256)
        const int ntx4 = (g.N + X256_BN - 1) / X256_BN;
        const dim3 grid4(ntx4 * ((g.M + X256_BM - 1) / X256_BM));
        hipLaunchKernelGGL((gemm_hh256_kernel<HEPI_PLAIN>), grid4, dim3(X256_THREADS), 0, st, g, ntx4);
        return hipGetLastError();
    }
    const int bm = tile == 1 ? 128 : 64, bn = tile == 3 ? 64 : 128;
    const int ntx = (g.N + bn - 1) / bn, nty = (g.M + bm - 1) / bm;
    const dim3 grid(ntx * nty);
    if (tile == 1) hipLaunchKernelGGL((gemm_hh_mfma_kernel<2, 2>), grid, dim3(ER_WG), 0, st, g, ntx);
    else if (tile == 2) hipLaunchKernelGGL((gemm_hh_mfma_kernel<1, 2>), grid, dim3(ER_WG), 0, st, g, ntx);
    else hipLaunchKernelGGL((gemm_hh_mfma_kernel<1, 1>), grid, dim3(ER_WG), 0, st, g, ntx);
    return hipGetLastError();
}

// x (fp32 rows) -> hi = fp16(x), lo = fp16(x - hi): the two A operands of the split form (exactly the split the register-staged
// gemm_f16s_mfma_kernel performs on its way into LDS)
// grid (ceil(cols / 4 / 256), rows' worth of blocks): a row per blockIdx.y (strided), a float4 per thread - no index division
__global__ __launch_bounds__(ER_WG) void split_rows_f16_kernel(const float* x, _Float16* hi, _Float16* lo, long long rows, int cols, int ldx) {
    const int c = (blockIdx.x * ER_WG + threadIdx.x) * 4;
    if (c >= cols) return;
    for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        const h16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        const h16x4 l = {(_Float16)(v.x - (float)h[0]), (_Float16)(v.y - (float)h[1]), (_Float16)(v.z - (float)h[2]), (_Float16)(v.w - (float)h[3])};
        *reinterpret_cast<h16x4*>(hi + r * cols + c) = h;
        *reinterpret_cast<h16x4*>(lo + r * cols + c) = l;
    }
}
inline dim3 split_rows_grid(long long rows, int cols) {
    return dim3((unsigned)((cols / 4 + ER_WG - 1) / ER_WG), (unsigned)(rows < 65535 ? (rows > 0 ? rows : 1) : 65535));
}

// split form: A = a_hi (g.A) + a_lo (g.a_lo), fp16 [M][lda]; 64-row tiles
inline hipError_t launch_gemm_hh_split(const GemmArgs& g, hipStream_t st) {
    if (g.K % XBK != 0 || (g.lda & 7) || (g.ldb & 7) || !g.a_lo) return hipErrorInvalidValue;
    const long long w2 = (long long)((g.M + 63) / 64) * ((g.N + 127) / 128);
    if (w2 >= 256) {
        const int ntx = (g.N + 127) / 128;
        hipLaunchKernelGGL((gemm_hh_mfma_kernel<1, 2, HEPI_PLAIN, true>), dim3(ntx * ((g.M + 63) / 64)), dim3(ER_WG), 0, st, g, ntx);
    } else {
        const int ntx = (g.N + 63) / 64;
        hipLaunchKernelGGL((gemm_hh_mfma_kernel<1, 1, HEPI_PLAIN, true>), dim3(ntx * ((g.M + 63) / 64)), dim3(ER_WG), 0, st, g, ntx);
    }
    return hipGetLastError();
}

// builds the permuted GEGLU operands: wp[p][:] = w[gemm_hh_geglu_row(p, F)][:], bp[p] = b[gemm_hh_geglu_row(p, F)]   (p < 2F)
__global__ __launch_bounds__(ER_WG) void geglu_permute_kernel(const _Float16* w, const float* b, _Float16* wp, float* bp, int F, int K) {
    const int p = blockIdx.x, src = gemm_hh_geglu_row(p, F);
    for (int k = threadIdx.x * 8; k < K; k += ER_WG * 8)
        *reinterpret_cast<h16x8*>(wp + (long long)p * K + k) = *reinterpret_cast<const h16x8*>(w + (long long)src * K + k);
    if (threadIdx.x == 0) bp[p] = b[src];
}

// out16[M][F] = fp16(GEGLU(A . Wp^T + bp)): Wp / bp = the [2F][K] weight / [2F] bias permuted by gemm_hh_geglu_row (g.B, g.bias),
// g.N = 2F (a multiple of 128), g.c16 = out16; no fp32 output
// force_tile (unit tests): 0 = the product rule, 1 = 128 x 128, 2 = 64 x 128 (4 waves), 4 = 256 x 256 (8 waves; N % 256 == 0)
inline hipError_t launch_gemm_hh_geglu(const GemmArgs& g, hipStream_t st, int force_tile = 0) {
    if (g.K % XBK != 0 || (g.lda & 7) || (g.ldb & 7) || (g.N & 127) || !g.c16 || !g.bias) return hipErrorInvalidValue;
    if (force_tile == 4 && (g.N & 255)) return hipErrorInvalidValue;
    if (force_tile == 4 || (force_tile == 0 && (g.N & 255) == 0 && gemm_hh_use_256(g.M, g.N))) {
        const int ntx4 = g.N / 256;
        const dim3 grid4(ntx4 * ((g.M + 255) / 256));
        hipLaunchKernelGGL((gemm_hh256_kernel<HEPI_GEGLU>), grid4, dim3(X256_THREADS), 0, st, g, ntx4);
        return hipGetLastError();
    }
    const int ntx = g.N / 128;
    if (force_tile == 1 || (force_tile == 0 && (long long)((g.M + 127) / 128) * ntx >= 768)) {
        hipLaunchKernelGGL((gemm_hh_mfma_kernel<2, 2, HEPI_GEGLU>), dim3(ntx * ((g.M + 127) / 128)), dim3(ER_WG), 0, st, g, ntx);
    } else {
        hipLaunchKernelGGL((gemm_hh_mfma_kernel<1, 2, HEPI_GEGLU>), dim3(ntx * ((g.M + 63) / 64)), dim3(ER_WG), 0, st, g, ntx);
    }
    return hipGetLastError();
}

inline hipError_t launch_gemm_f16s(const GemmArgs& g, hipStream_t st) {  // NT only, K % 32 == 0, B = fp16 weights, A split hi/lo
    static const bool bk64 = [] { const char* v = getenv("ER_F16S_BK"); return v && atoi(v) == 64; }();
    if (bk64 && g.K % 64 == 0) ER_GEMM_DISPATCH(ER_K_F16S64, g, 1, st);
    else ER_GEMM_DISPATCH(ER_K_F16S32, g, 1, st);
    return hipGetLastError();
}

// the plain NT products (no batch, no causal bound, K-major B) with K % 32 == 0 and 16-byte aligned rows take the LDS-DMA kernel;
// ER_GEMM_F32_DMA=0 keeps the register-staged kernel everywhere (A/B, parity matrix: the two give the same bits)
inline bool gemm_f32_dma_ok(const GemmArgs& g, int batch) {
    const char* e = getenv("ER_GEMM_F32_DMA");       // read per launch (a few dozen launches per prefill): the unit test flips it in-process
    const bool off = e && e[0] == '0';
    return !off && batch == 1 && !g.b_is_kn && !g.causal && g.K % 32 == 0 && !(g.lda & 3) && !(g.ldb & 3) &&
           !((reinterpret_cast<unsigned long long>(g.A) | reinterpret_cast<unsigned long long>(g.B)) & 15);
}
inline hipError_t launch_gemm(const GemmArgs& g, int batch, hipStream_t st) {
    if (gemm_f32_dma_ok(g, batch)) {
        // tile rule of the LDS-DMA kernel: it is MFMA-bound per tile, so what matters is the number of ROUNDS the CUs need - a 2050 x 6144
        // product is 816 tiles of 128 x 128 = 3.19 per CU, i.e. FOUR rounds on 48 CUs, against 1584 tiles of 64 x 128 = 6.19 -> seven half-size
        // rounds (3.5): the big tile is kept only from ER_GEMM_F32D_MIN128 (default 4) tiles per CU on.  Any shape gives the same bits.
        const char* mv = getenv("ER_GEMM_F32D_MIN128");
        const long long min128 = 256LL * (mv ? atoi(mv) : 4);
        int tile = gemm_pick_tile(g.M, g.N, 1);
        if (!getenv("ER_GEMM_TILE") && tile == 1 && (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) < min128) tile = 2;
        const int bm = tile == 1 ? 128 : 64, bn = tile == 3 ? 64 : 128;
        const int ntx = (g.N + bn - 1) / bn, nty = (g.M + bm - 1) / bm;
        if (tile == 1) hipLaunchKernelGGL((gemm_f32d_mfma_kernel<2, 2>), dim3(ntx * nty), dim3(ER_WG), 0, st, g, ntx);
        else if (tile == 2) hipLaunchKernelGGL((gemm_f32d_mfma_kernel<1, 2>), dim3(ntx * nty), dim3(ER_WG), 0, st, g, ntx);
        else hipLaunchKernelGGL((gemm_f32d_mfma_kernel<1, 1>), dim3(ntx * nty), dim3(ER_WG), 0, st, g, ntx);
        return hipGetLastError();
    }
    ER_GEMM_DISPATCH(ER_K_F32, g, batch, st);
    return hipGetLastError();
}

}  // namespace er
