// Batched decode projections on the fp32 matrix cores: out[b][n] = sum_k X[b][k] W[n][k] for up to 32 batch rows per
// pass with the weights streamed ONCE (the VALU kernel in k_gemv.h needs one pass per 16 rows and is FMA-issue-bound
// there: 32 rows x 28 M weights per layer is 1.8 GFLOP of scalar fmaf).  Same callers, prologue (prep_rows_kernel) and
// epilogues as k_gemv.h; replaces the reference's nn.Linear calls at batch > 4
// (core/transformer/modeling_opt.py:185,189-190,232,281,284,497).
//
// Weight layout: the kernel streams a TILED copy of the matrix (tile_weights_kernel, made once when a batch > 4 is first
// reserved): [N/16][K/(4*EPL)] tiles of 1 KiB, inside a tile piece number (kq*16 + i) = the lane that consumes it, so a
// wave-load is one contiguous 1 KiB line group and a wave's whole slice is 6 (fp16: 3) consecutive KiB.  Reading the
// row-major matrix with lane = row (16 rows x 64 B, 6 KiB apart, per instruction) ran at a third of the speed.
//
// HBM-bound: v_mfma_f32_16x16x4_f32 with the 16 weight rows of a tile on the A side and 16 batch rows on the B side.
// Lane (i = lane & 15, kq = lane >> 4) loads 16 bytes of weight row n0 + i - no LDS transpose is needed because the
// matrix core sums over k in any order as long as A and B agree on which k sits in which (lane, element) slot:
// element m of the lane's vector feeds MFMA number m, and the B operand of that MFMA is element m of the matching
// 16-byte piece of X[b = lane & 15].  A workgroup = 16 waves = one 32-row tile x 16 K-slices of 96, every load issued
// before the first MFMA; the 16 partial tiles meet in LDS and are added in slice order (deterministic, independent of
// the other batch rows).  K = 6144 (fc2) is additionally split across workgroups (grid.y = 4); those partials go to
// scratch and splitk_finish_kernel applies the epilogue.
#pragma once
#include "k_gemv.h"

// timeline hooks of scripts/probes/gemv_mfma_timeline_probe.hip (expand to nothing in the library build)
#ifndef ER_TPG
#define ER_TPG(i)
#define ER_TPG_WAIT(n)
#endif

namespace er {

typedef float gm_f4 __attribute__((ext_vector_type(4)));
constexpr int GM_WAVES = 16, GM_THREADS = GM_WAVES * 64, GM_R2 = 2, GM_ROWS = 16 * GM_R2, GM_KW = 96;

// A workgroup = 16 waves = a 32-row tile (two 16-row A tiles per wave, sharing the X registers: X comes from L2 once
// per 32 weight rows) x 16 K-slices of 96 = K-range 1536.  NBH = 16-row batch halves (1 or 2); SPLIT: raw partials.
// XT (fast mode only): a.xin is the tiled hi | lo image of the input (k_gemv.h xt_entry) instead of row-major fp32 - same values, same
// MFMA sequence, bit-identical results; EPI_RELU then writes its output in that layout too (a.xt_out: the next projection's input).
// NWV = waves per workgroup = 96-wide K-slices it reduces through LDS: 16 (K-range 1536, the round-1..3 shape) or 4 (K-range 384:
// four times the workgroups, each with a quarter of the input image to fetch - for the projections whose split-K partials a later
// LayerNorm launch finishes anyway: out_proj and fc2 in the tiled fast-mode path).
template <typename WT, int NBH, int EPI, bool SPLIT, bool XT = false, int NWV = GM_WAVES>
__global__ __launch_bounds__(64 * NWV) void gemv_mfma_kernel(GemvArgs a, int nb_valid, int K, float* part) {
    static_assert(!XT || sizeof(WT) == 2, "the tiled activation layout feeds the fp16 matrix cores");
    static_assert(NWV == 16 || (NWV == 4 && SPLIT), "4-wave workgroups only produce split-K partials");
    constexpr int GM_WAVES = NWV;               // shadows the namespace constant inside this kernel
    constexpr int EPL = WTraits<WT>::EPL;            // weights per 16-byte load: 4 (fp32) or 8 (fp16)
    constexpr int NLD = GM_KW / (4 * EPL);           // loads per lane and row tile: a wave-load covers 16 rows x 4*EPL k
    constexpr int XV = EPL / 4;
    __shared__ __attribute__((aligned(16))) float red[GM_WAVES][GM_R2 * NBH][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    ER_TPG(0);
    const int li = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * GM_ROWS;
    const int kbase = (blockIdx.y * GM_WAVES + wid) * GM_KW + kq * EPL;

    // issue order = arrival order (vmcnt is in-order): first X (L2-resident, back within a microsecond), then the weights
    // chunk by chunk, so the MFMAs of chunk c start as soon as that chunk lands while the later chunks still stream
    f32x4 x[NBH][NLD][XV];
#pragma unroll
    for (int h = 0; h < NBH; ++h) {
        if constexpr (XT) {
            // entry (k4, b): lanes li = 16 consecutive batch rows = 256 contiguous bytes, kq / u / c step whole k-quads (512 B each)
            const f32x4* xp = reinterpret_cast<const f32x4*>(a.xin) + (long long)(kbase / 4) * 32 + h * 16 + li;
#pragma unroll
            for (int c = 0; c < NLD; ++c)
#pragma unroll
                for (int u = 0; u < XV; ++u) x[h][c][u] = xp[(c * EPL + u) * 32];
        } else {
            const float* xp = a.xin + (long long)min(h * 16 + li, nb_valid - 1) * K + kbase;
#pragma unroll
            for (int c = 0; c < NLD; ++c)
#pragma unroll
                for (int u = 0; u < XV; ++u) x[h][c][u] = *reinterpret_cast<const f32x4*>(xp + c * 4 * EPL + 4 * u);
        }
    }
    f32x4 w[GM_R2][NLD];
    const f32x4* wp[GM_R2];
    const int KT = K / (4 * EPL), kt0 = (blockIdx.y * GM_WAVES + wid) * NLD, NT = (a.N + 15) / 16;
#pragma unroll
    for (int t = 0; t < GM_R2; ++t)     // tile (nt, kt) starts at piece (nt*KT + kt)*64; this lane's piece is number `lane`
        wp[t] = reinterpret_cast<const f32x4*>(a.W) + ((long long)min(n0 / 16 + t, NT - 1) * KT + kt0) * 64 + lane;
#pragma unroll
    for (int c = 0; c < NLD; ++c)
#pragma unroll
        for (int t = 0; t < GM_R2; ++t) w[t][c] = __builtin_nontemporal_load(wp[t] + c * 64);
    // (weights issued in front of X measured equal at B = 16 / 32: profiles/r05_ab_gm_wfirst.log)
    // the output this thread finishes: slot = tid >> 8 = (row tile t, batch half h), lane image ol, register orr
    //   n = n0 + 16*t + 4*(ol >> 4) + orr,  b = 16*h + (ol & 15)
    const int slot = tid >> 8, ol = (tid >> 2) & 63, orr = tid & 3;
    const int ot = slot / NBH, oh = slot - ot * NBH;
    const int on = n0 + 16 * ot + 4 * (ol >> 4) + orr, ob = oh * 16 + (ol & 15);
    const bool active = slot < GM_R2 * NBH && on < a.N && ob < nb_valid;
    EpiPre pre{0.f, 0.f, nullptr, 0, 0, 0};
    if constexpr (NWV == 16) {
        if (!SPLIT && active) pre = gemv_epi_prefetch<EPI>(a, on, ob);
    }

    // keep every load above the first MFMA: without this the scheduler sinks the loads next to their uses to save
    // registers, and a wave then has a few hundred bytes in flight instead of its whole slice
    __builtin_amdgcn_sched_barrier(0);
    ER_TPG(1);                          // every load issued
    ER_TPG_WAIT(GM_R2 * NLD);           // probe only: the input image has landed (the weight loads are the youngest GM_R2 * NLD)
    ER_TPG(2);
    ER_TPG_WAIT(0);                     // probe only: the weights have landed
    ER_TPG(3);
    gm_f4 acc[GM_R2][NBH];
#pragma unroll
    for (int t = 0; t < GM_R2; ++t)
#pragma unroll
        for (int h = 0; h < NBH; ++h) acc[t][h] = (gm_f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(WT) == 2) {
        // fast mode: fp16 weights go to the fp16 matrix cores as they are (v_mfma_f32_16x16x16_f16, 16x the fp32 matrix
        // rate: the fp32-MFMA form of this loop was MFMA-issue-bound at 32 rows, 3072 cycles per wave); the fp32
        // activations keep fp32 grade by entering as two fp16 numbers x = hi + lo (|x - hi - lo| <= 2^-22 |x|), one MFMA
        // each.  A lane's 8 weights of a piece split into two k-groups of 4 (elements 0-3 / 4-7 of every lane): the matrix
        // core sums k in any order, A and B only have to agree on which k sits in which (lane, element) slot.
        // (round 5: ONE v_mfma_f32_16x16x32_f16 per piece instead of two 16x16x16 - gfx950's K = 32 form takes a lane's eight weights and
        // eight inputs at once at the same issue cost, so the matrix phase of a B = 32 projection halves: 1.6 -> 0.8 us of the
        // 7 us launch, profiles/r05_gemv_mfma_timeline.log)
#pragma unroll
        for (int c = 0; c < NLD; ++c) {
            f16x8 xh[NBH], xl[NBH];
#pragma unroll
            for (int h = 0; h < NBH; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (XT) {        // the producer split the value: halves 0..3 = hi, 4..7 = lo of this k-quad
                        const f16x8 t = __builtin_bit_cast(f16x8, x[h][c][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { xh[h][4 * j + e] = t[e]; xl[h][4 * j + e] = t[4 + e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xv = x[h][c][j][e];
                            const _Float16 hi = (_Float16)xv;
                            xh[h][4 * j + e] = hi;
                            xl[h][4 * j + e] = (_Float16)(xv - (float)hi);
                        }
                    }
                }
#pragma unroll
            for (int t = 0; t < GM_R2; ++t) {
                const f16x8 av = __builtin_bit_cast(f16x8, w[t][c]);
#pragma unroll
                for (int h = 0; h < NBH; ++h) {
                    acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, xl[h], acc[t][h], 0, 0, 0);
                    acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, xh[h], acc[t][h], 0, 0, 0);
                }
            }
        }
    } else
#pragma unroll
    for (int c = 0; c < NLD; ++c) {
        float wf[GM_R2][EPL];
#pragma unroll
        for (int t = 0; t < GM_R2; ++t) {
            if constexpr (sizeof(WT) == 4) {
                wf[t][0] = w[t][c].x; wf[t][1] = w[t][c].y; wf[t][2] = w[t][c].z; wf[t][3] = w[t][c].w;
            } else {
                const f16x8 hv = __builtin_bit_cast(f16x8, w[t][c]);
#pragma unroll
                for (int m = 0; m < 8; ++m) wf[t][m] = (float)hv[m];
            }
        }
#pragma unroll
        for (int m = 0; m < EPL; ++m)
#pragma unroll
            for (int t = 0; t < GM_R2; ++t)
#pragma unroll
                for (int h = 0; h < NBH; ++h)
                    acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][m], x[h][c][m >> 2][m & 3], acc[t][h], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < GM_R2; ++t)
#pragma unroll
        for (int h = 0; h < NBH; ++h) *reinterpret_cast<gm_f4*>(&red[wid][t * NBH + h][lane][0]) = acc[t][h];
    ER_TPG(4);                          // MFMAs done, partial tile in LDS
    __syncthreads();
    ER_TPG(5);
    if constexpr (NWV != 16) {
        // 256 threads finish the GM_R2 * NBH output tiles one after the other (raw partials only)
#pragma unroll
        for (int sl = 0; sl < GM_R2 * NBH; ++sl) {
            const int t2 = sl / NBH, h2 = sl - t2 * NBH;
            const int n2 = n0 + 16 * t2 + 4 * (ol >> 4) + orr, b2 = h2 * 16 + (ol & 15);
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) s += red[wv][sl][ol][orr];
            if (n2 < a.N && b2 < nb_valid) part[((long long)blockIdx.y * nb_valid + b2) * a.N + n2] = s;
        }
        ER_TPG(6);
        return;
    }
    if (slot < GM_R2 * NBH) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < GM_WAVES; ++wv) s += red[wv][slot][ol][orr];
        if (XT && !SPLIT && EPI == EPI_RELU) {
            // fc1 -> fc2: the four threads orr = 0..3 hold four consecutive columns of one batch row = one tiled entry of the next
            // projection's input; the quad's values meet in thread orr = 0 (DPP quad_perm), which stores the 16 bytes
            float v = fmaxf(s + pre.bias, 0.0f);
            if (!active) v = 0.f;
            const float v1 = dpp_mov<0x39>(v), v2 = dpp_mov<0x4E>(v), v3 = dpp_mov<0x93>(v);   // quad_perm [1,2,3,0] / [2,3,0,1] / [3,0,1,2]: lane 0 of a quad reads lanes 1 / 2 / 3
            if (active && orr == 0) reinterpret_cast<xt_h8*>(a.xt_out)[xt_entry(a.N, ob, on >> 2)] = xt_pack(v, v1, v2, v3);
        } else if (active) {
            if (SPLIT) part[((long long)blockIdx.y * nb_valid + ob) * a.N + on] = s;
            else gemv_epilogue<EPI>(a, on, ob, s, pre);
        }
    }
    ER_TPG(6);
}

// row-major [N][K] -> the tiled layout above; N is padded to a multiple of 16 with zero rows
template <typename WT>
__global__ __launch_bounds__(ER_WG) void tile_weights_kernel(const WT* src, f32x4* dst, int N, int K) {
    constexpr int EPL = WTraits<WT>::EPL;
    const int KT = K / (4 * EPL), NT = (N + 15) / 16;
    const long long total = (long long)NT * KT * 64;
    for (long long p = (long long)blockIdx.x * ER_WG + threadIdx.x; p < total; p += (long long)gridDim.x * ER_WG) {
        const int lane = (int)(p & 63);
        const long long tile = p >> 6;
        const int nt = (int)(tile / KT), kt = (int)(tile - (long long)nt * KT);
        const int n = nt * 16 + (lane & 15), k = kt * 4 * EPL + (lane >> 4) * EPL;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N) v = *reinterpret_cast<const f32x4*>(src + (long long)n * K + k);
        dst[p] = v;
    }
}
template <typename WT>
inline size_t tiled_weight_bytes(int N, int K) { return (size_t)((N + 15) / 16) * 16 * K * sizeof(WT); }

// out(b, n) = epilogue(sum_s part[s][b][n]) in slice order.  (Round 3 tried this finish fused with the NEXT layer's LayerNorm - one
// workgroup per batch row, saving the prep_rows launch in front of qkv: measured SLOWER, 12.7 -> 18.5 us for fc2 against 11.6 -> 9.3
// for qkv at B = 5, configs[3] 7023 -> 6944 tok/s: B workgroups walking the slices serially lose more than a launch costs;
// profiles/r03_splitk_finish_ln_fused.log.)
template <int EPI>
__global__ __launch_bounds__(ER_WG) void splitk_finish_kernel(GemvArgs a, const float* part, int S, int nb_valid) {
    const unsigned i = blockIdx.x * ER_WG + threadIdx.x;          // nb_valid <= 32 rows x N columns: 32-bit (a 64-bit division is ~100 instructions)
    if (i >= (unsigned)nb_valid * (unsigned)a.N) return;
    const int b = (int)(i / (unsigned)a.N), n = (int)(i - (unsigned)b * (unsigned)a.N);
    const EpiPre pre = gemv_epi_prefetch<EPI>(a, n, b);
    float s = 0.f;
    if (S == 4) {          // fc2: the four slices' loads go out together (the generic loop is four dependent round trips); same order of adds
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = part[((long long)k * nb_valid + b) * a.N + n];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += p[k];
    } else if (S == 16) {  // 4-wave workgroups (last layer's fc2 in the tiled path): same idea
        float p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = part[((long long)k * nb_valid + b) * a.N + n];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += p[k];
    } else {
        for (int k = 0; k < S; ++k) s += part[((long long)k * nb_valid + b) * a.N + n];
    }
    gemv_epilogue<EPI>(a, n, b, s, pre);
}

// one pass over <= 32 rows; K = ksplit * 1536; a.W must point at the TILED copy of the matrix.  `part` must hold ksplit * nb_valid * N floats when ksplit > 1.
// defer_finish: a split-K launch leaves its partials in `part` (the consumer - prep_rows_kernel with sk_part - finishes them)
template <typename WT, int EPI, bool XT = false>
inline hipError_t launch_gemv_mfma(const GemvArgs& a, int nb_valid, int K, float* part, hipStream_t st, bool defer_finish = false,
                                   bool narrow = false) {
    if (narrow) {       // 4-wave workgroups: K / 384 K-ranges per row tile, partials only (the caller's consumer finishes them)
        if constexpr (XT) {
            const int ks4 = K / (4 * GM_KW);
            if (K != ks4 * 4 * GM_KW) return hipErrorInvalidValue;
            const dim3 grid4((a.N + GM_ROWS - 1) / GM_ROWS, ks4);
            if (nb_valid > 16) hipLaunchKernelGGL((gemv_mfma_kernel<WT, 2, EPI, true, true, 4>), grid4, dim3(256), 0, st, a, nb_valid, K, part);
            else hipLaunchKernelGGL((gemv_mfma_kernel<WT, 1, EPI, true, true, 4>), grid4, dim3(256), 0, st, a, nb_valid, K, part);
            hipError_t e4 = hipGetLastError();
            if (e4 != hipSuccess || defer_finish) return e4;
            const long long total4 = (long long)nb_valid * a.N;
            hipLaunchKernelGGL((splitk_finish_kernel<EPI>), dim3((unsigned)((total4 + ER_WG - 1) / ER_WG)), dim3(ER_WG), 0, st, a, part, ks4, nb_valid);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    const int ksplit = K / (GM_WAVES * GM_KW);
    if (K != ksplit * GM_WAVES * GM_KW) return hipErrorInvalidValue;
    const dim3 grid((a.N + GM_ROWS - 1) / GM_ROWS, ksplit);
    const bool two = nb_valid > 16;
    if (ksplit == 1) {
        if (two) hipLaunchKernelGGL((gemv_mfma_kernel<WT, 2, EPI, false, XT>), grid, dim3(GM_THREADS), 0, st, a, nb_valid, K, part);
        else hipLaunchKernelGGL((gemv_mfma_kernel<WT, 1, EPI, false, XT>), grid, dim3(GM_THREADS), 0, st, a, nb_valid, K, part);
        return hipGetLastError();
    }
    if (two) hipLaunchKernelGGL((gemv_mfma_kernel<WT, 2, EPI, true, XT>), grid, dim3(GM_THREADS), 0, st, a, nb_valid, K, part);
    else hipLaunchKernelGGL((gemv_mfma_kernel<WT, 1, EPI, true, XT>), grid, dim3(GM_THREADS), 0, st, a, nb_valid, K, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || defer_finish) return e;
    const long long total = (long long)nb_valid * a.N;
    hipLaunchKernelGGL((splitk_finish_kernel<EPI>), dim3((unsigned)((total + ER_WG - 1) / ER_WG)), dim3(ER_WG), 0, st, a, part, ksplit,
                       nb_valid);
    return hipGetLastError();
}

}  // namespace er
