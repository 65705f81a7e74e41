// fp16 x fp16 GEMM for the front-end Linears (DiT, reference core/transformer/dit.py:100-140) as ONE STREAM of k-tiles per
// workgroup: loader waves and matrix waves, a five-deep LDS ring, a persistent tile loop (round 6).
//
// Why (profiles/r05_dit_fp16_kernel_stats.csv): the 4-wave LDS-DMA kernel (k_gemm.h gemm_hh_mfma_kernel) keeps ONE
// k-tile in flight per workgroup and ends every 64-deep k-step with vmcnt(0) + barrier, so a k-step costs an L2 -> LDS round trip
// (~1300 cycles for 512 cycles of MFMA), and a K = 1024 product (16 k-steps) spends as long in launch + first-tile latency + the
// LDS-staged epilogue as in its k-loop: 17-18 us for 8.6 GFLOP (3.4 us of MFMA).  The 8-wave 256 x 256 kernel halves the operand bytes
// per flop but leaves 64 tiles on the N = 1024 products.  Here, on a 128 x 128 tile:
//   * waves 4..7 only LOAD: each issues its 8 LDS-DMA pieces (global_load_lds_dwordx4, 1 KB each) of a k-tile, keeps THREE k-tiles in
//     flight behind the two that have landed (ring of S = 5 stages x 32 KB = all 160 KB of LDS) and retires them with counted vmcnt;
//   * waves 0..3 only MULTIPLY: 64 x 64 per wave (2 x 2 accumulators of 32 x 32, v_mfma_f32_32x32x16_f16), fragments prefetched two
//     16-deep sub-steps ahead into four register sets - across the k-step barrier too, which is why a k-tile is certified (every
//     loader's pieces waited for, then a barrier) one step before it is multiplied;
//   * one s_barrier per k-step is the only synchronisation: after barrier u every wave knows k-tiles u + 1 and u + 2 are in LDS and
//     k-tile u's stage is free, so the loaders refill it with k-tile u + S - 1;
//   * the k-tiles of ALL the output tiles a workgroup owns form one stream (persistent grid, one workgroup per CU): while the matrix
//     waves run a tile's epilogue the ring fills with the next tile's first k-tiles, so only the first tile of a launch pays the
//     first-byte latency;
//   * the epilogue is k_gemm.h's row-wise one (hh_epi_stage / hh_epi_rows: transposed through LDS, 16 bytes per lane, 256 contiguous
//     bytes per 16 lanes; V^T and GEGLU forms included) on ALL EIGHT waves: a matrix wave stages a 32 x 64 half of its block into the
//     ONE ring stage that is free after a tile's last k-step, then it and its loader wave finish half of the row passes each; the
//     refill of that stage waits for the epilogue's last barrier.  Plain form: bias / gate / residual of a wave's passes are requested a
//     k-loop (matrix waves) or a k-step (loaders) ahead.  (A first version accumulated the product transposed and stored straight from
//     the accumulator registers - 32-byte row segments per lane pair: 16 us against 10 us on the 4096 x 1024 residual shape,
//     profiles/r06_gemm_stream_probe_v2.log; a second ran the row-wise epilogue on the four matrix waves only: 11-14 k cycles per tile,
//     r06_gemm_stream_probe_v6.log.)
// Where it runs: launch_gemm_hh's rule (k_gemm.h gemm_hh_use_stream) - the deep feed-forward-out product of the DiT, where it is faster
// IN SITU; the probe builds (GS_TIMELINE, GS_ABL_*, ER_GEMM_PROBE_NO_EPILOGUE: scripts/probes/gemm_stream_probe.hip) are how its phases
// were measured.
// Same 128-byte row images, XOR swizzle, fragment reads and the same k order per accumulator element as gemm_hh_mfma_kernel; a product
// a.w is the same number whichever operand slot it enters by: results are BIT-IDENTICAL to the 4-wave kernels (tests/test_gpu_kernels.py).
#pragma once
#include "k_gemm.h"
#include <type_traits>

namespace er {

constexpr int GS_BM = 128, GS_BN = 128, GS_THREADS = 512;
constexpr int GS_STAGE_B = (GS_BM + GS_BN) * XBK * 2;        // bytes per ring stage: 128 A rows, then 128 B rows, 128 bytes each (32 KB)

// Tile order inside the stream: bands of GS_GH tile rows, walked column by column.  The tiles an XCD's 32 workgroups hold at the same
// time are 32 consecutive indices = a GS_GH x 8 block: per k-step the XCD's L2 fetches 4 A slices + 8 B slices (192 KB) for the 1 MB its
// CUs pull, instead of 1 + 32 (528 KB) with row-major order - the row-major first version of this kernel ran 4096^3 at 575 TFLOP/s,
// fabric-bound (profiles/r06_gemm_stream_probe_v1.log, r06_dma_rate_probe.log: a CU pulls 130 GB/s from L2 but 24 GB/s from HBM).
constexpr int GS_GH = 4;
__device__ __forceinline__ void gs_tile_coords(int lin, int ntx, int nty, int& ty, int& tx) {
    const int band = lin / (GS_GH * ntx), r0 = band * GS_GH;
    const int gh = min(GS_GH, nty - r0), in = lin - band * GS_GH * ntx;
    tx = in / gh;
    ty = r0 + in - tx * gh;
}

// (a loader wave issues EIGHT LDS-DMA pieces per k-tile - 32 pieces of 8 rows over 4 loaders: the vmcnt counts below are multiples of 8)
template <int NT>
__device__ __forceinline__ void gs_wait_tiles(int n) {      // at most n (<= NT) of this wave's k-tiles may still be in flight
    if constexpr (NT >= 3) { if (n >= 3) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; } }
    if constexpr (NT >= 2) { if (n == 2) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); return; } }
    if constexpr (NT >= 1) { if (n == 1) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Operands of one wave's share of a tile's epilogue (plain form), requested ahead of use: half h = rows 32 h .. 32 h + 31 of the 64 x 64
// block at (mwb, nwb); part 0 (the matrix wave that owns the block) takes row passes 0..3 of a half, part 1 (its loader wave) 4..7.
__device__ __forceinline__ void gs_load_pre(const GemmArgs& g, HhEpiPre (&pre)[2], int mwb, int nwb, int lane, int part) {
    const int gn = nwb + 4 * (lane & 15), rr0 = lane >> 4;
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + gn) : zero;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m0h = mwb + 32 * h;
        pre[h].bias = bias;
        pre[h].gate0 = pre[h].gate1 = one;
        pre[h].gb0 = pre[h].gb1 = 0;
        if (g.gate) {
            pre[h].gb0 = min(m0h, g.M - 1) / g.gate_rows;
            pre[h].gb1 = min(m0h + 31, g.M - 1) / g.gate_rows;
            pre[h].gate0 = *reinterpret_cast<const f32x4*>(g.gate + (long long)pre[h].gb0 * g.gate_bstride + gn);
            pre[h].gate1 = *reinterpret_cast<const f32x4*>(g.gate + (long long)pre[h].gb1 * g.gate_bstride + gn);
        }
        if (g.resid) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int gm = min(m0h + rr0 + 4 * (t + 4 * part), g.M - 1);
                pre[h].res[t] = *reinterpret_cast<const f32x4*>(g.resid + (long long)(g.resid_mod > 0 ? gm % g.resid_mod : gm) * g.ldr + gn);
            }
        }
    }
}

template <int S, int HEPI>
__global__ __launch_bounds__(GS_THREADS) void gemm_hh_stream_kernel(GemmArgs g, int ntx, int ntiles) {
    static_assert(S >= 3 && S <= 5, "ring depth");
    __shared__ __attribute__((aligned(16))) char lds[S * GS_STAGE_B];            // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.K / XBK, nty = ntiles / ntx;
    // the tiles of this workgroup: each XCD owns a contiguous run of tiles (neighbours share their A row panel in that XCD's L2),
    // the run is dealt round-robin to the XCD's workgroups (block b runs on XCD b % 8: a speed assumption only)
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int nx = tq + (xcd < tr ? 1 : 0);
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int mine = slot < nx ? (nx - slot + wx - 1) / wx : 0;                    // tiles base + slot + j wx, j < mine
    const int total = mine * nk;                                                   // k-tiles in this workgroup's stream
    if (total == 0) return;
    // EPILOGUE of a tile, all eight waves: the 64 x 64 block of matrix wave b goes through LDS in two halves of 32 rows - b stages a
    // half into its 8 KB of the ring stage the tile's last k-tile left, a barrier, then b finishes row passes 0..3 and loader wave b + 4
    // passes 4..7 (hh_epi_rows, PARTS = 2), every wave's reads of the stage returned, a barrier.  Four epilogue waves per CU took 11 us
    // for the stores of a 4096 x 1024 residual tile set (profiles/r06_gemm_stream_probe_v6.log); the loaders are idle anyway:
    // the refill of that stage has to wait for the epilogue's last barrier.
    const int eb = wid & 3, epart = wid >> 2;
    auto epilogue_rows = [&](int stage, int h, int mwb, int nwb, const HhEpiPre& pre, bool use_pre) {
        const float* sw = reinterpret_cast<const float*>(lds + stage * GS_STAGE_B) + eb * (32 * 64);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // the half is staged (the stager's ds_writes have completed)
#ifndef ER_GEMM_PROBE_NO_EPILOGUE
        hh_epi_rows<1, 2, HEPI, 2>(g, sw, mwb + 32 * h, nwb, lane, epart, &pre, use_pre);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // everybody's reads of the stage have returned
    };

    if (wid >= 4) {
        // ---------------- loader waves: pieces lw * 4 + jj of the A image and of the B image of every k-tile ----------------
        __builtin_amdgcn_s_setprio(3);
        const int lw = wid - 4, lrow = lane >> 3, lslot = lane & 7;
        const unsigned lds0 = (unsigned)(unsigned long long)(er_lptr)lds;
        // address form: scalar base (operand + k offset - 3072 bytes) + 32-bit lane offset; the instruction offset j * 1024 moves the LDS
        // destination AND the global address, so piece j's lane offset carries + 3072 - 1024 j: M0 is written once per four pieces
        // (27-38 cycles per piece against 47 with M0 saved / set / restored around every piece: profiles/r06_dma_rate_probe.log)
        const char* A = reinterpret_cast<const char*>(g.A);
        const char* B = reinterpret_cast<const char*>(g.B);
        unsigned va[4], vb[4];
        auto set_tile = [&](int j) {
            int ty, tx;
            gs_tile_coords(base + slot + j * wx, ntx, nty, ty, tx);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = 8 * (lw * 4 + jj) + lrow, sw = (lslot ^ ((r >> 1) & 7)) << 4;
                va[jj] = (unsigned)min(ty * GS_BM + r, g.M - 1) * (unsigned)(g.lda * 2) + sw + 3072u - 1024u * jj;
                vb[jj] = (unsigned)min(tx * GS_BN + r, g.N - 1) * (unsigned)(g.ldb * 2) + sw + 3072u - 1024u * jj;
            }
        };
        int ji = 0, kt = 0, cur = 0;                                                // the next k-tile to issue: tile index, k index, ring stage
        set_tile(0);
        auto issue_next = [&]() {
            const unsigned sa = lds0 + (unsigned)cur * GS_STAGE_B + (unsigned)lw * 4096u, sb = sa + GS_BM * XBK * 2;
            const char* ga = A + (long long)kt * (XBK * 2) - 3072;
            const char* gb = B + (long long)kt * (XBK * 2) - 3072;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %6\n\tglobal_load_lds_dwordx4 %2, %6 offset:1024\n\t"
                         "global_load_lds_dwordx4 %3, %6 offset:2048\n\tglobal_load_lds_dwordx4 %4, %6 offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(va[3]), "s"(sa), "s"(ga) : "memory");
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %6\n\tglobal_load_lds_dwordx4 %2, %6 offset:1024\n\t"
                         "global_load_lds_dwordx4 %3, %6 offset:2048\n\tglobal_load_lds_dwordx4 %4, %6 offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vb[0]), "v"(vb[1]), "v"(vb[2]), "v"(vb[3]), "s"(sb), "s"(gb) : "memory");
            if (++kt == nk) {
                kt = 0;
                if (++ji < mine) set_tile(ji);
            }
            cur = cur + 1 == S ? 0 : cur + 1;
        };
        int issued = 0;
        for (; issued < S - 1 && issued < total; ++issued) issue_next();
        gs_wait_tiles<S - 3>(issued - 2);                                           // k-tiles 0 and 1 have landed (this wave's pieces) ...
        asm volatile("s_barrier" ::: "memory");                                     // ... and everybody's
#ifdef GS_TIMELINE
        unsigned long long t_issue = 0, t_wait = 0, t_bar = 0;
#endif
        int u = 0, est = 0;                                                         // est: ring stage of k-tile u
        for (int jt = 0; jt < mine; ++jt) {
            int ty, tx;
            gs_tile_coords(base + slot + jt * wx, ntx, nty, ty, tx);
            const int mwb = ty * GS_BM + (lw >> 1) * 64, nwb = tx * GS_BN + (lw & 1) * 64;     // the block of matrix wave lw
            HhEpiPre pre[2];
            bool use_pre = false;
            for (int t = 0; t < nk; ++t, ++u) {
#ifdef GS_TIMELINE
                const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
                if (issued < total) { issue_next(); ++issued; }                     // k-tile u + S - 1 into the stage k-tile u - 1 left
#ifdef GS_TIMELINE
                const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
                if constexpr (HEPI == HEPI_PLAIN) {
                    if (t == nk - 1) {                                              // this wave's epilogue operands, one k-step ahead of their use
                        use_pre = hh_epi_vec(g, nwb + 60) && !(g.vt16 && nwb >= g.vt_col0);
                        if (use_pre) gs_load_pre(g, pre, mwb, nwb, lane, 1);
                    }
                }
                gs_wait_tiles<S - 3>(issued - 3 - u);                               // k-tile u + 2 has landed
#ifdef GS_TIMELINE
                const unsigned long long c2 = __builtin_amdgcn_s_memtime();
#endif
                asm volatile("s_barrier" ::: "memory");
#ifdef GS_TIMELINE
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                t_issue += c1 - c0; t_wait += c2 - c1; t_bar += c3 - c2;
#endif
                est = est + 1 == S ? 0 : est + 1;
            }
            const int free_stage = est == 0 ? S - 1 : est - 1;                      // the stage the tile's last k-tile left
            epilogue_rows(free_stage, 0, mwb, nwb, pre[0], use_pre);
            epilogue_rows(free_stage, 1, mwb, nwb, pre[1], use_pre);
        }
#ifdef GS_TIMELINE
        if (wid == 4 && lane == 0) {
            float* dbg = g.q + (long long)blockIdx.x * 8;      // (probe builds: GemmArgs::q is unused by this kernel)
            dbg[0] = (float)t_issue / total; dbg[1] = (float)t_wait / total; dbg[2] = (float)t_bar / total;
        }
#endif
        return;
    }

    // ---------------- matrix waves ----------------
    const int wm = wid >> 1, wn = wid & 1;
    const int kh = lane >> 5, li = lane & 31, swz = (li >> 1) & 7;
    const char* a_row = lds + (wm * 64 + li) * (XBK * 2);                           // + 32 i rows
    const char* b_row = lds + GS_BM * XBK * 2 + (wn * 64 + li) * (XBK * 2);
    h16x8 av[4][2], bv[4][2];                                                       // four fragment sets: sub-step ks uses set ks
    int cur = 0;
    f32x16 acc[2][2];
    asm volatile("s_barrier" ::: "memory");                                         // k-tiles 0 and 1 are in LDS
#define GS_FRAGS(STAGE, KS, SET)                                                                                        \
    do {                                                                                                                \
        const int co_ = (((2 * (KS) + kh) ^ swz) << 4) + (STAGE) * GS_STAGE_B;                                          \
        bv[SET][0] = *reinterpret_cast<const h16x8*>(b_row + co_);                                                      \
        bv[SET][1] = *reinterpret_cast<const h16x8*>(b_row + 32 * XBK * 2 + co_);                                       \
        av[SET][0] = *reinterpret_cast<const h16x8*>(a_row + co_);                                                      \
        av[SET][1] = *reinterpret_cast<const h16x8*>(a_row + 32 * XBK * 2 + co_);                                       \
    } while (0)
    GS_FRAGS(0, 0, 0);
    GS_FRAGS(0, 1, 1);
    int u = 0;
#ifdef GS_TIMELINE
    unsigned long long tl_work = 0, tl_bar = 0, tl_epi = 0, tl_drain = 0;
#endif
    for (int jt = 0; jt < mine; ++jt) {
        int ty, tx;
        gs_tile_coords(base + slot + jt * wx, ntx, nty, ty, tx);
        const int mw = ty * GS_BM + wm * 64, nw = tx * GS_BN + wn * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // plain epilogue: this wave's global operands (bias, the two possible gate rows, the residual of its 2 x 4 row passes) are requested
        // NOW and arrive under the k-loop.  Wave-uniform condition: the wave's 64 columns lie inside N and every row-wise access is a
        // 16-byte one (else hh_epi_rows loads by itself); lane map and conditions of hh_epi_rows<1, 2>.
        HhEpiPre pre[2];
        bool use_pre = false;
        if constexpr (HEPI == HEPI_PLAIN) {
            use_pre = hh_epi_vec(g, nw + 60) && !(g.vt16 && nw >= g.vt_col0);
            if (use_pre) gs_load_pre(g, pre, mw, nw, lane, 0);
        }
        for (int t = 0; t < nk; ++t, ++u) {
#ifdef GS_TIMELINE
            const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
            const int nxt = cur + 1 == S ? 0 : cur + 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // fragments two sub-steps ahead: k-tile u + 1 was certified by barrier u - 1.  UNCONDITIONAL - behind the stream's last
                // k-tile the two sets are read from a stage nobody fills and never used: a branch here makes hipcc wait for every
                // outstanding ds_read at the join (lgkmcnt(1) / (0) in front of sub-step 3's MFMAs instead of lgkmcnt(8))
#ifndef GS_ABL_NOREAD
                if (ks == 0) GS_FRAGS(cur, 2, 2);
                if (ks == 1) GS_FRAGS(cur, 3, 3);
                if (ks == 2) GS_FRAGS(nxt, 0, 0);
                if (ks == 3) GS_FRAGS(nxt, 1, 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifdef GS_ABL_NOMFMA
                if (ks == 0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv[0][0] + bv[1][1] + bv[2][0] + bv[3][1], av[0][0] + av[1][1] + av[2][0] + av[3][1], acc[0][0], 0, 0, 0);
                if (ks < 0)
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ks][i], bv[ks][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef GS_TIMELINE
            asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));     // the MFMAs have retired
            const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_barrier" ::: "memory");                                 // k-tile u's stage is free; k-tile u + 2 is in LDS
#ifdef GS_TIMELINE
            const unsigned long long c2 = __builtin_amdgcn_s_memtime();
            tl_work += c1 - c0; tl_bar += c2 - c1;
#endif
            cur = nxt;
        }
#ifdef GS_TIMELINE
        const unsigned long long e0 = __builtin_amdgcn_s_memtime();
#endif
        {
            // the stage k-tile u - 1 occupied is free (barrier u - 1) and stays free until the loaders are through the epilogue's last barrier
            const int free_stage = cur == 0 ? S - 1 : cur - 1;
            float* sw = reinterpret_cast<float*>(lds + free_stage * GS_STAGE_B) + wid * (32 * 64);
#ifdef ER_GEMM_PROBE_NO_EPILOGUE      // scripts/probes/gemm_stream_probe.hip: what the k-loop costs without the epilogue
            if (acc[0][0][0] == 12345.678f && acc[1][1][5] == 3.f && acc[0][1][7] == 1.f && acc[1][0][2] == 9.f) sw[lane] = 1.f;
#else
            hh_epi_stage<1, 2>(sw, *reinterpret_cast<const f32x16(*)[1][2]>(&acc[0]), lane);
#endif
            epilogue_rows(free_stage, 0, mw, nw, pre[0], use_pre);
#ifndef ER_GEMM_PROBE_NO_EPILOGUE
            hh_epi_stage<1, 2>(sw, *reinterpret_cast<const f32x16(*)[1][2]>(&acc[1]), lane);
#endif
            epilogue_rows(free_stage, 1, mw, nw, pre[1], use_pre);
        }
#ifdef GS_TIMELINE
        const unsigned long long e1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long e2 = __builtin_amdgcn_s_memtime();
        tl_epi += e1 - e0; tl_drain += e2 - e1;
#endif
    }
#undef GS_FRAGS
#ifdef GS_TIMELINE
    if (wid == 0 && lane == 0) {
        float* dbg = g.q + (long long)blockIdx.x * 8;      // (probe builds: GemmArgs::q is unused by this kernel)
        dbg[4] = (float)tl_work / total; dbg[5] = (float)tl_bar / total; dbg[6] = (float)tl_epi / mine; dbg[7] = (float)tl_drain / mine;
    }
#endif
}

inline int gs_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// same contract as launch_gemm_hh / launch_gemm_hh_geglu (A = fp16 [M][lda], B = fp16 weights [N][ldb], K % 64 == 0)
inline hipError_t launch_gemm_hh_stream(const GemmArgs& g, hipStream_t st, bool geglu) {
    if (g.K % XBK != 0 || g.K <= 0 || (g.lda & 7) || (g.ldb & 7)) return hipErrorInvalidValue;
    if (geglu && ((g.N & 127) || !g.c16 || !g.bias)) return hipErrorInvalidValue;
    if (g.vt16 && ((g.M & 63) || (g.vt_rows & 63) || (g.vt_col0 & 63) || ((g.N - g.vt_col0) & 63) || (g.vt_ld & 7) || g.div != 0.f || g.relu || g.gate || g.resid))
        return hipErrorInvalidValue;
    const int ntx = (g.N + GS_BN - 1) / GS_BN, ntiles = ntx * ((g.M + GS_BM - 1) / GS_BM);
    const int grid = ntiles < gs_num_cus() ? ntiles : gs_num_cus();
    if (geglu) hipLaunchKernelGGL((gemm_hh_stream_kernel<5, HEPI_GEGLU>), dim3(grid), dim3(GS_THREADS), 0, st, g, ntx, ntiles);
    else hipLaunchKernelGGL((gemm_hh_stream_kernel<5, HEPI_PLAIN>), dim3(grid), dim3(GS_THREADS), 0, st, g, ntx, ntiles);
    return hipGetLastError();
}

}  // namespace er
