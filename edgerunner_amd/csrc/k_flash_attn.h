// Flash-style attention on the fp16-input matrix cores for the compute-bound front-end (DiT self/cross attention in
// fast mode): softmax(Q K^T / sqrt(D)) V without materialising the scores (the fp32 path writes and re-reads a
// [H, N, M] matrix per sample: 60 % of a DiT forward).  Non-causal, head_dim 64, fp32 in / fp32 out,
// fp16 operands, fp32 accumulation and fp32 online softmax.  Mirrors attention(q, k, v) of
// core/transformer/attention.py:27-62 as used by SelfAttention / CrossAttention (:98-153).
//
// Both products are computed TRANSPOSED so that everything a query row needs stays in one lane column:
//   S^T = K Q^T   (A = K tile rows from LDS, B = Q fragment in registers)  -> lane (q = lane&31) holds 16 keys x 2 blocks
//   O^T = V^T P^T (A = V^T rows from LDS,    B = P^T = the S^T registers repacked to fp16, no data movement)
// so the row max / sum need one xor-32 exchange and the O rescale factor is lane-local.  The MFMA sums over k in
// whatever internal order it likes: A and B are always fed the same (lane-half, element) -> k assignment.
#pragma once
#include "er_common.h"

namespace er {

struct FlashArgs {
    const float* Q; const float* K; const float* V; float* O;
    int N, M;                       // queries, keys
    int ldq, ldk, ldv, ldo;         // row strides (floats)
    long long qs_b, ks_b, vs_b, os_b;   // batch strides
    int head_stride;                // offset between heads inside a row (= D)
    float scale;                    // 1/sqrt(D)
    _Float16* O16;                  // optional: the output rounded to fp16 INSTEAD of O (same strides): it only feeds an fp16 Linear
};

constexpr int FA_D = 64, FA_QW = 32, FA_KT = 64, FA_LD = 72;   // head dim, q rows per wave, keys per tile, LDS row stride (halves)
typedef _Float16 fa_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 fa_h4 __attribute__((ext_vector_type(4)));
typedef float fa_f16v __attribute__((ext_vector_type(16)));

// grid (ceil(N / 128), H, B), 256 threads: wave w owns q rows [128*bx + 32*w, +32)
__global__ __launch_bounds__(ER_WG) void flash_attn_f16_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 Ks[FA_KT * FA_LD];   // [key][d]
    __shared__ __attribute__((aligned(16))) _Float16 Vt[FA_D * FA_LD];    // [d][key]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (ER_NWAVES * FA_QW) + wid * FA_QW;
    const float* Q = a.Q + b * a.qs_b + h * a.head_stride;
    const float* K = a.K + b * a.ks_b + h * a.head_stride;
    const float* V = a.V + b * a.vs_b + h * a.head_stride;
    float* O = a.O16 ? nullptr : a.O + b * a.os_b + h * a.head_stride;
    _Float16* O16 = a.O16 ? a.O16 + b * a.os_b + h * a.head_stride : nullptr;

    // Q fragment (B operand of S^T): q = q0 + li, d = ks*16 + half*8 + e
    fa_h8 qb[4];
    {
        const int q = min(q0 + li, a.N - 1);
        const float* qr = Q + (long long)q * a.ldq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(qr + ks * 16 + half * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(qr + ks * 16 + half * 8 + 4);
            qb[ks] = (fa_h8){(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w,
                             (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
        }
    }
    fa_f16v ot[2];                    // O^T accumulators: rows d = db*32 + (r&3)+8*(r>>2)+4*half, column q = li
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;     // per query (lane column); l_run covers this lane-half's keys only

    const int ntiles = (a.M + FA_KT - 1) / FA_KT;
    // softmax in base 2 on the hardware exponential (v_exp_f32): p = 2^((s - m) scale log2 e).  The probabilities are rounded to fp16
    // before P.V, far coarser than the 1-ulp difference to expf; the accurate expf cost ~10 instructions x 32 scores per lane and
    // tile and made the kernel VALU-bound at 9 % of the matrix rate (profiles/r03_dit_fp16_kernel_stats_*.csv)
    const float sl2 = a.scale * 1.4426950408889634f;
    for (int t = 0; t < ntiles; ++t) {
        const int kbase = t * FA_KT;
        __syncthreads();              // previous tile fully consumed
        // stage K -> Ks[key][d], V -> Vt[d][key] (fp16); 64x64 floats each = 4 float4 per thread
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + ER_WG * u, key = idx >> 4, c4 = idx & 15;
            const int gk = min(kbase + key, a.M - 1);
            const f32x4 kv = *reinterpret_cast<const f32x4*>(K + (long long)gk * a.ldk + 4 * c4);
            *reinterpret_cast<fa_h4*>(&Ks[key * FA_LD + 4 * c4]) = (fa_h4){(_Float16)kv.x, (_Float16)kv.y, (_Float16)kv.z, (_Float16)kv.w};
            const f32x4 vv = *reinterpret_cast<const f32x4*>(V + (long long)gk * a.ldv + 4 * c4);
            Vt[(4 * c4 + 0) * FA_LD + key] = (_Float16)vv.x;
            Vt[(4 * c4 + 1) * FA_LD + key] = (_Float16)vv.y;
            Vt[(4 * c4 + 2) * FA_LD + key] = (_Float16)vv.z;
            Vt[(4 * c4 + 3) * FA_LD + key] = (_Float16)vv.w;
        }
        __syncthreads();

        // S^T = K Q^T: two 32-key blocks, keys kbase + kb*32 + (r&3)+8*(r>>2)+4*half
        fa_f16v st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const fa_h8 ka = *reinterpret_cast<const fa_h8*>(&Ks[(kb * 32 + li) * FA_LD + ks * 16 + half * 8]);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qb[ks], st[kb], 0, 0, 0);
            }
        }
        // online softmax for query li
        float mloc = -INFINITY;
        if (kbase + FA_KT <= a.M) {                  // wave-uniform: only the last tile can hold keys beyond M
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mloc = fmaxf(fmaxf(mloc, st[kb][r]), st[kb][r + 1]);      // RAW scores (v_max3_f32); scaled below
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float s = key < a.M ? st[kb][r] : -INFINITY;
                    st[kb][r] = s;
                    mloc = fmaxf(mloc, s);
                }
        }
        // log2 units: exp(x * scale) = exp2(x * sl2).  The maximum is taken over the raw scores (sl2 > 0) and every probability is ONE
        // fma + exp2: exp2(fma(s, sl2, -m)) - round 3 spent a multiply and a subtract per score here, and this VALU loop, not the 16
        // MFMAs of a key tile, bounds the kernel at head_dim 64.  (Packed v_pk_fma_f32 / v_pk_add_f32 forms of the same loop and
        // a zero C operand instead of clearing the score registers measured no different on the same box: profiles/r04_dit_softmax_ab.log)
        mloc = xor_max<32>(mloc) * sl2;
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // m_run = -inf on the first tile -> 0
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[kb][r], sl2, -m_new));  // masked keys: exp2(-inf) = 0
                st[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull) {      // wave-uniform: no query of this wave met a new maximum -> alpha = 1
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;
        }
        m_run = m_new;

        // O^T += V^T P^T: per 32-key block two 16-key steps; B = P registers 8*step .. +8 of this lane (keys
        // 16*step + 4*half + {0..3} and + 8 + {0..3}); A = the same keys of row d from Vt
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int stp = 0; stp < 2; ++stp) {
                const fa_h8 pb = {(_Float16)st[kb][8 * stp + 0], (_Float16)st[kb][8 * stp + 1], (_Float16)st[kb][8 * stp + 2],
                                  (_Float16)st[kb][8 * stp + 3], (_Float16)st[kb][8 * stp + 4], (_Float16)st[kb][8 * stp + 5],
                                  (_Float16)st[kb][8 * stp + 6], (_Float16)st[kb][8 * stp + 7]};
                const int kcol = kb * 32 + 16 * stp + 4 * half;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const _Float16* vr = &Vt[(db * 32 + li) * FA_LD + kcol];
                    const fa_h4 v0 = *reinterpret_cast<const fa_h4*>(vr);
                    const fa_h4 v1 = *reinterpret_cast<const fa_h4*>(vr + 8);
                    const fa_h8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, ot[db], 0, 0, 0);
                }
            }
    }
    const float l_tot = xor_sum<32>(l_run);
    const int q = q0 + li;
    if (q < a.N) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long o = (long long)q * a.ldo + db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = ot[db][r] / l_tot;
                if (O16) O16[o] = (_Float16)v;
                else O[o] = v;
            }
    }
}

inline hipError_t launch_flash_attn_f16(const FlashArgs& a, int H, int B, hipStream_t st) {
    dim3 grid((a.N + ER_NWAVES * FA_QW - 1) / (ER_NWAVES * FA_QW), H, B);
    hipLaunchKernelGGL(flash_attn_f16_kernel, grid, dim3(ER_WG), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The same attention with q / k / v ALREADY in fp16 (written by the qkv GEMM's epilogue) and V pre-transposed: the K and V^T tiles
// come in by LDS-DMA (global_load_lds_dwordx4) into the XOR-swizzled image of k_gemm.h (chunk c of row r at slot c ^ ((r >> 1) & 7)),
// two LDS stages, one barrier per key tile - no staging registers, no conversion, no ds_write (flash_attn_f16_kernel spends its
// time there: fp32 loads, 12 conversions and 16 two-byte transposing LDS stores per thread and tile: 183 TFLOP/s on the DiT
// self-attention).  Arithmetic identical to flash_attn_f16_kernel: the operands are the same fp16 roundings.
//   Q, K : fp16, row = token, the head's 64 values contiguous (row strides ldq / ldk halves)
//   Vt   : fp16 [batch][head][64][ldvt], a row = the keys in fa_vt_pos order (a fixed permutation inside every group of 16), ZERO
//          beyond M up to a multiple of 64 (p = 0 there, 0 * NaN is not)
struct FlashHArgs {
    const _Float16* Q; const _Float16* K; const _Float16* Vt; _Float16* O16;
    int N, M, ldq, ldk, ldvt, ldo;
    long long qs_b, ks_b, vts_b, vts_h, os_b;
    int head_stride;
    float scale;
};
typedef __attribute__((address_space(3))) void* fa_lptr;

// One 16-byte-per-lane LDS-DMA piece: LDS[lds_byte_addr + 16 lane] <- gsrc (per lane).  Inline asm ON PURPOSE: hipcc treats the
// builtin as an LDS write that every later ds_read may alias and drains vmcnt(0) before the first fragment read of EVERY tile, which
// serialises a multi-stage pipeline (seen in the ISA of the 4-stage loop below); an asm statement is invisible to that bookkeeping,
// so the waits are the counted ones written in the loop.  M0 (the DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void fa_glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

__global__ __launch_bounds__(ER_WG) void flash_attn_hh_kernel(FlashHArgs a) {
    constexpr int TILE = 64 * 64;                                   // halves per K (or V^T) tile image
    // FOUR stages: a key tile is only 16 MFMAs per wave (~0.25 us) while a tile's LDS-DMA takes 1-3 us to land with 512 workgroups
    // loading at once - with two stages the kernel waited for memory on every tile (profiles/r03_dit_*: 146 us for the 2048 x 2048
    // self-attention).  Three tiles stay in flight across the (raw) barrier; waits are counted, never vmcnt(0) in the steady state.
    constexpr int NST = 4;
    __shared__ __attribute__((aligned(16))) _Float16 lds[NST * 2 * TILE];   // [stage][K | Vt][64 rows][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (ER_NWAVES * FA_QW) + wid * FA_QW;
    const _Float16* Q = a.Q + b * a.qs_b + h * a.head_stride;
    const _Float16* K = a.K + b * a.ks_b + h * a.head_stride;
    const _Float16* Vt = a.Vt + b * a.vts_b + h * a.vts_h;
    _Float16* O16 = a.O16 + b * a.os_b + h * a.head_stride;

    fa_h8 qb[4];
    {
        const _Float16* qr = Q + (long long)min(q0 + li, a.N - 1) * a.ldq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qb[ks] = *reinterpret_cast<const fa_h8*>(qr + ks * 16 + half * 8);
    }
    // consume q HERE: otherwise hipcc parks its waits for these four loads at their first use inside the tile loop, where - not
    // knowing about the asm LDS-DMA pieces queued behind them - its vmcnt(3..0) would drain the whole pipeline on every tile
    asm volatile("" :: "v"(qb[0]), "v"(qb[1]), "v"(qb[2]), "v"(qb[3]));
    // this wave's LDS-DMA pieces: piece i = rows 8i .. 8i+7 of a tile; wave w takes pieces 2w, 2w+1 of K and of V^T
    const int lrow = lane >> 3, lslot = lane & 7;
    const _Float16* pk[2];
    const _Float16* pv[2];
    int krow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (2 * wid + j) + lrow;
        const int sl = (lslot ^ ((r >> 1) & 7)) << 3;
        krow[j] = r;
        pk[j] = K + sl;                                             // + (clamped key) * ldk per tile
        pv[j] = Vt + (long long)r * a.ldvt + sl;                    // + kbase per tile
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(fa_lptr)lds;
    auto issue = [&](int t, int s) {
        const int kbase = t * FA_KT;
        const unsigned ks_ = lds_base + (unsigned)(s * 2 * TILE + 8 * 2 * wid * 64) * 2u;      // bytes; this wave's first piece
        const unsigned vs_ = ks_ + TILE * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const _Float16* src = pk[j] + (long long)min(kbase + krow[j], a.M - 1) * a.ldk;
            fa_glds16(src, ks_ + j * 8 * 64 * 2u);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) fa_glds16(pv[j] + kbase, vs_ + j * 8 * 64 * 2u);
    };

    fa_f16v ot[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int swz = (li >> 1) & 7;
    const int ntiles = (a.M + FA_KT - 1) / FA_KT;
    const float sl2 = a.scale * 1.4426950408889634f;       // base-2 softmax, see flash_attn_f16_kernel
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (p < ntiles) issue(p, p);
    for (int t = 0; t < ntiles; ++t) {
        const int kbase = t * FA_KT, cur = t & (NST - 1);
        // tile t has landed when at most the pieces of the tiles issued after it (4 per tile and wave) are outstanding
        const int ahead = min(NST - 2, ntiles - 1 - t);
        // wait and barrier in ONE statement: the s_barrier builtin carries no fence, so nothing else would keep a compiler-scheduled
        // fragment read from moving between the counted wait and the barrier.  (The counts assume the loop issues no other VMEM
        // operation - no scratch: tests/test_isa_hygiene.py checks the kernel for both.)
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // everybody's pieces of tile t are in LDS; everybody is done reading tile t - 1
        if (t + NST - 1 < ntiles) issue(t + NST - 1, (t + NST - 1) & (NST - 1));      // into the stage tile t - 1 just vacated
        const _Float16* Ks = lds + cur * 2 * TILE;
        const _Float16* Vs = Ks + TILE;
        fa_f16v st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const fa_h8 ka = *reinterpret_cast<const fa_h8*>(Ks + (kb * 32 + li) * 64 + (((2 * ks + half) ^ swz) << 3));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qb[ks], st[kb], 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
        if (kbase + FA_KT <= a.M) {                  // wave-uniform: only the last tile can hold keys beyond M
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mloc = fmaxf(fmaxf(mloc, st[kb][r]), st[kb][r + 1]);      // RAW scores (v_max3_f32); scaled below
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float s = key < a.M ? st[kb][r] : -INFINITY;
                    st[kb][r] = s;
                    mloc = fmaxf(mloc, s);
                }
        }
        mloc = xor_max<32>(mloc) * sl2;
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[kb][r], sl2, -m_new));
                st[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull) {      // wave-uniform: no query of this wave met a new maximum -> alpha = 1
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;
        }
        m_run = m_new;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int stp = 0; stp < 2; ++stp) {
                const fa_h8 pb = {(_Float16)st[kb][8 * stp + 0], (_Float16)st[kb][8 * stp + 1], (_Float16)st[kb][8 * stp + 2],
                                  (_Float16)st[kb][8 * stp + 3], (_Float16)st[kb][8 * stp + 4], (_Float16)st[kb][8 * stp + 5],
                                  (_Float16)st[kb][8 * stp + 6], (_Float16)st[kb][8 * stp + 7]};
                // V^T keys are stored in the order the P registers hold them (fa_vt_pos): this lane-half's 8 keys of the step
                // - 16 stp + 4 half + {0..3} and + 8 + {0..3} - are ONE 16-byte chunk, number kb*4 + 2*stp + half of the row
                const int c0 = kb * 4 + 2 * stp + half;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const fa_h8 va = *reinterpret_cast<const fa_h8*>(Vs + (db * 32 + li) * 64 + ((c0 ^ swz) << 3));
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, ot[db], 0, 0, 0);
                }
            }
    }
    const float l_tot = xor_sum<32>(l_run);
    const int q = q0 + li;
    if (q < a.N) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {                        // registers r .. r+3 = 4 consecutive head dims
                const long long o = (long long)q * a.ldo + db * 32 + 8 * (r >> 2) + 4 * half;
                *reinterpret_cast<fa_h4*>(O16 + o) = (fa_h4){(_Float16)(ot[db][r] / l_tot), (_Float16)(ot[db][r + 1] / l_tot),
                                                             (_Float16)(ot[db][r + 2] / l_tot), (_Float16)(ot[db][r + 3] / l_tot)};
            }
    }
}

// Round 6 measured two restructurings of this kernel and kept neither (both live in commit 44ab76c with their probe,
// scripts/probes/flash_hh_timeline_probe.hip; logs profiles/r06_flash_hh_*.log, r06_dit_trace_ab.log):
//  * SOFTWARE PIPELINING - S^T of key tile t + 1 issued ahead of the softmax of tile t (five stages): bit-identical, 63.0 -> 54.4 us for
//    the 2048 x 2048 self-attention when the kernel is replayed back to back on random data, but 51.2 -> 51.4 us IN SITU (rocprofv3
//    medians of the two libraries on one box, the kernel between the DiT's GEMMs at a higher clock): no gain where it counts;
//  * an EIGHT-wave workgroup whose two waves per SIMD alternate a matrix phase and a softmax phase in anti-phase behind the workgroup
//    barrier: 72 us.  Its cycle stamps explain both results: a wave that has a pipe to itself still needs ~1400 cycles for its eight
//    S + eight P V MFMAs with their sixteen fragment reads, and ~1400 for its ~200 softmax instructions - one wave issues an instruction
//    every 4-7 cycles, so the loop is bound by the instruction issue of the two waves a SIMD holds, not by either pipe; s_setprio 1
//    around the score MFMAs measured equal.  (Serial form, stamps per key tile and wave: wait + barrier 698 | S 333 | softmax 1205 |
//    P V 595 cycles.)
inline hipError_t launch_flash_attn_hh(const FlashHArgs& a, int H, int B, hipStream_t st) {
    dim3 grid((a.N + ER_NWAVES * FA_QW - 1) / (ER_NWAVES * FA_QW), H, B);
    hipLaunchKernelGGL(flash_attn_hh_kernel, grid, dim3(ER_WG), 0, st, a);
    return hipGetLastError();
}

// position of key k inside its group of 16 in a V^T row: {0-3, 8-11, 4-7, 12-15} - the order in which the S^T / P^T accumulator
// registers of a lane half hold the keys of a 16-key MFMA step, so that the matching V^T operand is 8 consecutive halves
__host__ __device__ inline int fa_vt_pos(int k) { return (k & ~15) | (k & 3) | ((k & 8) >> 1) | ((k & 4) << 1); }

// vt[b][h][d][fa_vt_pos(key)] = v[b*M + key][h*64 + d] for key < M, 0 for M <= key < Mp   (v: fp16 rows of stride ldv halves)
__global__ __launch_bounds__(ER_WG) void transpose_v_f16_kernel(const _Float16* v, _Float16* vt, int M, int Mp, int ldv, long long vs_b) {
    __shared__ _Float16 tile[64][66];
    const int h = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 4 rows per pass
    for (int r = ty; r < 64; r += 4) {
        const int key = k0 + r;
        tile[r][tx] = key < M ? v[b * vs_b + (long long)key * ldv + h * 64 + tx] : (_Float16)0.f;
    }
    __syncthreads();
    _Float16* out = vt + (((long long)b * gridDim.y + h) * 64) * Mp;
    for (int d = ty; d < 64; d += 4) out[(long long)d * Mp + k0 + fa_vt_pos(tx)] = tile[tx][d];
}

}  // namespace er
