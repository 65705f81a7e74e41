// Weight-streaming GEMV for the decode step (batch 1..4 rows per pass; 16 per pass in the batched
// variant below); weights fp32 (exact mode) or fp16 (fast mode), fp32 accumulate.
//
// Replaces the per-token nn.Linear calls of the reference decoder
// (core/transformer/modeling_opt.py:185,189-190 q/k/v, :232 out_proj, :281 fc1,
// :284 fc2, :497 lm_head) and fuses what surrounds them:
//   prologue  LayerNorm of the previous sub-block's residual sum (post-LN decoder,
//             modeling_opt.py:273-274 / :287-288), or token+position embedding
//             (modeling_opt.py:340-342, 355-357) for layer 0;
//   epilogue  +bias, ReLU (:282), +residual (:273 / :287), or the KV-cache append
//             that replaces torch.cat (:191-192).
//
// HBM-bound: every weight element is read exactly once per token.  Layout: W is the
// nn.Linear weight [N][K] row-major, so one output row is a contiguous K-vector.
// A wave owns RW rows; lane l reads float4 #(j*64+l) of the row slice (1 KiB per
// wave-instruction, fully coalesced), keeps J*RW loads in flight, FMAs against the
// LayerNorm'd input held in registers (staged once per workgroup through LDS) and
// finishes with a 64-lane shuffle reduction.  For K = 4 slices (fc2) the 4 waves of
// a workgroup split K and combine through LDS.
#pragma once
#include "er_common.h"

namespace er {

enum { PRO_NONE = 0, PRO_LN = 1, PRO_EMBED = 2 };
enum { EPI_STORE = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_QKV = 3 };

struct GemvArgs {
    const void* W;         // [N][K] in the kernel's weight type
    const float* bias;     // [N] or nullptr
    int N;
    // prologue
    const float* xin;      // PRO_NONE: input [NB][K]; PRO_LN: pre-LN vector [NB][K]
    const float* ln_w;     // PRO_LN
    const float* ln_b;
    float eps;
    float* hout;           // PRO_LN / PRO_EMBED: block 0 stores the prologue result here ([NB][K]); may be null
    const float* embd;     // PRO_EMBED: token table [V][K]
    const float* posemb;   //            position table [P][K]
    const int* tok;        //            current token per row  (device)
    const int* pos;        // PRO_EMBED / EPI_QKV: position of the token being fed, per row (device)
    // epilogue
    float* out;            // [NB][N]
    const float* resid;    // EPI_RESID: [NB][N]
    float* q;              // EPI_QKV: [NB][hidden]
    void* kcache;          //          [B][H][Lcap][D] (this layer), fp32 or fp16
    void* vcache;
    int kv_half;           //          1: the cache holds _Float16
    int hidden, head_dim, l_cap;
    long long kv_bstride;  // H*Lcap*D
    // fast-mode batches on the matrix cores (k_gemv_mfma.h, XT): activations exchanged in the matrix cores' own operand layout
    void* xt_out;          // prep_rows / EPI_RELU: also (resp. instead) write the row in the tiled hi | lo layout xt_entry() describes
    // prep_rows<PRO_LN> reading a DEFERRED split-K finish (the previous layer's fc2, k_gemv_mfma.h): its input row is
    // ((p0 + p1 + p2 + p3) + sk_bias) + sk_resid - splitk_finish_kernel's sum and gemv_epilogue<EPI_RESID>'s adds, in their order
    const float* sk_part;  // [groups][4][rows of the group][K] partials (null: read xin)
    const float* sk_bias;  // [K]
    const float* sk_resid; // [B][K]
    int sk_batch;          // B: a group holds min(32, B - 32 g) rows
    int sk_slices;         // K-range partials per row: 4 (16-wave workgroups) or 16 (4-wave workgroups of fc2) - 0 counts as 4
};

// ---- tiled activations for the batched matrix-core projections (fast mode).  A group of 32 batch rows x K columns is stored as
// [K/4][32][8 halves]: entry (k4, b) = 16 bytes = {hi(x[b][4 k4 .. 4 k4+3]), lo(...)} with x = hi + lo (hi = (fp16)x, lo = (fp16)(x - hi)):
// exactly the B operands of the two v_mfma_f32_16x16x16_f16 a lane issues per weight quad, so the consumer's wave-load is 4 runs of
// 256 contiguous bytes (16 batch rows x 16 B) instead of 16 rows x 64 B out of 128-byte lines 6 KiB apart, and nobody converts on
// the consumer side (round 3: every one of the 144..192 workgroups re-split the whole [32][1536] input; profiles/r04_batch_kinds_*).
// Group g of a batch (rows 32 g ..) starts K * 32 floats behind group g - 1: the same offset as 32 row-major rows.
typedef _Float16 xt_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 xt_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ long long xt_entry(int K, int b, int k4) {            // index of the 16-byte entry (in entries)
    return (long long)(b >> 5) * (K / 4) * 32 + (long long)k4 * 32 + (b & 31);
}
__device__ __forceinline__ xt_h8 xt_pack(float x0, float x1, float x2, float x3) {
    xt_h8 r;
    const float xs[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hi = (_Float16)xs[e];
        r[e] = hi;
        r[4 + e] = (_Float16)(xs[e] - (float)hi);
    }
    return r;
}

// Weight storage types: fp32 (exact mode) or fp16 (fast mode, fp32 accumulate).  One 16-byte load
// carries EPL weights; the matching EPL inputs are EPL/4 consecutive float4.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename WT> struct WTraits;
template <> struct WTraits<float> { static constexpr int EPL = 4; };
template <> struct WTraits<_Float16> { static constexpr int EPL = 8; };

template <typename WT>
__device__ __forceinline__ float dot_w(const f32x4& wraw, const f32x4* x, float s) {
    if constexpr (sizeof(WT) == 4) {
        return dot4(wraw, x[0], s);
    } else {
        const f16x8 h = __builtin_bit_cast(f16x8, wraw);
        s = fmaf((float)h[0], x[0].x, s); s = fmaf((float)h[1], x[0].y, s);
        s = fmaf((float)h[2], x[0].z, s); s = fmaf((float)h[3], x[0].w, s);
        s = fmaf((float)h[4], x[1].x, s); s = fmaf((float)h[5], x[1].y, s);
        s = fmaf((float)h[6], x[1].z, s); s = fmaf((float)h[7], x[1].w, s);
        return s;
    }
}

// bias / residual / cache position are fetched at kernel entry (EpiPre) and the destination ADDRESS is finished there too, so that
// the epilogue after the reduction is one add and one store.  Left to the compiler, the tail of the qkv kernel re-read five kernel
// arguments (s_load behind the reduction), ran two integer divisions (n / hidden, c / head_dim) and - worse - fetched pos[b] with a
// SCALAR load issued in the middle of the LayerNorm prologue: scalar loads share lgkmcnt with LDS, so the s_waitcnt before the
// prologue's first barrier sat out the whole memory round trip of that load (the ~0.9 us the qkv launch was above fc1's shape for
// shape, round 3).  pos[b] is therefore read with a VECTOR load (an opaque zero in the index keeps hipcc from proving the address
// uniform): it queues first in vmcnt order and nobody waits for it before the tail.
struct EpiPre { float bias; float resid; char* dst; int half; int pos; int pos_scale; };   // final address = dst + pos * pos_scale

__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// The two halves of the prefetch: the LOADS (bias, residual, position) are issued in front of the weight stream, the address
// ARITHMETIC (two integer divisions per row for EPI_QKV, ~100 instructions) behind it - it has until the tail, and in front of the
// weight loads it would only delay their issue.
template <int EPI>
__device__ __forceinline__ EpiPre gemv_epi_loads(const GemvArgs& a, int n, int b) {
    EpiPre e{0.f, 0.f, nullptr, 0, 0, 0};
    n = min(n, a.N - 1);
    if (a.bias) e.bias = a.bias[n];
    if (EPI == EPI_RESID) e.resid = a.resid[(long long)b * a.N + n];
    if (EPI == EPI_QKV) e.pos = a.pos[b + opaque_zero()];      // in flight until the tail: nothing before it may depend on it
    return e;
}

template <int EPI>
__device__ __forceinline__ void gemv_epi_address(const GemvArgs& a, int n, int b, EpiPre& e) {
    n = min(n, a.N - 1);
    if (EPI == EPI_QKV) {   // rows [0,hidden) = q, [hidden,2h) = k, [2h,3h) = v
        const int which = n / a.hidden;
        const int c = n - which * a.hidden;
        if (which == 0) {
            e.dst = reinterpret_cast<char*>(a.q + (long long)b * a.hidden + c);
        } else {
            const int h = c / a.head_dim, d = c - h * a.head_dim;
            const int esz = a.kv_half ? 2 : 4;
            char* cache = reinterpret_cast<char*>((which == 1) ? a.kcache : a.vcache);
            e.dst = cache + ((long long)b * a.kv_bstride + (long long)h * a.l_cap * a.head_dim + d) * esz;
            e.pos_scale = a.head_dim * esz;
            e.half = a.kv_half;
        }
    } else {
        e.dst = reinterpret_cast<char*>(a.out + (long long)b * a.N + n);
    }
    unsigned long long pin = reinterpret_cast<unsigned long long>(e.dst);      // keep the address arithmetic up here (machine sinking
    asm volatile("" : "+v"(pin));                                              // would move it back behind the reduction)
    e.dst = reinterpret_cast<char*>(pin);
}

template <int EPI>
__device__ __forceinline__ EpiPre gemv_epi_prefetch(const GemvArgs& a, int n, int b) {
    EpiPre e = gemv_epi_loads<EPI>(a, n, b);
    gemv_epi_address<EPI>(a, n, b, e);
    return e;
}

template <int EPI>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, int n, int b, float v, const EpiPre& e) {
    v += e.bias;
    if (EPI == EPI_RELU) v = fmaxf(v, 0.0f);
    if (EPI == EPI_RESID) v += e.resid;
    char* dst = e.dst;
    if (EPI == EPI_QKV) dst += (long long)e.pos * e.pos_scale;
    if (EPI == EPI_QKV && e.half) *reinterpret_cast<_Float16*>(dst) = (_Float16)v;     // round-to-nearest-even
    else *reinterpret_cast<float*>(dst) = v;
}

// K = KS * 1536 (a wave reduces one 1536-slice = J 16-byte loads per lane, J = 6 for fp32 weights,
// 3 for fp16).  Dynamic LDS: NB*K floats (input) + 64 floats scratch.
// NW = waves per workgroup.  The grid should be a whole number of workgroups per CU (256 CUs): 4608 qkv rows are 768
// workgroups of 3 waves x 2 rows (3 per CU) - with 4-wave workgroups they are 1152 (4.5 per CU: the CUs that get 5 set
// the kernel's time) - and the 1536 out_proj rows are 512 workgroups of 3 waves x 1 row.
//
// Kernel arguments: the seven pointers the FIRST loads of a wave need lead the argument list as scalars, the struct follows.  With
// `-mllvm -amdgpu-kernarg-preload-count` (edgerunner_amd/build.py) the firmware hands leading scalar arguments to every wave in SGPRs
// at launch, so the input / LayerNorm / bias / position loads and the weight stream go out without first waiting for an s_load of the
// argument block (a struct passed by value is never preloaded; measured on the plain-kernel chain: -0.09 us per launch,
// profiles/r03_kernarg_preload_probe.log).  The same fields inside the struct are ignored.
template <typename WT, int KS, int NB, int RW, int PRO, int EPI, int NW = ER_NWAVES>
__global__ __launch_bounds__(64 * NW) void gemv_kernel(const void* pW, const float* pxin, const float* pln_w, const float* pln_b,
                                                       const float* pbias, const float* presid, const int* ppos, GemvArgs a_) {
    GemvArgs a = a_;
    a.W = pW; a.xin = pxin; a.ln_w = pln_w; a.ln_b = pln_b; a.bias = pbias; a.resid = presid; a.pos = ppos;
    constexpr int TPB = 64 * NW;
    constexpr int EPL = WTraits<WT>::EPL, XV = EPL / 4, SL = 1536, J = SL / (64 * EPL);
    constexpr int K = KS * SL;
    // the prologue (LayerNorm / embedding) always runs on at most 4 waves with the 256-thread element mapping and reduction
    // order, so a 6-wave workgroup (768 qkv workgroups = 3 per CU) produces bit-identical inputs; waves 4 and 5 only take
    // part in its barriers
    constexpr int PW = NW > 4 ? 4 : NW, PTPB = 64 * PW;
    constexpr int PT = K / PTPB;   // elements per thread in the prologue
    static_assert(K % PTPB == 0 && (KS == 1 || NW == ER_NWAVES), "prologue / split-K shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;              // [NB][K] (PRO_NONE reads its input straight from global memory: no image)
    float* red = smem + (PRO == PRO_NONE ? 0 : NB * K);    // 64 floats
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int slice = (KS == 1) ? 0 : wid;                 // K-slice this wave reduces
    const bool pro = (NW <= 4) || wid < PW;                // wave-uniform: this wave takes part in the prologue arithmetic
    const int row0 = (KS == 1) ? (blockIdx.x * NW + wid) * RW : blockIdx.x * RW;

    // ---------------- loads, in the order their consumers run.  Vector loads complete in issue order (vmcnt), so the
    // small prologue / epilogue operands go FIRST and the weight stream behind them: the LayerNorm reductions then
    // overlap the weights' HBM round trip instead of waiting for the last weight byte (round 1 issued the weights
    // first and every wave sat in s_waitcnt until the whole stream had landed before it could touch x).
    float v[NB][PT];
    float v2[PRO == PRO_EMBED ? NB : 1][PT];   // PRO_EMBED: position rows (added after the weight loads are out)
    float lw[PT], lb[PT];
    if (PRO == PRO_EMBED && pro) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float* e = a.embd + (long long)a.tok[b] * K;
            const float* p = a.posemb + (long long)a.pos[b] * K;
#pragma unroll
            for (int i = 0; i < PT; ++i) { v[b][i] = e[tid + i * PTPB]; v2[b][i] = p[tid + i * PTPB]; }
        }
    } else if (PRO == PRO_LN && pro) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float* x = a.xin + (long long)b * K;
#pragma unroll
            for (int i = 0; i < PT; ++i) v[b][i] = x[tid + i * PTPB];
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) { lw[i] = a.ln_w[tid + i * PTPB]; lb[i] = a.ln_b[tid + i * PTPB]; }
    }
    // epilogue operands of the (row, batch) pairs this thread will finish
    EpiPre pre[RW][NB];
    if (KS == 1) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) pre[r][b] = gemv_epi_loads<EPI>(a, row0 + r, b);
    } else {
        const int t = min(tid, RW * NB - 1);
        pre[0][0] = gemv_epi_loads<EPI>(a, row0 + t / NB, t % NB);
    }
    // (Round 3 shipped, unmeasured, a variant that loaded PRO_NONE's input slice HERE, in front of the weight stream; measured in
    // round 4 it made fc2 SLOWER - 8.10 vs 7.97 us, profiles/r04_ab_b5b758c_and_om_rpw.log - the six extra 16-byte loads per lane delay
    // the first weight request of every wave by more than the L2 round trip they take off the tail.  The input is read in the main
    // loop again.)
    f32x4 w[RW][J];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = min(row0 + r, a.N - 1);            // clamp: out-of-range rows are loaded but never stored
        const f32x4* wr = reinterpret_cast<const f32x4*>(reinterpret_cast<const WT*>(a.W) + (long long)row * K + slice * SL);
#pragma unroll
        for (int j = 0; j < J; ++j) w[r][j] = __builtin_nontemporal_load(wr + j * 64 + lane);
    }
    __builtin_amdgcn_sched_barrier(0);         // keep the whole stream issued before the prologue arithmetic
    if (KS == 1) {                             // destination addresses of the epilogue: behind the weight issue, long before the tail
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) gemv_epi_address<EPI>(a, row0 + r, b, pre[r][b]);
    } else {
        const int t = min(tid, RW * NB - 1);
        gemv_epi_address<EPI>(a, row0 + t / NB, t % NB, pre[0][0]);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- prologue: build the input vector(s) in LDS (PRO_NONE reads them straight into the
    // dot-product register layout below: no staging, no barrier).  The statistics of all NB rows share their barriers: one for
    // the means, one for the variances (round 2 looped over the rows with two barriers each - at NB = 4 the prologue, repeated
    // by every workgroup, cost as much as the weight stream: qkv 7.5 us at one row, 15.8 us at four)
    if (PRO == PRO_EMBED && pro) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < PT; ++i) v[b][i] += v2[b][i];
    }
    if (PRO == PRO_LN) {
        // each reduction has its own LDS slot (same summation order as block_sum)
        float s[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            s[b] = 0.f;
#pragma unroll
            for (int i = 0; i < PT; ++i) s[b] += pro ? v[b][i] : 0.f;
        }
        block_sum_slots<PW, NB>(s, red, 8, pro);
        float s2[NB], mean[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            mean[b] = s[b] / (float)K;
            s2[b] = 0.f;
#pragma unroll
            for (int i = 0; i < PT; ++i) { const float d = (pro ? v[b][i] : 0.f) - mean[b]; s2[b] = fmaf(d, d, s2[b]); }
        }
        block_sum_slots<PW, NB>(s2, red + 4, 8, pro);
        if (pro) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float rstd = 1.0f / sqrtf(s2[b] / (float)K + a.eps);
#pragma unroll
                for (int i = 0; i < PT; ++i) v[b][i] = (v[b][i] - mean[b]) * rstd * lw[i] + lb[i];
            }
        }
    }
    if (PRO != PRO_NONE && pro) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int i = 0; i < PT; ++i) xs[b * K + tid + i * PTPB] = v[b][i];
            if (a.hout != nullptr && blockIdx.x == 0) {
#pragma unroll
                for (int i = 0; i < PT; ++i) a.hout[(long long)b * K + tid + i * PTPB] = v[b][i];
            }
        }
    }
    if (PRO != PRO_NONE) __syncthreads();

    // ---------------- main: dot the (already in flight) weight rows with the input, ONE batch row's slice in registers at a time
    // (round 2 held all NB slices at once: 24 VGPRs per row pushed the 2..4-row launches to half the occupancy, so that the grid
    // no longer fitted the chip in one round - qkv 7.5 us at one row, 15.5 us at four, profiles/r03_batch_table.log).  The
    // arithmetic per (row, batch row) is unchanged: same per-lane fmaf chain, same butterfly.
    float acc[RW][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float* xsrc = (PRO == PRO_NONE) ? a.xin + (long long)b * K + slice * SL : xs + b * K + slice * SL;
        f32x4 xr[J * XV];      // the EPL inputs matching load j are float4 #(j*64+lane)*XV .. +XV of the slice
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int u = 0; u < XV; ++u) xr[j * XV + u] = reinterpret_cast<const f32x4*>(xsrc)[(j * 64 + lane) * XV + u];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) s = dot_w<WT>(w[r][j], &xr[j * XV], s);
            acc[r][b] = wave_sum(s);
        }
    }

    if (KS == 1) {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (row0 + r < a.N) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) gemv_epilogue<EPI>(a, row0 + r, b, acc[r][b], pre[r][b]);
                }
        }
    } else {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int b = 0; b < NB; ++b) red[(r * NB + b) * KS + wid] = acc[r][b];
        }
        __syncthreads();
        if (tid < RW * NB) {
            const int r = tid / NB, b = tid - r * NB;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k) s += red[tid * KS + k];
            if (row0 + r < a.N) gemv_epilogue<EPI>(a, row0 + r, b, s, pre[0][0]);
        }
    }
}

template <typename WT, int KS, int NB, int RW, int PRO, int EPI, int NW = ER_NWAVES>
inline hipError_t launch_gemv(const GemvArgs& a, hipStream_t st) {
    static_assert(KS == 1 || KS == ER_NWAVES, "K is reduced by one wave or by all four");
    constexpr int K = KS * 1536;
    const int rows_per_block = (KS == 1) ? NW * RW : RW;
    const int grid = (a.N + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)((PRO == PRO_NONE ? 0 : NB * K) + 64) * sizeof(float);
    hipLaunchKernelGGL((gemv_kernel<WT, KS, NB, RW, PRO, EPI, NW>), dim3(grid), dim3(64 * NW), lds, st, a.W, a.xin, a.ln_w, a.ln_b, a.bias, a.resid,
                       a.pos, a);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Batched decode (B > 4 rows): the same projections with the weights streamed ONCE per pass of NB
// rows.  The NB input rows (already LayerNorm'd / embedded by prep_rows_kernel) sit in LDS; a wave
// owns RW weight rows, keeps RW*NB accumulators and re-reads the inputs from LDS (ds_read_b128,
// conflict-free) instead of holding them in registers.  Arithmetic per (row, batch-row) is kept
// IDENTICAL to the B <= 4 kernel - same per-lane fmaf chain over the 1536-slice, same xor-32..1
// reduction tree (done as a reduce-scatter so NB sums cost 17 lane exchanges instead of 6*NB), same
// slice order for K = 6144 - so a row of a batch is bit-identical to the same row run alone.
constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }

template <int NB> struct AccRow { float v[NB]; };

template <int NB>
__device__ __forceinline__ float reduce_scatter(AccRow<NB> a, int lane) {   // by value: keeps everything in registers
    float (&v)[NB] = a.v;
    constexpr int STEPS = ilog2(NB);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const int h = NB >> (k + 1), m = 32 >> k;      // compile-time after unrolling
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < NB / 2; ++i)
            if (i < h) {
                const float send = up ? v[i] : v[i + h];
                const float keep = up ? v[i + h] : v[i];
                v[i] = keep + lane_xor_pow2(send, m, lane);
            }
    }
    float r = v[0];
#pragma unroll
    for (int k = STEPS; k < 6; ++k) r += lane_xor_pow2(r, 32 >> k, lane);
    return r;
}

template <int NB>
__device__ __forceinline__ int reduce_scatter_owner(int lane) {   // batch row whose total this lane holds
    constexpr int STEPS = ilog2(NB);
    int b = 0;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) b += (lane & (32 >> k)) ? (NB >> (k + 1)) : 0;
    return b;
}

// grid = ceil(N / (BW*RW)) workgroups of BW = 8 waves (the LDS image limits residency to one workgroup per
// CU, so the workgroup is made as wide as the row supply allows); dynamic LDS = NB*1536 floats.  PH = K / 1536 slices.
constexpr int GB_WAVES = 8, GB_THREADS = GB_WAVES * 64;
template <typename WT, int PH, int NB, int RW, int EPI>
__global__ __launch_bounds__(GB_THREADS) void gemv_batched_kernel(GemvArgs a, int nb_valid) {
    constexpr int EPL = WTraits<WT>::EPL, SL = 1536, J = SL / (64 * EPL), XV = EPL / 4, K = PH * SL;
    constexpr int OWN = 64 / NB;               // lanes per owner group after the reduce-scatter
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [NB][SL]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int row0 = (blockIdx.x * GB_WAVES + wid) * RW;

    // weights: the current 1536-slice in registers, the next slice prefetched while this one is consumed
    const f32x4* wrow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r)
        wrow[r] = reinterpret_cast<const f32x4*>(reinterpret_cast<const WT*>(a.W) + (long long)min(row0 + r, a.N - 1) * K);
    f32x4 wc[RW][J], wn[RW][J];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int j = 0; j < J; ++j) wc[r][j] = __builtin_nontemporal_load(wrow[r] + j * 64 + lane);
    const int b_own = reduce_scatter_owner<NB>(lane);
    const bool owner = (lane & (OWN - 1)) == 0 && b_own < nb_valid;
    EpiPre pre[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) pre[r] = gemv_epi_prefetch<EPI>(a, row0 + r, min(b_own, nb_valid - 1));

    // input staging: each thread moves FILL float4 of the [NB][1536] slice image; the loads of slice ph+1 are
    // issued before slice ph is consumed and land in registers while it is being multiplied
    constexpr int PER_ROW = SL / 4, FILL = NB * PER_ROW / GB_THREADS;   // 12 at NB = 16
    static_assert(NB * PER_ROW % GB_THREADS == 0, "fill loop shape");
    f32x4 xf[FILL];
    auto issue_fill = [&](int ph) {
#pragma unroll
        for (int u = 0; u < FILL; ++u) {
            const int i = tid + u * GB_THREADS;
            const int b = i / PER_ROW, c = i - b * PER_ROW;
            // unconditional load from a clamped row (a per-element branch would serialise the loads);
            // rows >= nb_valid replicate the last valid row, their accumulators are never stored
            xf[u] = reinterpret_cast<const f32x4*>(a.xin + (long long)min(b, nb_valid - 1) * K + ph * SL)[c];
        }
    };
    issue_fill(0);

    float total[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) total[r] = 0.f;
#pragma unroll 1
    for (int ph = 0; ph < PH; ++ph) {      // a real loop: unrolling the phases lets the scheduler pile up 4x the live values
        if (ph > 0) __syncthreads();       // readers of the previous slice image are done
#pragma unroll
        for (int u = 0; u < FILL; ++u) reinterpret_cast<f32x4*>(xs)[tid + u * GB_THREADS] = xf[u];
        __syncthreads();
        if (ph + 1 < PH) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int j = 0; j < J; ++j) wn[r][j] = __builtin_nontemporal_load(wrow[r] + ((ph + 1) * J + j) * 64 + lane);
            issue_fill(ph + 1);
        }
        AccRow<NB> acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[r].v[b] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 x[XV];
#pragma unroll
                for (int u = 0; u < XV; ++u) x[u] = reinterpret_cast<const f32x4*>(xs + b * SL)[(j * 64 + lane) * XV + u];
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r].v[b] = dot_w<WT>(wc[r][j], x, acc[r].v[b]);
            }
#pragma unroll
        for (int r = 0; r < RW; ++r) total[r] += reduce_scatter<NB>(acc[r], lane);
        if (ph + 1 < PH) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int j = 0; j < J; ++j) wc[r][j] = wn[r][j];
        }
    }
    if (owner) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (row0 + r < a.N) gemv_epilogue<EPI>(a, row0 + r, b_own, total[r], pre[r]);
    }
}

template <typename WT, int PH, int NB, int RW, int EPI>
inline hipError_t launch_gemv_batched(const GemvArgs& a, int nb_valid, hipStream_t st) {
    const int grid = (a.N + GB_WAVES * RW - 1) / (GB_WAVES * RW);
    const size_t lds = (size_t)NB * 1536 * sizeof(float);
    static bool configured = false;
    if (!configured) {   // > 64 KiB of dynamic LDS needs an explicit opt-in
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_batched_kernel<WT, PH, NB, RW, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL((gemv_batched_kernel<WT, PH, NB, RW, EPI>), dim3(grid), dim3(GB_THREADS), lds, st, a, nb_valid);
    return hipGetLastError();
}

// ((p_0 + p_1 + ... + p_{S-1}) + bias) + resid for the PT elements of a thread: every load first, then the adds in slice order
template <int S, int PT>
__device__ __forceinline__ void sk_finish_row(const float* p0, long long slice_stride, const float* bias, const float* resid, int tid,
                                              float (&v)[PT]) {
    float p[S][PT], bs[PT], rs[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
#pragma unroll
        for (int k = 0; k < S; ++k) p[k][i] = p0[(long long)k * slice_stride + tid + i * ER_WG];
        bs[i] = bias[tid + i * ER_WG];
        rs[i] = resid[tid + i * ER_WG];
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < S; ++k) s += p[k][i];
        s += bs[i];
        v[i] = s + rs[i];
    }
}

// One workgroup per batch row: the LayerNorm / embedding prologue of gemv_kernel as its own kernel
// (identical thread->element mapping and reduction order), writing the GEMV input / residual row.
template <int PRO>
__global__ __launch_bounds__(ER_WG) void prep_rows_kernel(GemvArgs a) {
    constexpr int K = 1536, PT = K / ER_WG;
    __shared__ float red[8];
    const int tid = threadIdx.x, b = blockIdx.x;
    float v[PT];
    if (PRO == PRO_EMBED) {
        const float* e = a.embd + (long long)a.tok[b] * K;
        const float* p = a.posemb + (long long)a.pos[b] * K;
#pragma unroll
        for (int i = 0; i < PT; ++i) v[i] = e[tid + i * ER_WG] + p[tid + i * ER_WG];
    } else {
        float lw[PT], lb[PT];                  // loaded with the row, not behind the two reductions (a dependent round trip per launch)
        if (a.sk_part) {
            // the previous layer's fc2 left its four K-range partials unfinished (23 of 24 splitk_finish launches per token removed):
            // every load of the row goes out at once, the adds keep splitk_finish_kernel's / gemv_epilogue's order
            const int g = b >> 5, nbg = min(32, a.sk_batch - 32 * g);
            const int S = a.sk_slices == 16 ? 16 : 4;
            const float* p0 = a.sk_part + ((long long)g * S * 32 + (b & 31)) * K;       // slice s of the group: + s * nbg * K
#pragma unroll
            for (int i = 0; i < PT; ++i) { lw[i] = a.ln_w[tid + i * ER_WG]; lb[i] = a.ln_b[tid + i * ER_WG]; }
            if (S == 16) sk_finish_row<16, PT>(p0, (long long)nbg * K, a.sk_bias, a.sk_resid + (long long)b * K, tid, v);
            else sk_finish_row<4, PT>(p0, (long long)nbg * K, a.sk_bias, a.sk_resid + (long long)b * K, tid, v);
        } else {
            const float* x = a.xin + (long long)b * K;
#pragma unroll
            for (int i = 0; i < PT; ++i) { v[i] = x[tid + i * ER_WG]; lw[i] = a.ln_w[tid + i * ER_WG]; lb[i] = a.ln_b[tid + i * ER_WG]; }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PT; ++i) s += v[i];
        const float mean = block_sum(s, red) / (float)K;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PT; ++i) { const float d = v[i] - mean; s2 = fmaf(d, d, s2); }
        const float var = block_sum(s2, red) / (float)K;
        const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
        for (int i = 0; i < PT; ++i) v[i] = (v[i] - mean) * rstd * lw[i] + lb[i];
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) a.hout[(long long)b * K + tid + i * ER_WG] = v[i];
    if (a.xt_out) {        // the same row in the matrix cores' operand layout: through LDS, so that a thread holds four consecutive k
        __shared__ __attribute__((aligned(16))) float row[K];
#pragma unroll
        for (int i = 0; i < PT; ++i) row[tid + i * ER_WG] = v[i];
        __syncthreads();
        xt_h8* xt = reinterpret_cast<xt_h8*>(a.xt_out);
        for (int k4 = tid; k4 < K / 4; k4 += ER_WG) {
            const f32x4 t = reinterpret_cast<const f32x4*>(row)[k4];
            xt[xt_entry(K, b, k4)] = xt_pack(t.x, t.y, t.z, t.w);
        }
    }
}

}  // namespace er
