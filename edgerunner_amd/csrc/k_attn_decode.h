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
#include "k_gemv.h"

// timeline hooks of scripts/probes/attn_timeline_probe.hip (expand to nothing in the library build)
#ifndef ER_TP
#define ER_TP(i)
#endif

namespace er {

struct AttnDecArgs {
    const float* q;        // [B][hidden]
    const void* kcache;    // [B][H][Lcap][D], fp32 or fp16 (the kernel's KT)
    const void* vcache;
    const int* pos;        // device, per row: index of the newest key (len = pos+1); or
    int fixed_len;         // >0: use this length for every row instead of pos
    const int* len_dev;    // optional per-row lengths (overrides pos when non-null)
    float* part;           // [B][H][S][D+2] = {m, l, o[0..D)}   (version 3: [B][H][NCH][D], o rows only)
    float* part_ml;        // version 3: [B][H][NCH][2] = {m, l}
    float* out;            // combine: [B][hidden]
    int H, l_cap, S, hidden;   // S = number of chunks the grid covers = ceil(l_cap / chunk)
    int chunk;             // keys per workgroup (fp32: 32*STEPS, fp16: 64*STEPS with half the steps)
    long long kv_bstride;
    float sqrt_d;          // sqrt(D): scores are divided by it, as the reference does
    const int* len_src;    // filled by the version-3 launcher: len = len_src[b] + len_add (len_dev + 0, or pos + 1) unless fixed_len > 0
    int len_add;
    void* out_xt;          // streaming kernel: also write the output row in the tiled hi | lo layout of k_gemv.h xt_entry (out_proj's input)
};

__device__ __forceinline__ int attn_len(const AttnDecArgs& a, int b) {
    if (a.len_dev) return a.len_dev[b];
    if (a.fixed_len > 0) return a.fixed_len;
    return a.pos[b] + 1;
}

// 16 bytes of a key row as EPL floats
template <typename KT> struct KVec;
template <> struct KVec<float> { static constexpr int EPL = 4, LPK = 8; };      // 8 lanes x NV float4 per key row
template <> struct KVec<_Float16> { static constexpr int EPL = 8, LPK = 4; };   // 4 lanes x NV (8 x fp16) per key row

template <typename KT>
__device__ __forceinline__ void kv_unpack(const f32x4& raw, float (&f)[KVec<KT>::EPL]) {
    if constexpr (sizeof(KT) == 4) {
        f[0] = raw.x; f[1] = raw.y; f[2] = raw.z; f[3] = raw.w;
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 h = __builtin_bit_cast(h8, raw);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
    }
}

// ---- fp16 cache: lanes mapped over PAIRS of key rows (round 5).
//
// A row of the fp16 cache is 96 halves = 192 bytes = one and a half 128-byte lines.  The first fp16 mapping gave a row to 4 lanes x
// 3 loads, so load j of a wave read the j-th 64-byte THIRD of 16 rows: sixteen half-used lines per wave-instruction, every line
// requested by two instructions - 12 % below the rate of whole-line requests at the same bytes (5.8 vs 5.15 us for 25 MB, 6.3 vs
// 7.2 TB/s incremental: scripts/probes/attn_load_pattern_probe.hip, profiles/r05_attn_load_pattern_probe.log; the fp32 cache's
// 384-byte rows already are whole lines and measure like a contiguous stream).  Two consecutive rows starting at an EVEN key are
// 384 bytes = three whole lines, so 8 lanes x 3 loads take a pair (A, B): lane p of the group reads the p-th 16 bytes of line j.
// In units of 8-half chunks ("parts" 0..11 of a row):
//     load 0: row A part p          load 1: p < 4 ? row A part 8 + p : row B part p - 4          load 2: row B part 4 + p
// Scores: a = t0 + (p < 4 ? t1 : 0), b = (p < 4 ? 0 : t1) + t2 summed over the 8 lanes; weights: pA on load 0, pA / pB on load 1,
// pB on load 2.  Every part is accumulated by two (lane, load) slots 4 lanes apart; hpair_fold() adds them with row_ror:4.
struct HPair {
    int p; bool lo; int part1;
    __device__ __forceinline__ explicit HPair(int lane) : p(lane & 7), lo((lane & 7) < 4), part1((lane & 7) < 4 ? 8 + (lane & 7) : (lane & 7) - 4) {}
};

__device__ __forceinline__ void hpair_load_q(const float* qp, const HPair& hp, float (&qv)[3][8]) {
    const int part[3] = {hp.p, hp.part1, 4 + hp.p};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + part[j] * 8 + e);
            qv[j][e] = t.x; qv[j][e + 1] = t.y; qv[j][e + 2] = t.z; qv[j][e + 3] = t.w;
        }
}
// rA / rB: the rows this lane reads for key A / key B of its pair (a masked key reads a valid row: its weight is 0)
__device__ __forceinline__ void hpair_load(const _Float16* base, int rA, int rB, const HPair& hp, f32x4 (&r)[3]) {
    r[0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (long long)rA * 96 + hp.p * 8));
    r[1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (long long)(hp.lo ? rA : rB) * 96 + hp.part1 * 8));
    r[2] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (long long)rB * 96 + (4 + hp.p) * 8));
}
__device__ __forceinline__ float hpair_dot8(const f32x4& raw, const float (&q)[8]) {
    float f[8];
    kv_unpack<_Float16>(raw, f);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(q[e], f[e], acc);
    return acc;
}
__device__ __forceinline__ void hpair_scores(const f32x4 (&k)[3], const float (&qv)[3][8], const HPair& hp, float& sA, float& sB) {
    const float t0 = hpair_dot8(k[0], qv[0]), t1 = hpair_dot8(k[1], qv[1]), t2 = hpair_dot8(k[2], qv[2]);
    sA = lane_group_sum<8>(t0 + (hp.lo ? t1 : 0.f));
    sB = lane_group_sum<8>((hp.lo ? 0.f : t1) + t2);
}
__device__ __forceinline__ void hpair_accum(const f32x4 (&v)[3], float pA, float pB, const HPair& hp, float (&o)[3][8]) {
    const float w[3] = {pA, hp.lo ? pA : pB, pB};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float f[8];
        kv_unpack<_Float16>(v[j], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[j][e] = fmaf(w[j], f[e], o[j][e]);
    }
}
// o[j][e] already summed over the two lane groups of the row of 16 (o += row_ror<8>(o)): main = part p (all 8 lanes), hi = part 8 + p
// (lanes p < 4 only)
__device__ __forceinline__ void hpair_fold(const float (&o)[3][8], const HPair& hp, float (&main)[8], float (&hi)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        main[e] = o[0][e] + row_ror<4>(hp.lo ? o[2][e] : o[1][e]);     // a lane hands its receiver (4 lanes below) what THAT lane lacks
        hi[e] = o[1][e] + row_ror<4>(o[2][e]);
    }
}

// grid (S, H, B), 256 threads.  LPK lanes cover one key row with NV 16-byte loads each (full 128-byte
// lines per wave-instruction); a wave step covers 64/LPK keys, the workgroup chunk is 4*STEPS*(64/LPK) keys:
// wave w, step i, lane group g -> key k0 + KPS*i + KPW*w + g.
template <typename KT, int D, int STEPS>
__global__ __launch_bounds__(ER_WG) void attn_decode_kernel(AttnDecArgs a) {
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK;
    constexpr int NV = D / (EPL * LPK);        // 16-byte loads per lane per key
    constexpr int KPW = 64 / LPK;              // keys per wave step
    constexpr int KPS = ER_NWAVES * KPW;       // keys per workgroup step
    constexpr int CHUNK = KPS * STEPS;
    static_assert(D % (EPL * LPK) == 0, "head_dim must tile into 16-byte loads");
    __shared__ __attribute__((aligned(16))) float ored[ER_NWAVES * D];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & (LPK - 1), g = lane / LPK;
    // heads fastest: workgroup id = h + H*s lands on XCD h % 8, so every XCD gets the same number of ACTIVE chunks.
    // With chunks fastest and S a multiple of 8, chunk s always lands on XCD s % 8: at 33 active chunks XCD 0 works on 5
    // chunks per head while the others hold 4 - the 137 -> 167 us jump of the batch-32 sweep between L = 3926 and 4176.
    const int h = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const int len = attn_len(a, b);
    const int k0 = s * CHUNK;
    if (k0 >= len) return;                    // inactive chunk: the merge only visits ceil(len/CHUNK) partials
    const int k1 = min(len, k0 + CHUNK);
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;
    const KT* kb = reinterpret_cast<const KT*>(a.kcache) + head_off;
    const KT* vb = reinterpret_cast<const KT*>(a.vcache) + head_off;

    if constexpr (sizeof(KT) == 2 && D == 96) {
        // fp16 cache, 192-byte rows: the PAIR mapping (whole 128-byte lines, see above); k0 is a multiple of the chunk = even
        const HPair hp(lane);
        const int g8 = lane >> 3;
        f32x4 kreg[STEPS][3], vreg[STEPS][3];
        bool vA[STEPS], vB[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = k0 + KPS * i + KPW * wid + 2 * g8;
            vA[i] = kA < k1;
            vB[i] = kA + 1 < k1;
            hpair_load(kb, vA[i] ? kA : k0, vB[i] ? kA + 1 : k0, hp, kreg[i]);
        }
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = k0 + KPS * i + KPW * wid + 2 * g8;
            hpair_load(vb, vA[i] ? kA : k0, vB[i] ? kA + 1 : k0, hp, vreg[i]);
        }
        float qh[3][8];
        hpair_load_q(a.q + (long long)b * a.hidden + h * D, hp, qh);
        float scA[STEPS], scB[STEPS];
        float mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            float sA, sB;
            hpair_scores(kreg[i], qh, hp, sA, sB);
            scA[i] = vA[i] ? sA / a.sqrt_d : -INFINITY;
            scB[i] = vB[i] ? sB / a.sqrt_d : -INFINITY;
            mloc = fmaxf(mloc, fmaxf(scA[i], scB[i]));
        }
        const float m = block_max(mloc, red);     // finite: the chunk holds at least one key
        float oh[3][8];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[j][e] = 0.f;
        float lloc = 0.f;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const float pA = vA[i] ? expf(scA[i] - m) : 0.f, pB = vB[i] ? expf(scB[i] - m) : 0.f;
            if (hp.p == 0) lloc += pA + pB;
            hpair_accum(vreg[i], pA, pB, hp, oh);
        }
        const float l = block_sum(lloc, red);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[j][e] = across_groups_sum<8>(oh[j][e]);        // every 8-lane group now holds the wave's totals
        float om[8], ohi[8];
        hpair_fold(oh, hp, om, ohi);
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ored[wid * D + hp.p * 8 + e] = om[e];
            if (hp.lo) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ored[wid * D + (8 + hp.p) * 8 + e] = ohi[e];
            }
        }
        __syncthreads();
        float* pout = a.part + (((long long)b * a.H + h) * a.S + s) * (D + 2);
        if (tid < D) pout[2 + tid] = (ored[tid] + ored[D + tid]) + (ored[2 * D + tid] + ored[3 * D + tid]);
        if (tid == 0) { pout[0] = m; pout[1] = l; }
        return;
    }
    // ---- all loads first: K rows, then V rows (K returns first, V lands while the scores are reduced)
    f32x4 kreg[STEPS][NV], vreg[STEPS][NV];
    bool valid[STEPS];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int kk = k0 + KPS * i + KPW * wid + g;
        valid[i] = kk < k1;
        const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) kreg[i][j] = kr[j * LPK + p];
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int kk = k0 + KPS * i + KPW * wid + g;
        const f32x4* vr = reinterpret_cast<const f32x4*>(vb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) vreg[i][j] = vr[j * LPK + p];
    }
    // this lane's query elements: dims (j*LPK + p)*EPL .. +EPL
    float qv[NV][EPL];
    const float* qp = a.q + (long long)b * a.hidden + h * D;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + (j * LPK + p) * EPL + e);
            qv[j][e] = t.x; qv[j][e + 1] = t.y; qv[j][e + 2] = t.z; qv[j][e + 3] = t.w;
        }

    // ---- scores q.k / sqrt(D) (every lane of a key's lane group ends up holding its score)
    float sc[STEPS];
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float kf[EPL];
            kv_unpack<KT>(kreg[i][j], kf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc = fmaf(qv[j][e], kf[e], acc);
        }
        acc = lane_group_sum<LPK>(acc);
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
    float o[NV][EPL];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float vf[EPL];
            kv_unpack<KT>(vreg[i][j], vf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[j][e] = fmaf(pw[i], vf[e], o[j][e]);
        }
    // sum over the key groups of the wave (lanes with equal p), then over the 4 waves
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            o[j][e] = across_groups_sum<LPK>(o[j][e]);
        }
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) ored[wid * D + (j * LPK + p) * EPL + e] = o[j][e];
    }
    __syncthreads();
    float* pout = a.part + (((long long)b * a.H + h) * a.S + s) * (D + 2);
    if (tid < D) pout[2 + tid] = (ored[tid] + ored[D + tid]) + (ored[2 * D + tid] + ored[3 * D + tid]);
    if (tid == 0) { pout[0] = m; pout[1] = l; }
}

// ---- version 2 of the partial kernel and the merge kernel (the round-1 merge, a serial walk over the partials, is gone).
//
// attn_decode2_kernel: same work decomposition and loads as attn_decode_kernel, but every WAVE runs its own softmax
// (max / sum through shuffles only) and the four waves are merged once through LDS - one workgroup barrier on the
// critical path instead of six (block_max + block_sum + the output combine).
template <typename KT, int D, int STEPS>
__global__ __launch_bounds__(ER_WG) void attn_decode2_kernel(AttnDecArgs a) {
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK;
    constexpr int NV = D / (EPL * LPK);
    constexpr int KPW = 64 / LPK;
    constexpr int KPS = ER_NWAVES * KPW;
    constexpr int CHUNK = KPS * STEPS;
    static_assert(D % (EPL * LPK) == 0, "head_dim must tile into 16-byte loads");
    __shared__ __attribute__((aligned(16))) float ored[ER_NWAVES * 4 * D];     // [wave][row of 16 lanes][dim]
    __shared__ float wm[ER_NWAVES], wl[ER_NWAVES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & (LPK - 1), g = lane / LPK;
    const int h = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const int len = attn_len(a, b);
    const int k0 = s * CHUNK;
    if (k0 >= len) return;
    const int k1 = min(len, k0 + CHUNK);
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;
    const KT* kb = reinterpret_cast<const KT*>(a.kcache) + head_off;
    const KT* vb = reinterpret_cast<const KT*>(a.vcache) + head_off;

    if constexpr (sizeof(KT) == 2 && D == 96) {
        // fp16 cache, 192-byte rows: the PAIR mapping (whole 128-byte lines, see above); k0 is a multiple of the chunk = even
        const HPair hp(lane);
        const int g8 = lane >> 3;
        float qh[3][8];
        hpair_load_q(a.q + (long long)b * a.hidden + h * D, hp, qh);
        f32x4 kreg[STEPS][3], vreg[STEPS][3];
        bool vA[STEPS], vB[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = k0 + KPS * i + KPW * wid + 2 * g8;
            vA[i] = kA < k1;
            vB[i] = kA + 1 < k1;
            hpair_load(kb, vA[i] ? kA : k0, vB[i] ? kA + 1 : k0, hp, kreg[i]);
        }
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = k0 + KPS * i + KPW * wid + 2 * g8;
            hpair_load(vb, vA[i] ? kA : k0, vB[i] ? kA + 1 : k0, hp, vreg[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        float scA[STEPS], scB[STEPS];
        float mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            float sA, sB;
            hpair_scores(kreg[i], qh, hp, sA, sB);
            scA[i] = vA[i] ? sA / a.sqrt_d : -INFINITY;
            scB[i] = vB[i] ? sB / a.sqrt_d : -INFINITY;
            mloc = fmaxf(mloc, fmaxf(scA[i], scB[i]));
        }
        const float m = wave_max(mloc);           // -inf when the wave holds no valid key
        float oh[3][8];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[j][e] = 0.f;
        float lloc = 0.f;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const float pA = vA[i] ? expf(scA[i] - m) : 0.f, pB = vB[i] ? expf(scB[i] - m) : 0.f;
            if (hp.p == 0) lloc += pA + pB;
            hpair_accum(vreg[i], pA, pB, hp, oh);
        }
        const float l = wave_sum(lloc);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[j][e] += row_ror<8>(oh[j][e]);
        float om[8], ohi[8];
        hpair_fold(oh, hp, om, ohi);
        if ((lane & 15) < 8) {
            float* dst = ored + (wid * 4 + (lane >> 4)) * D;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[hp.p * 8 + e] = om[e];
            if (hp.lo) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[(8 + hp.p) * 8 + e] = ohi[e];
            }
        }
        if (lane == 0) { wm[wid] = m; wl[wid] = l; }
    } else {
    // this lane's query elements first: vector loads return in issue order, so anything issued behind the K/V
        // stream would only become usable after ALL of it has landed
        float qv[NV][EPL];
        const float* qp = a.q + (long long)b * a.hidden + h * D;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(qp + (j * LPK + p) * EPL + e);
                qv[j][e] = t.x; qv[j][e + 1] = t.y; qv[j][e + 2] = t.z; qv[j][e + 3] = t.w;
            }
        f32x4 kreg[STEPS][NV], vreg[STEPS][NV];
        bool valid[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kk = k0 + KPS * i + KPW * wid + g;
            valid[i] = kk < k1;
            const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
            for (int j = 0; j < NV; ++j) kreg[i][j] = __builtin_nontemporal_load(kr + j * LPK + p);
        }
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kk = k0 + KPS * i + KPW * wid + g;
            const f32x4* vr = reinterpret_cast<const f32x4*>(vb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
            for (int j = 0; j < NV; ++j) vreg[i][j] = __builtin_nontemporal_load(vr + j * LPK + p);
        }
        __builtin_amdgcn_sched_barrier(0);        // every K and V load is issued before the first score is computed

        float sc[STEPS];
        float mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float kf[EPL];
                kv_unpack<KT>(kreg[i][j], kf);
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc = fmaf(qv[j][e], kf[e], acc);
            }
            acc = lane_group_sum<LPK>(acc);
            sc[i] = valid[i] ? acc / a.sqrt_d : -INFINITY;
            mloc = fmaxf(mloc, sc[i]);
        }
        const float m = wave_max(mloc);           // -inf when the wave holds no valid key (ragged tail of the last chunk)

        float pw[STEPS];
        float lloc = 0.f;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            pw[i] = valid[i] ? expf(sc[i] - m) : 0.f;
            if (p == 0) lloc += pw[i];
        }
        const float l = wave_sum(lloc);

        float o[NV][EPL];
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[j][e] = 0.f;
#pragma unroll
        for (int i = 0; i < STEPS; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float vf[EPL];
                kv_unpack<KT>(vreg[i][j], vf);
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[j][e] = fmaf(pw[i], vf[e], o[j][e]);
            }
        // key groups sit LPK lanes apart: add them inside each row of 16 lanes with DPP rotations (VALU, no LDS crossbar), and
        // leave the four rows of a wave to the LDS merge below (round 2a did all of it with 36 / 96 ds_bpermutes per lane)
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                o[j][e] += row_ror<8>(o[j][e]);
                if (LPK == 4) o[j][e] += row_ror<4>(o[j][e]);
            }
        if ((lane & 15) < LPK) {
            float* dst = ored + (wid * 4 + (lane >> 4)) * D;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < EPL; ++e) dst[(j * LPK + p) * EPL + e] = o[j][e];
        }
        if (lane == 0) { wm[wid] = m; wl[wid] = l; }
    }
    __syncthreads();
    if (tid < D) {
        const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));   // finite: the chunk holds at least one key
        float ov = 0.f, lv = 0.f;
#pragma unroll
        for (int k = 0; k < ER_NWAVES; ++k) {
            const float w = (wm[k] == -INFINITY) ? 0.f : expf(wm[k] - M);
            const float* src = ored + k * 4 * D + tid;
            ov = fmaf((src[0] + src[D]) + (src[2 * D] + src[3 * D]), w, ov);
            lv = fmaf(wl[k], w, lv);
        }
        float* pout = a.part + (((long long)b * a.H + h) * a.S + s) * (D + 2);
        pout[2 + tid] = ov;
        if (tid == 0) { pout[0] = M; pout[1] = lv; }
    }
}

// attn_combine2_kernel, grid (H, B), 256 threads: one memory round trip.  Thread (c, half) issues ALL of its column loads
// (partials half, half+2, ... of a pass) before anything else; every wave then redundantly loads the pass's {m_s, l_s}
// with lane = s, so the global max / weights / sum need wave shuffles only (no LDS, no barrier), and the weight of
// partial s reaches the column accumulation through a wave-uniform readlane.  A pass covers up to 64 partials; its
// unrolled load count is picked from the number of partials actually left (8 / 16 / 24 / 32 per thread) so that short
// contexts do not wait for redundant loads.  Longer contexts (> 64 partials, L > 8192) run further passes with the
// usual running-max rescale.
template <int D, int PER>
__device__ __forceinline__ void combine_pass(const float* pb, int base, int n_act, int cc, int half, int lane, float& M_run,
                                             float& l_run, float& o_run) {
    constexpr int W = D + 2;
    float v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int sidx = min(base + half + 2 * u, n_act - 1);   // clamped: loads stay unconditional
        v[u] = pb[sidx * W + 2 + cc];
    }
    const int sl = base + lane;
    const bool act = sl < n_act && lane < 2 * PER;
    const float m_s = act ? pb[sl * W] : -INFINITY;
    const float l_s = act ? pb[sl * W + 1] : 0.f;
    const float M_new = fmaxf(M_run, wave_max(m_s));
    const float w = act ? expf(m_s - M_new) : 0.f;
    const float l_blk = wave_sum(l_s * w);
    const float alpha = (M_run == -INFINITY) ? 0.f : expf(M_run - M_new);
    float o = o_run * alpha;
    const int hs = __builtin_amdgcn_readfirstlane(half);         // wave-uniform: the weights come through v_readlane, not the LDS crossbar
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const float wu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), hs + 2 * u));   // 0 for partials beyond n_act
        o = fmaf(v[u], wu, o);
    }
    o_run = o;
    l_run = l_run * alpha + l_blk;
    M_run = M_new;
}

template <int D>
__global__ __launch_bounds__(ER_WG) void attn_combine2_kernel(AttnDecArgs a) {
    const int CHUNK = a.chunk;
    constexpr int W = D + 2;
    __shared__ float half1[128];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int n_act = (attn_len(a, b) + CHUNK - 1) / CHUNK;
    const float* pb = a.part + ((long long)b * a.H + h) * a.S * W;
    const int c = tid & 127, half = tid >> 7;          // half is wave-uniform (waves 0,1 -> 0; waves 2,3 -> 1)
    const int cc = min(c, D - 1);
    float M_run = -INFINITY, l_run = 0.f, o_run = 0.f;
    int base = 0;
    while (base < n_act) {
        const int rem = n_act - base;
        if (rem <= 16) { combine_pass<D, 8>(pb, base, n_act, cc, half, lane, M_run, l_run, o_run); base += 16; }
        else if (rem <= 32) { combine_pass<D, 16>(pb, base, n_act, cc, half, lane, M_run, l_run, o_run); base += 32; }
        else if (rem <= 48) { combine_pass<D, 24>(pb, base, n_act, cc, half, lane, M_run, l_run, o_run); base += 48; }
        else { combine_pass<D, 32>(pb, base, n_act, cc, half, lane, M_run, l_run, o_run); base += 64; }
    }
    if (half == 1) half1[c] = o_run;
    __syncthreads();
    if (half == 0 && c < D) a.out[(long long)b * a.hidden + h * D + c] = (o_run + half1[c]) / l_run;
}

// ---- version 3 (ER_DECODE_V=3, single row): BALANCED chunks + 16-byte aligned partials for the fused merge/out_proj kernel.
//
// v2 cuts the keys into fixed 128-key chunks, so the number of workgroups per head grows with the context and the kernel's
// time is set by the CUs that happen to hold one workgroup more than the others (the 8.4 -> 10.3 -> 11.9 us staircase of
// profiles/r02_bench.json).  Here the grid is ALWAYS (H, NCH): every head is cut into NCH equal chunks of ceil(len / NCH)
// keys and one NW-wave workgroup takes a whole chunk (up to STEPS * NW * KPW keys), so with H * NCH = 256 every CU holds
// exactly one workgroup at every context length and the merge always sees NCH partials per head.  The wave-step that holds
// key group i*NW + w is step-major, so the number of active steps is uniform over the workgroup and is dispatched OUTSIDE the
// unrolled load block (a per-load condition would make hipcc branch around - and wait for - every load).
// Partials: o rows [B][H][NCH][D] (16-byte aligned float4 columns) and {m, l} pairs [B][H][NCH][2].
constexpr int A3_LD = 97;     // row stride of the wave partials in LDS (D + 1: lanes that read a column down the rows hit 32 different banks)

template <int D, int NW>
__device__ __forceinline__ void attn3_merge_rows(float* ored, const float* wm, const float* wl, float* po, float* pml);

template <typename KT, int D, int NS, int NW>
__device__ __forceinline__ void attn3_body(const AttnDecArgs& a, const KT* kb, const KT* vb,
                                           const float (&qv)[D / (KVec<KT>::EPL * KVec<KT>::LPK)][KVec<KT>::EPL], int k0, int k1,
                                           float* ored, float* wm, float* wl, float* po, float* pml) {
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK;
    constexpr int NV = D / (EPL * LPK);
    constexpr int KPW = 64 / LPK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & (LPK - 1), g = lane / LPK;
    f32x4 kreg[NS][NV], vreg[NS][NV];
    bool valid[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int kk = k0 + (i * NW + wid) * KPW + g;
        valid[i] = kk < k1;
        const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) kreg[i][j] = __builtin_nontemporal_load(kr + j * LPK + p);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int kk = k0 + (i * NW + wid) * KPW + g;
        const f32x4* vr = reinterpret_cast<const f32x4*>(vb + (long long)(valid[i] ? kk : k0) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) vreg[i][j] = __builtin_nontemporal_load(vr + j * LPK + p);
    }
    __builtin_amdgcn_sched_barrier(0);
    ER_TP(2);

    float sc[NS];
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float kf[EPL];
            kv_unpack<KT>(kreg[i][j], kf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc = fmaf(qv[j][e], kf[e], acc);
        }
        acc = lane_group_sum<LPK>(acc);
        sc[i] = valid[i] ? acc / a.sqrt_d : -INFINITY;
        mloc = fmaxf(mloc, sc[i]);
    }
    const float m = wave_max(mloc);           // -inf when the wave holds no valid key
    ER_TP(3);
    float pw[NS];
    float lloc = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        pw[i] = valid[i] ? expf(sc[i] - m) : 0.f;
        if (p == 0) lloc += pw[i];
    }
    const float l = wave_sum(lloc);
    float o[NV][EPL];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float vf[EPL];
            kv_unpack<KT>(vreg[i][j], vf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[j][e] = fmaf(pw[i], vf[e], o[j][e]);
        }
    ER_TP(4);
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            o[j][e] += row_ror<8>(o[j][e]);
            if (LPK == 4) o[j][e] += row_ror<4>(o[j][e]);
        }
    if ((lane & 15) < LPK) {
        float* dst = ored + (wid * 4 + (lane >> 4)) * A3_LD;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) dst[(j * LPK + p) * EPL + e] = o[j][e];
    }
    if (lane == 0) { wm[wid] = m; wl[wid] = l; }
    __syncthreads();
    ER_TP(5);
    attn3_merge_rows<D, NW>(ored, wm, wl, po, pml);
    ER_TP(7);
}

template <int D, int NW>
__device__ __forceinline__ void attn3_merge_rows(float* ored, const float* wm, const float* wl, float* po, float* pml) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // ONE barrier: the 4 * NW = 64 row partials {wave k, row r of 16 lanes} of a column meet in the 64 lanes of ONE wave - wave w
    // owns columns [w * D/NW, (w + 1) * D/NW), lane = 4k + r reads them from row (k, r) (row stride A3_LD = D + 1 floats: conflict
    // free), weighs them with exp(m_k - M) and the wave adds them up: four row_ror steps inside each row of 16 lanes, then the four
    // row totals through readlane in a fixed order.  (Round 2 used two barriers and two LDS round trips: 0.75 us of tail per launch
    // in profiles/r03_attn_timeline.log, against ~0.25 us here.)
    constexpr int CPW = D / NW;                // columns per wave
    const int k = lane >> 2;
    const float mk = wm[k], lk = wl[k];
    const float* src = ored + lane * A3_LD + wid * CPW;
    float acc[CPW + 1];
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc[i] = src[i];
    // maximum over the 16 waves: two rotations inside each row of 16 lanes (4 waves per row), then the classic row_bcast steps
    // (row 1 <- row 0, row 3 <- row 2, rows 2..3 <- row 1), the result read back from lane 63
    float mx = fmaxf(mk, row_ror<4>(mk));
    mx = fmaxf(mx, row_ror<8>(mx));
    mx = fmaxf(mx, row_bcast<0x142, 0xa>(mx, -INFINITY));
    mx = fmaxf(mx, row_bcast<0x143, 0xc>(mx, -INFINITY));
    const float M = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 63));     // finite: the chunk holds at least one key
    const float w = (mk == -INFINITY) ? 0.f : expf(mk - M);
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc[i] *= w;
    acc[CPW] = (lane & 3) == 0 ? lk * w : 0.f;
#pragma unroll
    for (int i = 0; i <= CPW; ++i) {           // CPW + 1 independent chains: the totals end up in lanes 48..63
        float v = acc[i];
        v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
        v += row_bcast<0x142, 0xa>(v, 0.f);
        v += row_bcast<0x143, 0xc>(v, 0.f);
        acc[i] = v;
    }
    if (lane >= 48 && lane < 48 + CPW) {
        float v = acc[0];
#pragma unroll
        for (int i = 1; i < CPW; ++i) v = (lane == 48 + i) ? acc[i] : v;
        po[wid * CPW + lane - 48] = v;
    }
    if (tid == 63) { pml[0] = M; pml[1] = acc[CPW]; }
}
// the fp16-cache body of the balanced kernel on the PAIR mapping above (k0 even: the chunk length of the fp16 kernel is rounded up to
// an even number of keys); everything behind the row partials - the one-barrier merge of the 4 * NW rows - is shared with the fp32 body
template <int D, int NS, int NW>
__device__ __forceinline__ void attn3_body_h(const AttnDecArgs& a, const _Float16* kb, const _Float16* vb, const float (&qv)[3][8], int k0, int k1,
                                             float* ored, float* wm, float* wl, float* po, float* pml) {
    static_assert(D == 96, "pair mapping: 96 halves per row");
    constexpr int KPW = 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const HPair hp(lane);
    const int g = lane >> 3;
    f32x4 kreg[NS][3], vreg[NS][3];
    bool vA[NS], vB[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int kA = k0 + (i * NW + wid) * KPW + 2 * g;
        vA[i] = kA < k1;
        vB[i] = kA + 1 < k1;
        hpair_load(kb, vA[i] ? kA : k0, vB[i] ? kA + 1 : k0, hp, kreg[i]);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int kA = k0 + (i * NW + wid) * KPW + 2 * g;
        hpair_load(vb, vA[i] ? kA : k0, vB[i] ? kA + 1 : k0, hp, vreg[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
    ER_TP(2);
    float scA[NS], scB[NS];
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        float sA, sB;
        hpair_scores(kreg[i], qv, hp, sA, sB);
        scA[i] = vA[i] ? sA / a.sqrt_d : -INFINITY;
        scB[i] = vB[i] ? sB / a.sqrt_d : -INFINITY;
        mloc = fmaxf(mloc, fmaxf(scA[i], scB[i]));
    }
    const float m = wave_max(mloc);           // -inf when the wave holds no valid key
    ER_TP(3);
    float o[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[j][e] = 0.f;
    float lloc = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const float pA = vA[i] ? expf(scA[i] - m) : 0.f, pB = vB[i] ? expf(scB[i] - m) : 0.f;
        if (hp.p == 0) lloc += pA + pB;
        hpair_accum(vreg[i], pA, pB, hp, o);
    }
    const float l = wave_sum(lloc);
    ER_TP(4);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[j][e] += row_ror<8>(o[j][e]);
    float om[8], oh[8];
    hpair_fold(o, hp, om, oh);
    if ((lane & 15) < 8) {
        float* dst = ored + (wid * 4 + (lane >> 4)) * A3_LD;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[hp.p * 8 + e] = om[e];
        if (hp.lo) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[(8 + hp.p) * 8 + e] = oh[e];
        }
    }
    if (lane == 0) { wm[wid] = m; wl[wid] = l; }
    __syncthreads();
    ER_TP(5);
    attn3_merge_rows<D, NW>(ored, wm, wl, po, pml);
    ER_TP(7);
}

// Argument list: the length word's address, q, the cache and partial pointers and the two length integers lead as SCALARS (preloaded
// into SGPRs at wave launch, see k_gemv.h gemv_kernel); the struct behind them supplies the rest, its copies of these fields are ignored.
template <typename KT, int D, int STEPS, int NW>
__global__ __launch_bounds__(64 * NW) void attn_decode3_kernel(const int* plen, const float* pq, const void* pk, const void* pv, float* ppart,
                                                               float* ppml, int pfixed, int padd, AttnDecArgs a_) {
    AttnDecArgs a = a_;
    a.len_src = plen; a.q = pq; a.kcache = pk; a.vcache = pv; a.part = ppart; a.part_ml = ppml; a.fixed_len = pfixed; a.len_add = padd;
    constexpr int KPW = 64 / KVec<KT>::LPK;
    static_assert(NW == 16 && D % NW == 0 && D + 1 == A3_LD, "wave merge: the 4 * NW row partials of a column fill the 64 lanes of one wave");
    __shared__ __attribute__((aligned(16))) float ored[NW * 4 * A3_LD];
    __shared__ float wm[NW], wl[NW];
    ER_TP(0);
    const int h = blockIdx.x, c = blockIdx.y, b = blockIdx.z, nch = gridDim.y;
    // The road to the first K load is a chain of dependent scalar round trips: kernel arguments -> the row's length (a device word the
    // sampling head wrote) -> chunk bounds -> addresses.  Left alone hipcc walked attn_len's three cases one scalar load at a time,
    // fetched the cache pointers only behind the early-exit test, and did every address computation (and the q loads) after the
    // length had arrived: 1140 cycles from "length here" to "loads issued" (profiles/r03_attn_timeline.log).  Here: ONE batch of
    // argument loads, the length load, then everything that does not depend on it - pointers, the q registers - while it is in
    // flight; the chunk count is a power of two (16 or 32), so the bounds are shifts.
    {
        const void *p0 = a.q, *p1 = a.kcache, *p2 = a.vcache, *p3 = a.part, *p4 = a.part_ml, *p5 = a.len_src;
        const long long s0 = a.kv_bstride;
        const int i0 = a.H, i1 = a.l_cap, i2 = a.hidden, i3 = a.fixed_len, i4 = a.len_add;
        const float f0 = a.sqrt_d;
        asm volatile("" ::"s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(s0), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(f0));
    }
    const int lmem = a.len_src[b];
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK, NV = D / (EPL * LPK);
    float* po = a.part + (((long long)b * a.H + h) * nch + c) * D;
    float* pml = a.part_ml + (((long long)b * a.H + h) * nch + c) * 2;
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;
    const KT* kb = reinterpret_cast<const KT*>(a.kcache) + head_off;
    const KT* vb = reinterpret_cast<const KT*>(a.vcache) + head_off;
    const float* qp = a.q + (long long)b * a.hidden + h * D;
    float qv[NV][EPL];
    if constexpr (sizeof(KT) == 2) {
        hpair_load_q(qp, HPair(threadIdx.x & 63), qv);          // pair mapping (fp16 cache): NV = 3 loads of EPL = 8 halves
    } else {
        const int p = threadIdx.x & (LPK - 1);
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(qp + (j * LPK + p) * EPL + e);
                qv[j][e] = t.x; qv[j][e + 1] = t.y; qv[j][e + 2] = t.z; qv[j][e + 3] = t.w;
            }
    }
    const int len = a.fixed_len > 0 ? a.fixed_len : lmem + a.len_add;
    int clen = (len + nch - 1) >> __builtin_ctz(nch);            // <= STEPS * NW * KPW (the launcher checks l_cap and that nch is 16 / 32)
    if constexpr (sizeof(KT) == 2) clen = (clen + 1) & ~1;       // fp16 cache: chunks start at even keys (a pair of rows = three whole lines)
    const int k0 = c * clen, k1 = min(len, k0 + clen);
    if (k0 >= k1) {                                        // empty chunk (len < nch): a partial the merge weighs with exp(-inf) = 0
        if (threadIdx.x < D) po[threadIdx.x] = 0.f;
        if (threadIdx.x == 0) { pml[0] = -INFINITY; pml[1] = 0.f; }
        return;
    }
    const int nsteps = (k1 - k0 + NW * KPW - 1) / (NW * KPW);      // workgroup-uniform
    ER_TP(1);
    if constexpr (sizeof(KT) == 2) {
        static_assert(sizeof(KT) != 2 || STEPS == 2, "fp16 cache: two wave-steps of 16 keys");
        if (nsteps >= 2) attn3_body_h<D, 2, NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
        else attn3_body_h<D, 1, NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    } else
    if (STEPS >= 4 && nsteps >= 4) attn3_body<KT, D, (STEPS >= 4 ? 4 : 1), NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    else if (STEPS >= 3 && nsteps == 3) attn3_body<KT, D, (STEPS >= 3 ? 3 : 1), NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    else if (STEPS >= 2 && nsteps == 2) attn3_body<KT, D, (STEPS >= 2 ? 2 : 1), NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    else attn3_body<KT, D, 1, NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
}

constexpr int ATTN3_NW = 16;                   // waves per workgroup of the balanced kernel
constexpr int ATTN3_CAP = 512;                 // keys per workgroup: fp32 4 steps x 16 waves x 8, fp16 2 steps x 16 waves x 16
inline int attn3_num_chunks(int heads) { return heads >= 16 ? 16 : 32; }   // H * NCH = 256 workgroups for the 16-head decoder
inline bool attn3_fits(int l_cap, int heads) { return l_cap <= attn3_num_chunks(heads) * ATTN3_CAP; }

template <int D>
inline hipError_t launch_attn_partial3_d(const AttnDecArgs& a_in, bool kv_half, int nch, int B, hipStream_t st) {
    const AttnDecArgs& a0 = a_in;
    if ((nch != 16 && nch != 32) || a0.l_cap > nch * ATTN3_CAP || !a0.part_ml || (!a0.pos && !a0.len_dev && a0.fixed_len <= 0)) return hipErrorInvalidValue;   // a chunk must fit one workgroup's wave-steps
    const dim3 grid(a0.H, nch, B), blk(64 * ATTN3_NW);
    AttnDecArgs a = a_in;
    // (with a fixed length and no length array the kernel's unconditional length load reads a word of the partials buffer and ignores it)
    a.len_src = a.len_dev ? a.len_dev : (a.pos ? a.pos : reinterpret_cast<const int*>(a.part_ml));
    a.len_add = a.len_dev ? 0 : 1;
    if (!kv_half) hipLaunchKernelGGL((attn_decode3_kernel<float, D, 4, ATTN3_NW>), grid, blk, 0, st, a.len_src, a.q, a.kcache, a.vcache, a.part,
                                     a.part_ml, a.fixed_len, a.len_add, a);
    else hipLaunchKernelGGL((attn_decode3_kernel<_Float16, D, 2, ATTN3_NW>), grid, blk, 0, st, a.len_src, a.q, a.kcache, a.vcache, a.part,
                            a.part_ml, a.fixed_len, a.len_add, a);
    return hipGetLastError();
}

// ---- streaming variant for batches (ER_ATTN_V_BATCHED=3): ONE workgroup per (row, head) walks the whole key range.
//
// With B*H >= 256 (row, head) pairs there is nothing to gain from cutting the keys into chunks: the split kernels above
// launch B*H*ceil(L/128) short-lived workgroups (24.5k at B = 32), each paying its own q load, load-wait-compute ramp,
// barriers and partial write, plus a merge launch.  Here grid = (H, B) (heads fastest: XCD = h % 8), 4 waves; wave w owns
// key groups (4*i + w) of every 4*STEPS*KPW-key tile and keeps its OWN running softmax {m, l-per-lane, o-per-lane} - no
// workgroup barrier inside the loop - with two register tiles in flight (the loads of tile t+1 are issued before tile t is
// reduced; loads are unconditional - past the end every lane re-reads the last key row - because a load under a condition
// makes hipcc wait for everything at the join).  The lane groups of a wave share the wave's running maximum, so their o / l accumulators add
// up without rescaling: one DPP + LDS reduction at the very end, the four waves merged like four partials, and the
// normalised head output written straight to out[b][h*D..] - no partials, no merge kernel.
template <typename KT, int D, int STEPS>
struct AttnTile { f32x4 k[STEPS][D / (KVec<KT>::EPL * KVec<KT>::LPK)], v[STEPS][D / (KVec<KT>::EPL * KVec<KT>::LPK)]; };

template <typename KT, int D, int STEPS>
__device__ __forceinline__ void attn_stream_load(AttnTile<KT, D, STEPS>& t, const KT* kb, const KT* vb, int kbase, int len, int wid,
                                                 int g, int p) {
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK, NV = D / (EPL * LPK), KPW = 64 / LPK;
    if constexpr (sizeof(KT) == 2) {              // fp16 cache: pair mapping (whole 128-byte lines), kbase is a multiple of the tile = even
        const int lane = g * LPK + p;
        const HPair hp(lane);
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = kbase + (i * ER_NWAVES + wid) * KPW + 2 * (lane >> 3);
            hpair_load(kb, max(0, min(kA, len - 1)), max(0, min(kA + 1, len - 1)), hp, t.k[i]);
        }
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = kbase + (i * ER_NWAVES + wid) * KPW + 2 * (lane >> 3);
            hpair_load(vb, max(0, min(kA, len - 1)), max(0, min(kA + 1, len - 1)), hp, t.v[i]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int kk = max(0, min(kbase + (i * ER_NWAVES + wid) * KPW + g, len - 1));    // clamped: never reads the unused tail
        const f32x4* kr = reinterpret_cast<const f32x4*>(kb + (long long)kk * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) t.k[i][j] = __builtin_nontemporal_load(kr + j * LPK + p);
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int kk = max(0, min(kbase + (i * ER_NWAVES + wid) * KPW + g, len - 1));
        const f32x4* vr = reinterpret_cast<const f32x4*>(vb + (long long)kk * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) t.v[i][j] = __builtin_nontemporal_load(vr + j * LPK + p);
    }
}

template <typename KT, int D, int STEPS>
__device__ __forceinline__ void attn_stream_reduce(const AttnTile<KT, D, STEPS>& t, const float (&qv)[D / (KVec<KT>::EPL * KVec<KT>::LPK)][KVec<KT>::EPL],
                                                   int kbase, int len, int wid, int g, int p, float sqrt_d, float& m_run, float& l_lane,
                                                   float (&o)[D / (KVec<KT>::EPL * KVec<KT>::LPK)][KVec<KT>::EPL]) {
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK, NV = D / (EPL * LPK), KPW = 64 / LPK;
    if constexpr (sizeof(KT) == 2) {              // fp16 cache: pair mapping, two keys (A, B) per 8 lanes
        const int lane = g * LPK + p;
        const HPair hp(lane);
        float scA[STEPS], scB[STEPS];
        float mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int kA = kbase + (i * ER_NWAVES + wid) * KPW + 2 * (lane >> 3);
            float sA, sB;
            hpair_scores(t.k[i], qv, hp, sA, sB);
            scA[i] = kA < len ? sA / sqrt_d : -INFINITY;
            scB[i] = kA + 1 < len ? sB / sqrt_d : -INFINITY;
            mloc = fmaxf(mloc, fmaxf(scA[i], scB[i]));
        }
        const float m_new = fmaxf(m_run, wave_max(mloc));
        if (m_new == -INFINITY) return;
        const float alpha = expf(m_run - m_new);
        float pA[STEPS], pB[STEPS];
        float ladd = 0.f;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            pA[i] = expf(scA[i] - m_new);
            pB[i] = expf(scB[i] - m_new);
            ladd += pA[i] + pB[i];
        }
        l_lane = fmaf(l_lane, alpha, hp.p == 0 ? ladd : 0.f);
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[j][e] *= alpha;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) hpair_accum(t.v[i], pA[i], pB[i], hp, o);
        m_run = m_new;
        return;
    }
    float sc[STEPS];
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float kf[EPL];
            kv_unpack<KT>(t.k[i][j], kf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc = fmaf(qv[j][e], kf[e], acc);
        }
        acc = lane_group_sum<LPK>(acc);
        const bool valid = kbase + (i * ER_NWAVES + wid) * KPW + g < len;
        sc[i] = valid ? acc / sqrt_d : -INFINITY;
        mloc = fmaxf(mloc, sc[i]);
    }
    const float m_new = fmaxf(m_run, wave_max(mloc));
    if (m_new == -INFINITY) return;                      // wave-uniform: this wave has not seen a valid key yet
    const float alpha = expf(m_run - m_new);             // exp(-inf) = 0 on the first tile with keys
    float pw[STEPS];
    float ladd = 0.f;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        pw[i] = expf(sc[i] - m_new);                     // 0 for masked keys
        ladd += pw[i];
    }
    l_lane = fmaf(l_lane, alpha, p == 0 ? ladd : 0.f);   // each key counted once (the LPK lanes of a key hold the same score)
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[j][e] *= alpha;
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float vf[EPL];
            kv_unpack<KT>(t.v[i][j], vf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[j][e] = fmaf(pw[i], vf[e], o[j][e]);
        }
    m_run = m_new;
}

template <typename KT, int D, int STEPS>
__global__ __launch_bounds__(ER_WG) void attn_stream_kernel(AttnDecArgs a) {
    static_assert(sizeof(KT) != 2 || D == 96, "fp16 cache: the pair mapping of this kernel is written for 192-byte rows");
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK;
    constexpr int NV = D / (EPL * LPK);
    constexpr int KPW = 64 / LPK;
    constexpr int TILE = ER_NWAVES * STEPS * KPW;        // keys per workgroup iteration
    __shared__ __attribute__((aligned(16))) float ored[ER_NWAVES * 4 * D];
    __shared__ float wm[ER_NWAVES], wl[ER_NWAVES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & (LPK - 1), g = lane / LPK;
    const int h = blockIdx.x, b = blockIdx.y;
    const int len = attn_len(a, b);
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;
    const KT* kb = reinterpret_cast<const KT*>(a.kcache) + head_off;
    const KT* vb = reinterpret_cast<const KT*>(a.vcache) + head_off;
    float qv[NV][EPL];
    const float* qp = a.q + (long long)b * a.hidden + h * D;
    if constexpr (sizeof(KT) == 2) {
        hpair_load_q(qp, HPair(lane), qv);
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(qp + (j * LPK + p) * EPL + e);
                qv[j][e] = t.x; qv[j][e + 1] = t.y; qv[j][e + 2] = t.z; qv[j][e + 3] = t.w;
            }
    }
    float o[NV][EPL];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[j][e] = 0.f;
    float m_run = -INFINITY, l_lane = 0.f;

    const int nt = (len + TILE - 1) / TILE;               // >= 1
    AttnTile<KT, D, STEPS> ta, tb;
    attn_stream_load<KT, D, STEPS>(ta, kb, vb, 0, len, wid, g, p);
    for (int t = 0; t < nt; t += 2) {
        attn_stream_load<KT, D, STEPS>(tb, kb, vb, (t + 1) * TILE, len, wid, g, p);   // past the end: every lane re-reads row len-1 (one L1 line set)
        __builtin_amdgcn_sched_barrier(0);
        attn_stream_reduce<KT, D, STEPS>(ta, qv, t * TILE, len, wid, g, p, a.sqrt_d, m_run, l_lane, o);
        __builtin_amdgcn_sched_barrier(0);
        attn_stream_load<KT, D, STEPS>(ta, kb, vb, (t + 2) * TILE, len, wid, g, p);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) attn_stream_reduce<KT, D, STEPS>(tb, qv, (t + 1) * TILE, len, wid, g, p, a.sqrt_d, m_run, l_lane, o);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- the wave's lane groups share m_run: add their accumulators (rows of 16 lanes by DPP, the rest through LDS)
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[j][e] += row_ror<8>(o[j][e]);
    const float l = wave_sum(l_lane);
    if constexpr (sizeof(KT) == 2) {          // pair mapping: every part sits in two (lane, load) slots 4 lanes apart
        const HPair hp(lane);
        float om[8], oh[8];
        hpair_fold(o, hp, om, oh);
        if ((lane & 15) < 8) {
            float* dst = ored + (wid * 4 + (lane >> 4)) * D;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[hp.p * 8 + e] = om[e];
            if (hp.lo) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[(8 + hp.p) * 8 + e] = oh[e];
            }
        }
    } else if ((lane & 15) < LPK) {
        float* dst = ored + (wid * 4 + (lane >> 4)) * D;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) dst[(j * LPK + p) * EPL + e] = o[j][e];
    }
    if (lane == 0) { wm[wid] = m_run; wl[wid] = l; }
    __syncthreads();
    if (tid < D) {
        const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));   // finite: len >= 1
        float ov = 0.f, lv = 0.f;
#pragma unroll
        for (int k = 0; k < ER_NWAVES; ++k) {
            const float w = (wm[k] == -INFINITY) ? 0.f : expf(wm[k] - M);
            const float* src = ored + k * 4 * D + tid;
            ov = fmaf((src[0] + src[D]) + (src[2 * D] + src[3 * D]), w, ov);
            lv = fmaf(wl[k], w, lv);
        }
        const float r = ov / lv;
        a.out[(long long)b * a.hidden + h * D + tid] = r;
        if (a.out_xt) {        // element k = h D + tid of row b: halves (k & 3) [hi] and 4 + (k & 3) [lo] of entry (k >> 2, b)
            const int k = h * D + tid;
            _Float16* e = reinterpret_cast<_Float16*>(a.out_xt) + xt_entry(a.hidden, b, k >> 2) * 8 + (k & 3);
            const _Float16 hi = (_Float16)r;
            e[0] = hi;
            e[4] = (_Float16)(r - (float)hi);
        }
    }
}

template <int D>
inline hipError_t launch_attn_stream_d(const AttnDecArgs& a, bool kv_half, int B, hipStream_t st) {
    const dim3 grid(a.H, B), blk(ER_WG);
    if (!kv_half) hipLaunchKernelGGL((attn_stream_kernel<float, D, 2>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((attn_stream_kernel<_Float16, D, 2>), grid, blk, 0, st, a);
    return hipGetLastError();
}

constexpr int ATTN_STEPS_DEFAULT = 4;          // fp32 KV: 128 keys per workgroup
inline int attn_chunk(int steps, bool /*kv_half*/) { return 32 * steps; }   // fp16 KV: 64 keys/step x steps/2
inline int attn_num_chunks(int l_cap, int chunk) { return (l_cap + chunk - 1) / chunk; }

// `steps` is the fp32 step count (chunk = 32*steps keys); fp16 KV uses half as many steps for the same chunk.
template <int D>
inline hipError_t launch_attn_partial_d(const AttnDecArgs& a, int steps, bool kv_half, int B, hipStream_t st, int version = 2) {
    const dim3 grid(a.H, a.S, B), blk(ER_WG);      // heads fastest: XCD = head mod 8 (see attn_decode_kernel)
    if (version == 2) {
        if (!kv_half) {
            if (steps == 2) hipLaunchKernelGGL((attn_decode2_kernel<float, D, 2>), grid, blk, 0, st, a);
            else if (steps == 8) hipLaunchKernelGGL((attn_decode2_kernel<float, D, 8>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((attn_decode2_kernel<float, D, 4>), grid, blk, 0, st, a);
        } else {
            if (steps == 2) hipLaunchKernelGGL((attn_decode2_kernel<_Float16, D, 1>), grid, blk, 0, st, a);
            else if (steps == 8) hipLaunchKernelGGL((attn_decode2_kernel<_Float16, D, 4>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((attn_decode2_kernel<_Float16, D, 2>), grid, blk, 0, st, a);
        }
        return hipGetLastError();
    }
    if (!kv_half) {
        if (steps == 2) hipLaunchKernelGGL((attn_decode_kernel<float, D, 2>), grid, blk, 0, st, a);
        else if (steps == 8) hipLaunchKernelGGL((attn_decode_kernel<float, D, 8>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((attn_decode_kernel<float, D, 4>), grid, blk, 0, st, a);
    } else {
        if (steps == 2) hipLaunchKernelGGL((attn_decode_kernel<_Float16, D, 1>), grid, blk, 0, st, a);
        else if (steps == 8) hipLaunchKernelGGL((attn_decode_kernel<_Float16, D, 4>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((attn_decode_kernel<_Float16, D, 2>), grid, blk, 0, st, a);
    }
    return hipGetLastError();
}
template <int D>
inline hipError_t launch_attn_combine_d(const AttnDecArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL((attn_combine2_kernel<D>), dim3(a.H, B), dim3(ER_WG), 0, st, a);
    return hipGetLastError();
}

}  // namespace er
