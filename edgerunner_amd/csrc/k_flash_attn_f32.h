// Flash-style attention in EXACT fp32 on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain at
// the fp32 vector rate) for the once-per-sample work of the hot path:
//   * the causal self-attention of the 2050-token prefill (core/transformer/modeling_opt.py:229 ->
//     core/transformer/attention.py:27-62 with N == M), head_dim 96;
//   * the point encoder's cross-attention of 2048 learned queries over the N points
//     (core/transformer/point.py:108-126 -> attention.py, non-causal), head_dim 64.
// softmax(Q K^T / sqrt(D) [+ causal mask]) V without materialising the [H, N, M] score matrix that the round-1 path wrote
// and re-read per sample (271 MB at S = 2050; 537 MB for the encoder at N = 4096): the reference's own GPU path
// (flash-attn, attention.py:44-46) never materialises it either.  All (query tile, head, sample) triples of a batch run
// in ONE launch.
//
// Same transposed formulation as the fp16 kernel (k_flash_attn.h): S^T = K Q^T and O^T = V^T P^T, so that everything a
// query needs stays in one lane column (q = lane & 31): the row max / sum need a single xor-32 exchange, the rescale
// factor is lane-local, and the S^T accumulator registers ARE the B operand of the second product (fp32 in, no
// repacking): MFMA step r of a 32-key block multiplies keys k0(r) (lanes 0-31) and k0(r) + 4 (lanes 32-63), exactly the
// two keys whose probabilities sit in accumulator register r of the two lane halves.  The matrix core sums k in any
// order, A and B only have to agree on the (lane half, step) -> k assignment; for S^T it is d = half * D/2 + step, which
// makes the K fragment a contiguous 16-byte LDS read per four steps.
#pragma once
#include "er_common.h"

namespace er {

struct Flash32Args {
    const float* Q; const float* K; const float* V; float* O;
    int N, M;                               // queries, keys per (sample, head)
    int ldq, ldk, ldv, ldo;                 // row strides (floats)
    long long qs_b, ks_b, vs_b, os_b;       // sample strides
    long long qs_h, ks_h, vs_h, os_h;       // head strides
    float sqrt_d;                           // scores are DIVIDED by sqrt(D), as the reference does (attention.py:52)
    int causal_off;                         // CAUSAL: key j is visible to query i iff j <= i + causal_off (= M - N)
    // key-range split (KSP instantiations, launch_flash_attn_f32): unnormalised partial outputs [2][B][H][N][D] and their
    // (running max, sum) pairs [2][B][H][N][2]; null = one workgroup walks the whole key range of its query tile
    float* part_o;
    float* part_ml;
};

typedef float fa32_acc __attribute__((ext_vector_type(16)));
constexpr int FA32_KT = 64, FA32_QW = 32;   // keys per tile, queries per wave

// grid (ceil(N / (32 NWV)), H, B), 64 NWV threads: wave w owns queries [32 NWV * tile + 32 * w, +32).  Causal launches walk
// the query tiles from the last (longest key range) to the first so the long workgroups start first.  NWV = 4 shares a
// staged key tile between 128 queries; NWV = 2 re-stages it twice as often but doubles the number of workgroups, which is
// what a single 2050-token prefill needs (17 x 16 four-wave workgroups = one per CU with nothing to overlap).
//
// KSP (causal prefill of ONE sample; staged in round 4, off by default - see launch_flash_attn_f32): a single 2050-token prefix is 33 x 16
// two-wave workgroups on 512 resident slots, and the launch lasts as long as its LONGEST workgroup - 33 key tiles for the last query
// tile against 17.5 on average, i.e. half of the SIMD time is idle waiting for the diagonal's end.  With KSP the grid holds two
// workgroups per query tile, each walking one half of its key tiles (at most 17) and leaving an unnormalised partial (O, m, l);
// flash32_merge_kernel combines the two.
template <int D, bool CAUSAL, int NWV, bool KSP = false>
__global__ __launch_bounds__(64 * NWV) void flash_attn_f32_kernel(Flash32Args a) {
    constexpr int THREADS = 64 * NWV;
    constexpr int LD = D + 4;               // LDS row stride (floats), LD/4 odd -> the 16-byte K reads of a 16-lane group hit 16 distinct 4-bank slots
    constexpr int HD = D / 2, NDB = D / 32, F4 = D / 4;
    static_assert(D % 32 == 0 && (LD % 8) == 4, "head_dim 64 or 96: LD/4 must be odd");
    __shared__ __attribute__((aligned(16))) float Ks[FA32_KT * LD];   // [key][d]
    __shared__ __attribute__((aligned(16))) float Vs[FA32_KT * LD];   // [key][d]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int bx = KSP ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, nqt = KSP ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int ks = KSP ? (int)(blockIdx.x & 1) : 0;
    const int qt = CAUSAL ? nqt - 1 - bx : bx;
    const int q0 = qt * (NWV * FA32_QW) + wid * FA32_QW;
    const float* Q = a.Q + b * a.qs_b + h * a.qs_h;
    const float* K = a.K + b * a.ks_b + h * a.ks_h;
    const float* V = a.V + b * a.vs_b + h * a.vs_h;
    float* O = a.O + b * a.os_b + h * a.os_h;

    // Q fragment (B operand of S^T): query q0 + li, dims half * D/2 + s
    float qreg[HD];
    const int qi = q0 + li;
    {
        const float* qr = Q + (long long)min(qi, a.N - 1) * a.ldq + half * HD;
#pragma unroll
        for (int u = 0; u < HD / 4; ++u) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qr + 4 * u);
            qreg[4 * u] = t.x; qreg[4 * u + 1] = t.y; qreg[4 * u + 2] = t.z; qreg[4 * u + 3] = t.w;
        }
    }
    fa32_acc ot[NDB];                        // O^T: rows d = db*32 + (r&3) + 8*(r>>2) + 4*half, column q = li
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;    // l_run covers this lane half's keys only
    const float inv_sqrt_d = 1.0f / a.sqrt_d;  // one true division per thread

    // keys any query of this WORKGROUP can see
    const int wg_last_q = min(a.N - 1, qt * (NWV * FA32_QW) + NWV * FA32_QW - 1);
    const int kmax = CAUSAL ? min(a.M, wg_last_q + a.causal_off + 1) : a.M;
    const int ntiles = (kmax + FA32_KT - 1) / FA32_KT;
    const int wave_last_key = CAUSAL ? (q0 + FA32_QW - 1 + a.causal_off) : (a.M - 1);   // beyond it this wave has nothing to do

    // global -> register staging of one key tile (issued a whole tile ahead: the loads of tile t+1 are in flight while
    // tile t is multiplied; one wave per SIMD has no other wave to hide that latency behind)
    constexpr int NST = (FA32_KT * F4) / THREADS;
    static_assert((FA32_KT * F4) % THREADS == 0, "staging loop shape");
    f32x4 kst[NST], vst[NST];
    auto load_tile = [&](int t) {
        const int kb0 = t * FA32_KT;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + THREADS * u, key = idx / F4, c4 = idx - key * F4;
            const int gk = min(kb0 + key, a.M - 1);
            kst[u] = *reinterpret_cast<const f32x4*>(K + (long long)gk * a.ldk + 4 * c4);
            vst[u] = *reinterpret_cast<const f32x4*>(V + (long long)gk * a.ldv + 4 * c4);
        }
    };
    int t_begin = 0, t_end = ntiles;
    if constexpr (KSP) {
        const int hlf = (ntiles + 1) >> 1;
        t_begin = ks * hlf;
        t_end = min(ntiles, t_begin + hlf);
    }
    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        const int kbase = t * FA32_KT;
        __syncthreads();                     // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + THREADS * u, key = idx / F4, c4 = idx - key * F4;
            *reinterpret_cast<f32x4*>(&Ks[key * LD + 4 * c4]) = kst[u];
            *reinterpret_cast<f32x4*>(&Vs[key * LD + 4 * c4]) = vst[u];
        }
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
        if (kbase > wave_last_key) continue;   // wave-uniform: the barriers above are still hit by every wave

        // S^T = K Q^T: two 32-key blocks; lane (li, half) holds keys kbase + kb*32 + (r&3) + 8*(r>>2) + 4*half
        fa32_acc st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int u = 0; u < HD / 4; ++u) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(&Ks[(kb * 32 + li) * LD + half * HD + 4 * u]);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qreg[4 * u], st[kb], 0, 0, 0);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qreg[4 * u + 1], st[kb], 0, 0, 0);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qreg[4 * u + 2], st[kb], 0, 0, 0);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qreg[4 * u + 3], st[kb], 0, 0, 0);
            }
        }
        // online softmax for query li over this lane half's 32 keys of the tile
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                bool ok = key < a.M;
                if (CAUSAL) ok = ok && key <= qi + a.causal_off;
                // x / sqrt(D) as the reference computes it (attention.py:52), in three instructions instead of hipcc's ten-instruction
                // division: q0 = x * y, r = fma(-q0, c, x), q = fma(r, y, q0) with y = 1 / c is the correctly rounded quotient for every
                // |x| in [1e-30, 1e30] - checked over all 2^32 inputs for c = sqrtf(96) and 8 (scripts/probes/div_const_probe.hip,
                // profiles/r04_div_const_probe.log); 32 divisions per lane and key tile were a third of this loop's VALU work
                const float q0 = st[kb][r] * inv_sqrt_d;
                const float s = ok ? fmaf(fmaf(-q0, a.sqrt_d, st[kb][r]), inv_sqrt_d, q0) : -INFINITY;
                st[kb][r] = s;
                mloc = fmaxf(mloc, s);
            }
        mloc = xor_max<32>(mloc);
        const float m_new = fmaxf(m_run, mloc);
        // m_new stays -inf only for a query that has not seen any key yet (padding rows q >= N of a causal tile)
        const float alpha = (m_new == -INFINITY) ? 1.0f : expf(m_run - m_new);   // m_run = -inf -> exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (st[kb][r] == -INFINITY) ? 0.f : expf(st[kb][r] - m_new);
                st[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;

        // O^T += V^T P^T: step r of block kb multiplies keys kb*32 + k0(r) (+4 in the upper lane half), whose
        // probabilities are accumulator register r of the two halves -> B = st[kb][r] as it stands
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vr = &Vs[(kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LD + li];
#pragma unroll
                for (int db = 0; db < NDB; ++db)
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[db * 32], st[kb][r], ot[db], 0, 0, 0);
            }
    }
    const float l_tot = xor_sum<32>(l_run);
    if constexpr (KSP) {
        // partial of this key range: O^T as accumulated (relative to m_run), m_run (the same in both lane halves) and the sum
        if (qi < a.N) {
            const long long row = (((long long)ks * gridDim.z + b) * gridDim.y + h) * a.N + qi;
            float* prow = a.part_o + row * D;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) prow[db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = ot[db][r];
            if (half == 0) {
                a.part_ml[2 * row] = m_run;
                a.part_ml[2 * row + 1] = l_tot;
            }
        }
        return;
    }
    if (qi < a.N) {
        float* orow = O + (long long)qi * a.ldo;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) orow[db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = ot[db][r] / l_tot;
    }
}

// O[b][q][h][:] = (O0 w0 + O1 w1) / (l0 w0 + l1 w1), w_i = exp(m_i - max(m0, m1)) (0 for an empty key range: m_i = -inf, l_i = 0);
// every causal query sees at least its own key, so the denominator is never 0.  One thread per four output floats.
__global__ __launch_bounds__(ER_WG) void flash32_merge_kernel(Flash32Args a, int D, int H, int B) {
    const int d4n = D / 4;
    const long long total = (long long)B * H * a.N * d4n, half_rows = (long long)B * H * a.N;
    for (long long i = (long long)blockIdx.x * ER_WG + threadIdx.x; i < total; i += (long long)gridDim.x * ER_WG) {
        const long long row = i / d4n;                 // (b * H + h) * N + q
        const int c4 = (int)(i - row * d4n);
        const int q = (int)(row % a.N);
        const long long bh = row / a.N;
        const int h = (int)(bh % H), b = (int)(bh / H);
        const float m0 = a.part_ml[2 * row], l0 = a.part_ml[2 * row + 1];
        const float m1 = a.part_ml[2 * (half_rows + row)], l1 = a.part_ml[2 * (half_rows + row) + 1];
        const float m = fmaxf(m0, m1);
        const float w0 = (m0 == -INFINITY) ? 0.f : expf(m0 - m), w1 = (m1 == -INFINITY) ? 0.f : expf(m1 - m);
        float den = l0 * w0 + l1 * w1;
        if (den == 0.f) den = 1.f;                     // a query with no visible key in either half (not a causal-prefill case): o = 0, not NaN
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(a.part_o + row * D + 4 * c4);
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(a.part_o + (half_rows + row) * D + 4 * c4);
        f32x4 o;
        o.x = (o0.x * w0 + o1.x * w1) / den; o.y = (o0.y * w0 + o1.y * w1) / den;
        o.z = (o0.z * w0 + o1.z * w1) / den; o.w = (o0.w * w0 + o1.w * w1) / den;
        *reinterpret_cast<f32x4*>(a.O + b * a.os_b + h * a.os_h + (long long)q * a.ldo + 4 * c4) = o;
    }
}
// floats of the two partial buffers a key-split launch needs
inline size_t flash32_part_o_floats(int B, int H, int N, int D) { return (size_t)2 * B * H * N * D; }
inline size_t flash32_part_ml_floats(int B, int H, int N) { return (size_t)4 * B * H * N; }

// waves per workgroup of a launch: two while four-wave workgroups would leave fewer than three per CU (ER_FLASH32_NWV = 2 / 4 forces one)
inline int flash32_waves(int N, int H, int B) {
    static const int forced = [] { const char* v = getenv("ER_FLASH32_NWV"); return v ? atoi(v) : 0; }();
    const long long wg4 = (long long)((N + 4 * FA32_QW - 1) / (4 * FA32_QW)) * H * B;
    return (forced == 2 || forced == 4) ? forced : (wg4 < 768 ? 2 : 4);
}
// Key-range split of the causal D = 96 launch (KSP): for ONE sample whose launch runs two-wave workgroups, i.e. the single-prefix
// prefill (encode + prefill 38.3 -> 37.9 ms, profiles/r05_ksplit.log); ER_FLASH32_KSPLIT=0 turns it off.  B == 1 explicitly (round 6,
// ADVICE r5): the merge rounds differently from the unsplit kernel by a few ulp, and up to round 5 a pair of prompts (B = 2 also runs
// two-wave workgroups) took the split while the same prompts in a batch of three did not - now every batch of two or more gives a
// prompt the unsplit kernel's bits whatever its size (tests/test_gpu_kernels.py::test_flash_attn_f32_batch_invariance); a prompt
// prefilled ALONE differs from them by the merge's rounding.  ONE rule for the callers that allocate the partial buffers and for the
// launcher that uses them.
inline bool flash32_ksplit(int N, int H, int B, int D, bool causal) {
    const char* v = getenv("ER_FLASH32_KSPLIT");           // read per call (the tests compare both forms in one process)
    return !(v && atoi(v) == 0) && causal && D == 96 && B == 1 && flash32_waves(N, H, B) == 2;
}

inline hipError_t launch_flash_attn_f32(const Flash32Args& a, int D, bool causal, int H, int B, hipStream_t st) {
    const int nwv = flash32_waves(a.N, H, B);
    dim3 grid((a.N + nwv * FA32_QW - 1) / (nwv * FA32_QW), H, B), blk(64 * nwv);
    // key-range split: when the caller supplies the partial buffers (it asks flash32_ksplit() too) and the output is float4-addressable
    if (a.part_o && a.part_ml && flash32_ksplit(a.N, H, B, D, causal) && (a.ldo & 3) == 0 && (a.os_b & 3) == 0 && (a.os_h & 3) == 0) {
        grid.x *= 2;
        hipLaunchKernelGGL((flash_attn_f32_kernel<96, true, 2, true>), grid, blk, 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const long long total = (long long)B * H * a.N * (D / 4), blocks = (total + ER_WG - 1) / ER_WG;
        hipLaunchKernelGGL(flash32_merge_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(ER_WG), 0, st, a, D, H, B);
        return hipGetLastError();
    }
    if (nwv == 4) {
        if (D == 96 && causal) hipLaunchKernelGGL((flash_attn_f32_kernel<96, true, 4>), grid, blk, 0, st, a);
        else if (D == 96) hipLaunchKernelGGL((flash_attn_f32_kernel<96, false, 4>), grid, blk, 0, st, a);
        else if (D == 64 && causal) hipLaunchKernelGGL((flash_attn_f32_kernel<64, true, 4>), grid, blk, 0, st, a);
        else if (D == 64) hipLaunchKernelGGL((flash_attn_f32_kernel<64, false, 4>), grid, blk, 0, st, a);
        else return hipErrorInvalidValue;
    } else {
        if (D == 96 && causal) hipLaunchKernelGGL((flash_attn_f32_kernel<96, true, 2>), grid, blk, 0, st, a);
        else if (D == 96) hipLaunchKernelGGL((flash_attn_f32_kernel<96, false, 2>), grid, blk, 0, st, a);
        else if (D == 64 && causal) hipLaunchKernelGGL((flash_attn_f32_kernel<64, true, 2>), grid, blk, 0, st, a);
        else if (D == 64) hipLaunchKernelGGL((flash_attn_f32_kernel<64, false, 2>), grid, blk, 0, st, a);
        else return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace er
