// Where do the ~11 us of the single-row decode attention launch go?  The library's own attn_decode3_kernel (k_attn_decode.h)
// compiled with timeline hooks: thread 0 of every workgroup stamps s_memtime (shader clock) at
//   0 entry | 1 pos loaded, chunk known | 2 all K/V loads issued | 3 scores + wave max done (K landed) | 4 P.V done (V landed)
//   5 workgroup barrier passed (round 2: 6 = its second barrier) | 7 partial stored
// and s_memrealtime (100 MHz, chip-wide) at entry and exit, so the launch boundary and the dispatch skew can be read too.
// In situ stand-in: a graph of 24 x (plain 28 MB weight-streaming kernel, attention over its own layer's 2 x 37.7 MB cache).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../edgerunner_amd/csrc -o attn_timeline_probe attn_timeline_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

extern __shared__ unsigned long long er_tp_dyn[];     // [0..7] s_memtime stamps, [8..9] s_memrealtime at entry / exit
#define ER_TP(i)                                                                         \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        if (threadIdx.x == 0) {                                                          \
            er_tp_dyn[i] = __builtin_amdgcn_s_memtime();                                 \
            if ((i) == 0) er_tp_dyn[8] = __builtin_amdgcn_s_memrealtime();               \
            if ((i) == 7) er_tp_dyn[9] = __builtin_amdgcn_s_memrealtime();               \
        }                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#include "k_attn_decode.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace er;

__device__ unsigned long long* g_tp_out;

// the library kernel + a tail that copies the stamps out: launch index travels in a.S (unused by version 3)
// NEWENTRY: the library kernel's round-3 entry restated (one batch of argument loads, the length load, length-independent work while
// it is in flight, shift bounds); false: the round-2 entry (attn_len's three cases, pointers behind the length)
template <typename KT, int STEPS, bool NEWENTRY>
__global__ __launch_bounds__(64 * ATTN3_NW) void attn3_timed(AttnDecArgs a) {
    // same body as attn_decode3_kernel (k_attn_decode.h), which cannot be called as a function: restated dispatch (round-2 form of the
    // entry: the library kernel now batches its argument loads and hoists the length-independent work), shared arrays here
    constexpr int D = 96, NW = ATTN3_NW;
    constexpr int KPW = 64 / KVec<KT>::LPK;
    __shared__ __attribute__((aligned(16))) float ored[NW * 4 * A3_LD];
    __shared__ float wm[NW], wl[NW];
    ER_TP(0);
    const int h = blockIdx.x, c = blockIdx.y, b = blockIdx.z, nch = gridDim.y;
    int lmem = 0, len_old = 0;
    if (!NEWENTRY) len_old = attn_len(a, b);
    if (NEWENTRY) {
        const void *p0 = a.q, *p1 = a.kcache, *p2 = a.vcache, *p3 = a.part, *p4 = a.part_ml, *p5 = a.len_src;
        const long long s0 = a.kv_bstride;
        const int i0 = a.H, i1 = a.l_cap, i2 = a.hidden, i3 = a.fixed_len, i4 = a.len_add;
        const float f0 = a.sqrt_d;
        asm volatile("" ::"s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(s0), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(f0));
        lmem = a.len_src[b];
    }
    float* po = a.part + (((long long)b * a.H + h) * nch + c) * D;
    float* pml = a.part_ml + (((long long)b * a.H + h) * nch + c) * 2;
    const long long head_off = (long long)b * a.kv_bstride + (long long)h * a.l_cap * D;
    const KT* kb = reinterpret_cast<const KT*>(a.kcache) + head_off;
    const KT* vb = reinterpret_cast<const KT*>(a.vcache) + head_off;
    const float* qp = a.q + (long long)b * a.hidden + h * D;
    constexpr int EPL = KVec<KT>::EPL, LPK = KVec<KT>::LPK, NV = D / (EPL * LPK);
    float qv[NV][EPL];
    if constexpr (sizeof(KT) == 2) {
        hpair_load_q(qp, HPair(threadIdx.x & 63), qv);          // round 5: the fp16 cache's pair mapping
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
    const int len = NEWENTRY ? (a.fixed_len > 0 ? a.fixed_len : lmem + a.len_add) : len_old;
    int clen = NEWENTRY ? ((len + nch - 1) >> __builtin_ctz(nch)) : (len + nch - 1) / nch;
    if constexpr (sizeof(KT) == 2) clen = (clen + 1) & ~1;
    const int k0 = c * clen, k1 = min(len, k0 + clen);
    const int nsteps = (k1 - k0 + NW * KPW - 1) / (NW * KPW);
    ER_TP(1);
    if constexpr (sizeof(KT) == 2) {
        if (nsteps >= 2) attn3_body_h<D, 2, NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
        else attn3_body_h<D, 1, NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    } else
    if (STEPS >= 4 && nsteps >= 4) attn3_body<KT, D, (STEPS >= 4 ? 4 : 1), NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    else if (STEPS >= 3 && nsteps == 3) attn3_body<KT, D, (STEPS >= 3 ? 3 : 1), NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    else if (STEPS >= 2 && nsteps == 2) attn3_body<KT, D, (STEPS >= 2 ? 2 : 1), NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    else attn3_body<KT, D, 1, NW>(a, kb, vb, qv, k0, k1, ored, wm, wl, po, pml);
    if (threadIdx.x == 0) {
        unsigned long long* o = g_tp_out + ((long long)a.S * 256 + (blockIdx.x + 16 * blockIdx.y)) * 10;
        for (int i = 0; i < 10; ++i) o[i] = er_tp_dyn[i];
    }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void filler_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N,
                                                     unsigned long long* end_rt) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    const f32x4v* wr = reinterpret_cast<const f32x4v*>(W + (long long)min(row, N - 1) * 1536);
    f32x4v w[6], x[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) w[j] = __builtin_nontemporal_load(wr + j * 64 + lane);
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = reinterpret_cast<const f32x4v*>(xin)[j * 64 + lane];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) { s = fmaf(w[j].x, x[j].x, s); s = fmaf(w[j].y, x[j].y, s); s = fmaf(w[j].z, x[j].z, s); s = fmaf(w[j].w, x[j].w, s); }
    s = wave_sum(s);
    if (lane == 0 && row < N) yout[row] = s;
    if (threadIdx.x == 0) end_rt[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char** argv) {
    const int len = argc > 1 ? atoi(argv[1]) : 4050;
    const bool half = argc > 2 && atoi(argv[2]) == 16;
    const int use_fixed = argc > 3 ? atoi(argv[3]) : 0;      // 1: the length travels as a kernel argument (no dependent pos load)
    const int new_entry = argc > 4 ? atoi(argv[4]) : 1;      // 0: the round-2 form of the kernel entry
    const int NL = 24, H = 16, D = 96, Lcap = 6144, NQ = 4608;
    const size_t esz = half ? 2 : 4;
    const size_t layer_elems = (size_t)H * Lcap * D;
    void *kc, *vc;
    CHECK(hipMalloc(&kc, layer_elems * NL * esz));
    CHECK(hipMalloc(&vc, layer_elems * NL * esz));
    {   // small finite values (fp32 or fp16 bit patterns alike)
        std::vector<unsigned short> hbuf(layer_elems * esz / 2);
        unsigned s = 99u;
        for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; v = half ? (unsigned short)(0x2000 + ((s >> 9) & 0x0fff)) : (unsigned short)(s >> 16); }
        if (!half) for (size_t i = 1; i < hbuf.size(); i += 2) hbuf[i] = (unsigned short)(0x3c00 + (hbuf[i] & 0x01ff));    // high half of a float ~ 0.0078..0.03
        for (int l = 0; l < NL; ++l) {
            CHECK(hipMemcpy((char*)kc + l * layer_elems * esz, hbuf.data(), layer_elems * esz, hipMemcpyHostToDevice));
            CHECK(hipMemcpy((char*)vc + l * layer_elems * esz, hbuf.data(), layer_elems * esz, hipMemcpyHostToDevice));
        }
    }
    float *q, *part, *part_ml, *Wf, *xa, *xb;
    int* pos;
    CHECK(hipMalloc(&q, 1536 * 4));
    CHECK(hipMemset(q, 0, 1536 * 4));
    CHECK(hipMalloc(&part, 16 * 16 * 96 * 4));
    CHECK(hipMalloc(&part_ml, 16 * 16 * 2 * 4));
    CHECK(hipMalloc(&pos, 4));
    const int hp = len - 1;
    CHECK(hipMemcpy(pos, &hp, 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&Wf, (size_t)NQ * 1536 * 4 * NL));
    CHECK(hipMemset(Wf, 0, (size_t)NQ * 1536 * 4 * NL));
    CHECK(hipMalloc(&xa, 8192 * 4));
    CHECK(hipMalloc(&xb, 8192 * 4));
    CHECK(hipMemset(xa, 0, 8192 * 4));
    unsigned long long *tp, *frt;
    CHECK(hipMalloc(&tp, (size_t)NL * 256 * 10 * 8));
    CHECK(hipMalloc(&frt, (size_t)NL * (NQ / 4) * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tp_out), &tp, sizeof(tp)));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipGraph_t graph;
    hipGraphExec_t gexec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int l = 0; l < NL; ++l) {
        hipLaunchKernelGGL(filler_kernel, dim3(NQ / 4), dim3(256), 0, st, Wf + (size_t)l * NQ * 1536, xa, xb, NQ, frt + (size_t)l * (NQ / 4));
        AttnDecArgs a{};
        a.q = q; a.kcache = (char*)kc + l * layer_elems * esz; a.vcache = (char*)vc + l * layer_elems * esz;
        a.pos = pos; a.fixed_len = use_fixed ? len : 0; a.len_dev = nullptr; a.part = part; a.part_ml = part_ml; a.out = nullptr;
        a.H = H; a.l_cap = Lcap; a.S = l; a.hidden = 1536; a.chunk = 0; a.kv_bstride = (long long)layer_elems; a.sqrt_d = sqrtf(96.f);
        a.len_src = pos; a.len_add = 1;
        if (!half && new_entry) hipLaunchKernelGGL((attn3_timed<float, 4, true>), dim3(16, 16, 1), dim3(1024), 128, st, a);
        else if (!half) hipLaunchKernelGGL((attn3_timed<float, 4, false>), dim3(16, 16, 1), dim3(1024), 128, st, a);
        else if (new_entry) hipLaunchKernelGGL((attn3_timed<_Float16, 2, true>), dim3(16, 16, 1), dim3(1024), 128, st, a);
        else hipLaunchKernelGGL((attn3_timed<_Float16, 2, false>), dim3(16, 16, 1), dim3(1024), 128, st, a);
    }
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(gexec, st));
    CHECK(hipEventRecord(e0, st));
    CHECK(hipGraphLaunch(gexec, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)NL * 256 * 10), hf((size_t)NL * (NQ / 4));
    CHECK(hipMemcpy(h.data(), tp, h.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hf.data(), frt, hf.size() * 8, hipMemcpyDeviceToHost));
    printf("%s entry | len %d, %s cache, length from %s: graph of 24 x (filler 28 MB, attention) = %.1f us -> %.2f us per pair\n", new_entry ? "round-3" : "round-2", len, half ? "fp16" : "fp32",
           use_fixed ? "kernel argument" : "device pos", ms * 1e3, ms * 1e3 / NL);
    // per launch: boundary (filler's last exit -> attention's first entry), entry skew, total (first entry -> last exit), all in us (realtime = 100 MHz)
    double acc[8] = {0}, mx[8] = {0}, bnd = 0, skew = 0, tot = 0, exitskew = 0;
    int n = 0;
    for (int l = 4; l < NL; ++l) {
        unsigned long long fend = 0, smin = ~0ull, smax = 0, emax = 0, emin = ~0ull;
        for (int i = 0; i < NQ / 4; ++i) fend = std::max(fend, hf[(size_t)l * (NQ / 4) + i]);
        double d[8] = {0}, dm[8] = {0};
        for (int w = 0; w < 256; ++w) {
            const unsigned long long* t = &h[((size_t)l * 256 + w) * 10];
            smin = std::min(smin, t[8]); smax = std::max(smax, t[8]); emax = std::max(emax, t[9]); emin = std::min(emin, t[9]);
            for (int i = 1; i < 8; ++i) { const double c = (double)(t[i] - t[0]); d[i] += c / 256.0; dm[i] = std::max(dm[i], c); }
        }
        for (int i = 1; i < 8; ++i) { acc[i] += d[i]; mx[i] += dm[i]; }
        bnd += (double)(smin - fend) / 100.0; skew += (double)(smax - smin) / 100.0; tot += (double)(emax - smin) / 100.0; exitskew += (double)(emax - emin) / 100.0;
        ++n;
    }
    printf("boundary filler-exit -> first attention entry %.2f us | entry skew %.2f us | first entry -> last exit %.2f us | exit skew %.2f us\n",
           bnd / n, skew / n, tot / n, exitskew / n);
    // where does the exit skew come from?  exit time (us after the launch's first entry) by XCD (= head % 8: workgroup (h, c) is number
    // h + 16 c of the grid) and by chunk index, mean over the launches
    {
        double xs[8] = {0}, xm[8] = {0}, cs[16] = {0}, es[8] = {0};
        int nl = 0;
        for (int l = 4; l < NL; ++l, ++nl) {
            unsigned long long smin = ~0ull;
            for (int w = 0; w < 256; ++w) smin = std::min(smin, h[((size_t)l * 256 + w) * 10 + 8]);
            double lx[8] = {0};
            for (int w = 0; w < 256; ++w) {
                const unsigned long long* t = &h[((size_t)l * 256 + w) * 10];
                const int hh = w & 15, cc = w >> 4;
                const double ex = (double)(t[9] - smin) / 100.0, en = (double)(t[8] - smin) / 100.0;
                xs[hh & 7] += ex / 32.0; lx[hh & 7] = std::max(lx[hh & 7], ex); cs[cc] += ex / 16.0; es[hh & 7] += en / 32.0;
            }
            for (int x = 0; x < 8; ++x) xm[x] += lx[x];
        }
        printf("exit time by XCD  (mean | last, us after first entry; entry mean):");
        for (int x = 0; x < 8; ++x) printf("  x%d %.2f|%.2f (%.2f)", x, xs[x] / nl, xm[x] / nl, es[x] / nl);
        printf("\nexit time by chunk (mean, us):");
        for (int c = 0; c < 16; ++c) printf(" %.2f", cs[c] / nl);
        printf("\n");
    }
    const char* names[8] = {"entry", "pos loaded", "loads issued", "K landed, scores+max", "V landed, P.V", "barrier passed", "(unused)", "stored"};
    printf("stamps relative to the workgroup's entry, shader-clock cycles (mean over workgroups | max):\n");
    for (int i = 1; i < 8; ++i) printf("  %d %-22s %8.0f | %8.0f\n", i, names[i], acc[i] / n, mx[i] / n);
    return 0;
}
