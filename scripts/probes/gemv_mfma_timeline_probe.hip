// Where do the 10-13 us of a batched (B = 16 / 32) fast-mode projection go?  The library's own gemv_mfma_kernel (k_gemv_mfma.h, tiled
// fp16 weights, tiled hi | lo activations) compiled with timeline hooks: thread 0 of every workgroup stamps s_memtime at
//   0 entry | 1 every load issued | 2 input image landed | 3 weights landed | 4 MFMAs done, partial tile in LDS | 5 barrier passed | 6 stored
// (stamps 2 / 3 add a counted s_waitcnt each; the un-stamped launch time is printed next to the stamped one) and s_memrealtime at
// entry / exit.  In situ stand-in: a graph of 24 x (plain 28 MB weight-streaming filler, projection over its own layer's weights).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../edgerunner_amd/csrc -o gemv_mfma_timeline_probe gemv_mfma_timeline_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ unsigned long long* g_tp_out;
#ifdef PROBE_STAMPS
#define ER_TPG(i)                                                                                             \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (threadIdx.x == 0) {                                                                               \
            unsigned long long* o_ = g_tp_out + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 10;        \
            o_[i] = __builtin_amdgcn_s_memtime();                                                             \
            if ((i) == 0) o_[8] = __builtin_amdgcn_s_memrealtime();                                           \
            if ((i) == 6) o_[9] = __builtin_amdgcn_s_memrealtime();                                           \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#define ER_TPG_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) + 2) : "memory")
#endif
#include "k_gemv_mfma.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace er;

typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void filler_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N) {
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
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;        // 0: fc1 form (N = 6144, ReLU, tiled output), 1: qkv form (N = 4608, KV append), 2: narrow out_proj form (4 waves, partials)
    const int NL = 24, K = 1536, NQ = 4608;
    const int N = kind == 0 ? 6144 : kind == 1 ? 4608 : 1536;
    const size_t wbytes = (size_t)N * K * 2;
    char* W;
    CHECK(hipMalloc(&W, wbytes * NL));
    {
        std::vector<unsigned short> h(wbytes / 2);
        unsigned s = 7u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2000 + ((s >> 9) & 0x0fff)); }
        for (int l = 0; l < NL; ++l) CHECK(hipMemcpy(W + l * wbytes, h.data(), wbytes, hipMemcpyHostToDevice));
    }
    char *xt, *xt_out;
    float *bias, *out, *q, *part, *Wf, *xa, *xb;
    void *kc, *vc;
    int* pos;
    CHECK(hipMalloc(&xt, (size_t)K * 128));
    CHECK(hipMemset(xt, 0, (size_t)K * 128));
    CHECK(hipMalloc(&xt_out, (size_t)6144 * 128));
    CHECK(hipMalloc(&bias, 6144 * 4));
    CHECK(hipMemset(bias, 0, 6144 * 4));
    CHECK(hipMalloc(&out, (size_t)32 * 6144 * 4));
    CHECK(hipMalloc(&q, (size_t)32 * 1536 * 4));
    CHECK(hipMalloc(&part, (size_t)16 * 32 * 6144 * 4));
    const int Lcap = 4160;
    const size_t kvb = (size_t)16 * Lcap * 96;
    CHECK(hipMalloc(&kc, kvb * 32 * 2));
    CHECK(hipMalloc(&vc, kvb * 32 * 2));
    CHECK(hipMalloc(&pos, 32 * 4));
    {
        std::vector<int> hp(32, 3000);
        CHECK(hipMemcpy(pos, hp.data(), 32 * 4, hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(&Wf, (size_t)NQ * 1536 * 4 * NL));
    CHECK(hipMemset(Wf, 0, (size_t)NQ * 1536 * 4 * NL));
    CHECK(hipMalloc(&xa, 8192 * 4));
    CHECK(hipMalloc(&xb, 8192 * 4));
    CHECK(hipMemset(xa, 0, 8192 * 4));
    unsigned long long* tp;
    CHECK(hipMalloc(&tp, (size_t)4096 * 10 * 8));
    CHECK(hipMemset(tp, 0, (size_t)4096 * 10 * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tp_out), &tp, sizeof(tp)));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipGraph_t graph;
    hipGraphExec_t gexec;
    int nwg = 0;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int l = 0; l < NL; ++l) {
        hipLaunchKernelGGL(filler_kernel, dim3(NQ / 4), dim3(256), 0, st, Wf + (size_t)l * NQ * 1536, xa, xb, NQ);
        GemvArgs a{};
        a.W = W + l * wbytes; a.bias = bias; a.N = N; a.xin = reinterpret_cast<const float*>(xt); a.out = out; a.xt_out = xt_out;
        a.q = q; a.kcache = kc; a.vcache = vc; a.kv_half = 1; a.hidden = 1536; a.head_dim = 96; a.l_cap = Lcap; a.kv_bstride = (long long)kvb; a.pos = pos;
        hipError_t e;
        if (kind == 0) { e = launch_gemv_mfma<_Float16, EPI_RELU, true>(a, B, K, part, st); nwg = N / 32; }
        else if (kind == 1) { e = launch_gemv_mfma<_Float16, EPI_QKV, true>(a, B, K, part, st); nwg = N / 32; }
        else { a.resid = out; e = launch_gemv_mfma<_Float16, EPI_RESID, true>(a, B, K, part, st, true, true); nwg = (N / 32) * 4; }
        CHECK(e);
    }
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(gexec, st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < 10; ++r) CHECK(hipGraphLaunch(gexec, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s, B = %d, %d workgroups: graph of 24 x (filler 28 MB, projection of %.1f MB) -> %.2f us per pair%s\n",
           kind == 0 ? "fc1 form (N 6144, ReLU, tiled out)" : kind == 1 ? "qkv form (N 4608, KV append)" : "out_proj narrow form (N 1536, 4-wave, partials)", B, nwg,
           wbytes / 1e6, ms * 1e3 / 10 / NL,
#ifdef PROBE_STAMPS
           "  [stamped build]"
#else
           "  [plain build]"
#endif
    );
#ifdef PROBE_STAMPS
    std::vector<unsigned long long> h((size_t)nwg * 10);
    CHECK(hipMemcpy(h.data(), tp, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0}, mx[8] = {0};
    unsigned long long smin = ~0ull, smax = 0, emax = 0, emin = ~0ull;
    for (int w = 0; w < nwg; ++w) {
        const unsigned long long* t = &h[(size_t)w * 10];
        smin = std::min(smin, t[8]); smax = std::max(smax, t[8]); emax = std::max(emax, t[9]); emin = std::min(emin, t[9]);
        for (int i = 1; i < 7; ++i) { const double c = (double)(t[i] - t[0]); acc[i] += c / nwg; mx[i] = std::max(mx[i], c); }
    }
    printf("last launch: entry skew %.2f us | first entry -> last exit %.2f us | exit skew %.2f us\n", (smax - smin) / 100.0, (emax - smin) / 100.0, (emax - emin) / 100.0);
    const char* names[7] = {"entry", "every load issued", "input image landed", "weights landed", "MFMAs done, tile in LDS", "barrier passed", "stored"};
    printf("stamps relative to the workgroup's entry, shader-clock cycles (mean over workgroups | max):\n");
    for (int i = 1; i < 7; ++i) printf("  %d %-26s %8.0f | %8.0f\n", i, names[i], acc[i], mx[i]);
#endif
    return 0;
}
