// Round 6: the streamed LDS-DMA GEMM (k_gemm_stream.h, loader + matrix waves, five-stage ring, persistent tile stream) against the
// 4-wave and 256 x 256 kernels of k_gemm.h on the DiT shapes: throughput AND bit-equality of every epilogue form
// (plain with bias / gate / residual / fp16 copy, V^T of a fused q/k/v projection, GEGLU), ragged shapes included.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I ../../edgerunner_amd/csrc -o gemm_stream_probe gemm_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "k_gemm_stream.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace er;

static unsigned rng = 7u;
static float urand() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 8388608.0f - 1.0f; }

template <class T>
static long long mismatches(const T* d_a, const T* d_b, size_t n) {
    std::vector<T> a(n), b(n);
    CHECK(hipMemcpy(a.data(), d_a, n * sizeof(T), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), d_b, n * sizeof(T), hipMemcpyDeviceToHost));
    long long bad = 0;
    for (size_t i = 0; i < n; ++i) bad += memcmp(&a[i], &b[i], sizeof(T)) != 0;
    return bad;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time_us = [&](auto&& run) {
        CHECK(run());
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHECK(run());
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
    };
    // mode 0: plain epilogue (bias + gate + residual + fp16 copy), 1: fused q/k/v with V^T, 2: GEGLU
    struct Shape { int M, N, K, mode; };
    const Shape shapes[] = {{4096, 1024, 1024, 0}, {4096, 1024, 4096, 0}, {4096, 3072, 1024, 1}, {4096, 8192, 1024, 2}, {4096, 4096, 4096, 0},
                            {3001, 4164, 192, 0}, {77, 200, 640, 0}, {129, 65, 128, 0}, {384, 768, 192, 1}, {300, 256, 192, 2}, {8192, 8192, 8192, 0}};
    const int only = argc > 2 ? atoi(argv[2]) : -1;        // index of the single shape to run
    int idx = -1;
    for (auto& s : shapes) {
        ++idx;
        if (only >= 0 && idx != only) continue;
        const int M = s.M, N = s.N, K = s.K;
        _Float16 *A, *B, *c16[2], *vt[2];
        float *C[2], *bias, *resid, *gate;
        CHECK(hipMalloc(&A, (size_t)M * K * 2));
        CHECK(hipMalloc(&B, (size_t)N * K * 2));
        CHECK(hipMalloc(&bias, (size_t)N * 4));
        CHECK(hipMalloc(&resid, (size_t)M * N * 4));
        CHECK(hipMalloc(&gate, (size_t)2 * N * 4));
        for (int v = 0; v < 2; ++v) {
            CHECK(hipMalloc(&C[v], (size_t)M * N * 4));
            CHECK(hipMalloc(&c16[v], (size_t)M * N * 2));
            CHECK(hipMalloc(&vt[v], (size_t)M * N * 2));
            CHECK(hipMemset(C[v], 0, (size_t)M * N * 4));
            CHECK(hipMemset(c16[v], 0, (size_t)M * N * 2));
            CHECK(hipMemset(vt[v], 0, (size_t)M * N * 2));
        }
        {
            std::vector<_Float16> h((size_t)M * K), hb((size_t)N * K);
            for (auto& v : h) v = (_Float16)urand();
            for (auto& v : hb) v = (_Float16)(urand() * 0.05f);
            CHECK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
            std::vector<float> f((size_t)M * N);
            for (auto& v : f) v = urand();
            CHECK(hipMemcpy(resid, f.data(), f.size() * 4, hipMemcpyHostToDevice));
            std::vector<float> fb((size_t)2 * N);
            for (auto& v : fb) v = urand();
            CHECK(hipMemcpy(bias, fb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(gate, fb.data(), (size_t)2 * N * 4, hipMemcpyHostToDevice));
        }
        auto args = [&](int v) {
            GemmArgs g = gemm_args_default();
            g.A = reinterpret_cast<const float*>(A); g.B = reinterpret_cast<const float*>(B); g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K;
            g.ldc = N; g.ldr = N; g.bias = bias; g.c16 = c16[v]; g.ldc16 = N;
            if (s.mode == 0) {
                g.C = C[v]; g.resid = resid;
                if (M % 2 == 0) { g.gate = gate; g.gate_rows = M / 2; g.gate_bstride = N; }
            } else if (s.mode == 1) {
                const int rows = M >= 2048 ? 2048 : 64;
                g.vt16 = vt[v]; g.vt_col0 = 2 * (N / 3); g.vt_rows = rows; g.vt_ld = rows;
            } else {
                g.ldc16 = N / 2;
            }
            return g;
        };
        const double flop = 2.0 * M * N * K;
        const GemmArgs g0 = args(0), g1 = args(1);
        const int ref_tile = s.mode == 2 ? 1 : 0;
        auto run_ref = [&]() { return s.mode == 2 ? launch_gemm_hh_geglu(g0, st, ref_tile) : launch_gemm_hh(g0, st, gemm_hh_pick_tile(g0.M, g0.N)); };   // (force_tile: the k_gemm.h kernels only)
        #ifdef GS_TIMELINE
        static float* dbg_all = nullptr;
        if (!dbg_all) CHECK(hipMalloc(&dbg_all, 256 * 8 * 4));
        GemmArgs g1t = g1;
        g1t.q = dbg_all;
        auto run_new = [&]() { return launch_gemm_hh_stream(g1t, st, s.mode == 2); };
#else
        auto run_new = [&]() { return launch_gemm_hh_stream(g1, st, s.mode == 2); };
#endif
        const double t_ref = time_us(run_ref), t_new = time_us(run_new);
#ifdef GS_TIMELINE
        {
            float* dbg;
            CHECK(hipMalloc(&dbg, 256 * 8 * 4));
            CHECK(hipMemset(dbg, 0, 256 * 8 * 4));
            GemmArgs gd = g1;
            gd.q = dbg;
            CHECK(launch_gemm_hh_stream(gd, st, s.mode == 2));
            CHECK(hipStreamSynchronize(st));
            float h[256 * 8];
            CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            hipFree(dbg);
            double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int w = 0; w < 256; ++w) for (int q = 0; q < 8; ++q) m[q] += h[w * 8 + q] / 256.0;
            printf("   cycles per k-step (mean over 256 workgroups): loader issue %.0f  wait %.0f  barrier %.0f | matrix wave work %.0f  barrier %.0f | per tile: epilogue issued %.0f  stores drained +%.0f\n", m[0], m[1], m[2], m[4], m[5], m[6], m[7]);
        }
#endif
        long long bad = 0;
        if (s.mode == 0) bad += mismatches(C[0], C[1], (size_t)M * N);
        bad += mismatches(c16[0], c16[1], (size_t)M * (s.mode == 2 ? N / 2 : N));
        if (s.mode == 1) bad += mismatches(vt[0], vt[1], (size_t)M * (N / 3));
        const char* mn[3] = {"plain+gate+resid", "qkv + V^T", "GEGLU"};
        printf("%5d x %5d x %5d  %-17s  k_gemm rule %8.1f us %7.1f TF   stream %8.1f us %7.1f TF   x%.2f   mismatches %lld\n", M, N, K, mn[s.mode],
               t_ref, flop / t_ref / 1e6, t_new, flop / t_new / 1e6, t_ref / t_new, bad);
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(bias); hipFree(resid); hipFree(gate);
        for (int v = 0; v < 2; ++v) { hipFree(C[v]); hipFree(c16[v]); hipFree(vt[v]); }
    }
    return 0;
}
