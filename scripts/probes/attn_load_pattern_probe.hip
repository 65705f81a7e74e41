// Does the fp16 KV cache's 192-byte row cost the decode attention bandwidth?
//
// attn_decode3_kernel<_Float16> reads a key row (96 halves = 192 bytes) with 4 lanes x 3 sixteen-byte loads: load j of a wave
// covers, for 16 consecutive keys, the j-th 64-byte THIRD of each row - sixteen half-used 128-byte lines per wave-instruction,
// every line requested by two different instructions.  The fp32 cache (384-byte rows, 8 lanes x 3 loads) requests whole lines.
// This probe streams the same bytes with both address patterns and with the fully contiguous one (lane l, load j -> chunk 64 j + l
// of the wave's 3 KB), at the workgroup shape of the real kernel (256 workgroups of 16 waves, every load issued before the first
// use), 24 launches over 24 distinct buffers per graph replay, `steps` wave-steps of 16 keys per wave.
//   hipcc --offload-arch=gfx950 -O3 -o attn_load_pattern_probe attn_load_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PAT 0: thirds (the product's fp16 pattern)   byte = key * 192 + j * 64 + p * 16,  key = lane / 4, p = lane % 4
// PAT 1: contiguous                            byte = (64 j + lane) * 16
// PAT 2: lane-contiguous 48 bytes              byte = lane * 48 + j * 16
// PAT 3: whole lines of 8 fp32 rows (product, fp32 cache: 384-byte rows)   byte = key * 384 + j * 128 + p * 16,  key = lane / 8, p = lane % 8
template <int PAT, int STEPS>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ K, const char* __restrict__ V, float* __restrict__ out, long long wg_bytes) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f32x4 k[STEPS][3], v[STEPS][3];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const long long base = (long long)blockIdx.x * wg_bytes + (long long)(i * 16 + wid) * 3072;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int off = PAT == 0 ? (lane >> 2) * 192 + j * 64 + (lane & 3) * 16 : PAT == 1 ? (64 * j + lane) * 16 : PAT == 2 ? lane * 48 + j * 16 : (lane >> 3) * 384 + j * 128 + (lane & 7) * 16;
            k[i][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(K + base + off));
        }
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const long long base = (long long)blockIdx.x * wg_bytes + (long long)(i * 16 + wid) * 3072;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int off = PAT == 0 ? (lane >> 2) * 192 + j * 64 + (lane & 3) * 16 : PAT == 1 ? (64 * j + lane) * 16 : PAT == 2 ? lane * 48 + j * 16 : (lane >> 3) * 384 + j * 128 + (lane & 7) * 16;
            v[i][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(V + base + off));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s += k[i][j].x * v[i][j].x + k[i][j].y * v[i][j].y + k[i][j].z * v[i][j].z + k[i][j].w * v[i][j].w;
    if (s == 12345.f) out[threadIdx.x] = s;
}

int main() {
    constexpr int NL = 24;
    const long long cap = 256LL * 4 * 16 * 3072;        // bytes of K (or V) per layer at 2 steps: 256 workgroups x 32 wave-steps x 3 KB = 25.2 MB
    char *K, *V;
    float* out;
    CHECK(hipMalloc(&K, cap * NL));
    CHECK(hipMalloc(&V, cap * NL));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(K, 0, cap * NL));
    CHECK(hipMemset(V, 0, cap * NL));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time_it = [&](int pat, int steps) {
        hipGraph_t g;
        hipGraphExec_t ge;
        const long long wgb = (long long)steps * 16 * 3072;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int l = 0; l < NL; ++l) {
            const char *k = K + l * cap, *v = V + l * cap;
#define LAUNCH(P, S) hipLaunchKernelGGL((stream_kernel<P, S>), dim3(256), dim3(1024), 0, st, k, v, out, wgb)
            if (steps == 1) { if (pat == 0) LAUNCH(0, 1); else if (pat == 1) LAUNCH(1, 1); else if (pat == 2) LAUNCH(2, 1); else LAUNCH(3, 1); }
            else if (steps == 2) { if (pat == 0) LAUNCH(0, 2); else if (pat == 1) LAUNCH(1, 2); else if (pat == 2) LAUNCH(2, 2); else LAUNCH(3, 2); }
            else { if (pat == 1) LAUNCH(1, 4); else LAUNCH(3, 4); }
        }
        CHECK(hipStreamEndCapture(st, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge, st));
        CHECK(hipStreamSynchronize(st));
        const int reps = 20;
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
        const double us = ms * 1000.0 / reps / NL, mb = 2.0 * 256 * wgb / 1e6;
        printf("  %-34s %d step(s) = %5.1f MB per launch: %6.2f us per launch (%.2f TB/s)\n",
               pat == 0 ? "thirds of 16 rows (product, fp16)" : pat == 1 ? "contiguous 1 KB per instruction" : pat == 2 ? "48 contiguous bytes per lane" : "lines of 8 rows (product, fp32)", steps, mb, us, mb / us);
    };
    for (int rep = 0; rep < 2; ++rep)
    {
        for (int steps = 1; steps <= 2; ++steps)
            for (int pat = 0; pat < 4; ++pat) time_it(pat, steps);
        time_it(1, 4);
        time_it(3, 4);
    }
    return 0;
}
