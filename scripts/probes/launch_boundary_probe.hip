// Micro-benchmark: what does one dependent kernel boundary cost inside a replayed hipGraph, as a function of the grid
// size (empty kernels, and kernels that read one 16-byte value per thread)?  Calibrates the "~3 us + bytes / 6 TB/s"
// model of the decode kernels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lbp scripts/probes/launch_boundary_probe.hip && /tmp/lbp
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 9999) *p = 0.f; }
__global__ __launch_bounds__(256) void touch_kernel(const float4* src, float* sink) {
    const float4 v = src[(size_t)blockIdx.x * 256 + threadIdx.x];
    if (v.x == 1.2345e-30f) *sink = v.y;
}

int main() {
    float4* src; float* sink;
    (void)hipMalloc(&src, (size_t)4096 * 256 * 16); (void)hipMalloc(&sink, 4);
    (void)hipMemset(src, 0, (size_t)4096 * 256 * 16);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int chain = 146, replays = 50;            // one decode step has 146 launches
    for (int touch = 0; touch < 2; ++touch)
        for (int grid : {1, 256, 768, 1152, 4096}) {
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            for (int i = 0; i < chain; ++i) {
                if (touch) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, st, src, sink);
                else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, sink);
            }
            (void)hipStreamEndCapture(st, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < replays; ++r) (void)hipGraphLaunch(ge, st);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("{\"kernel\": \"%s\", \"grid\": %d, \"us_per_launch_in_graph\": %.3f}\n", touch ? "touch16B" : "empty", grid,
                   ms * 1000.0f / (chain * replays));
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    return 0;
}
