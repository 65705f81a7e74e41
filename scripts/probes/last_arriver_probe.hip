// Micro-benchmark for DESIGN.md section 10 item 1a: is "the last workgroup of a group merges" (release fence + agent-scope
// atomic counter, no spinning) cheaper than a second kernel?  768 workgroups = 16 groups of 48 (the attention grid at
// L = 6050): each writes a 98-float partial; variant B adds fence + counter and lets the last arriver of each group read
// its 48 partials and write 96 floats; variant C is the same merge as a separate 16-workgroup kernel after variant A.
// Chains of launches replayed from a hipGraph, microseconds per chain element.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int GROUPS = 16, PER = 48, W = 98;

__global__ __launch_bounds__(256) void partial_only(float* part) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    if (tid < W) part[(size_t)wg * W + tid] = (float)(wg + tid);
}
__device__ void merge(const float* part, float* out, int g, int tid) {
    if (tid < 96) {
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < PER; ++k) s += part[((size_t)g * PER + k) * W + 2 + tid];
        out[g * 96 + tid] = s;
    }
}
__global__ __launch_bounds__(256) void partial_fused(float* part, float* out, unsigned* cnt) {
    __shared__ int last;
    const int wg = blockIdx.x, tid = threadIdx.x, g = wg / PER;
    if (tid < W) part[(size_t)wg * W + tid] = (float)(wg + tid);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&cnt[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old == PER - 1);
        if (last) cnt[g] = 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    merge(part, out, g, tid);
}
__global__ __launch_bounds__(256) void merge_only(const float* part, float* out) { merge(part, out, blockIdx.x, threadIdx.x); }

int main() {
    float *part, *out; unsigned* cnt;
    (void)hipMalloc(&part, (size_t)GROUPS * PER * W * 4); (void)hipMalloc(&out, GROUPS * 96 * 4); (void)hipMalloc(&cnt, GROUPS * 4);
    (void)hipMemset(cnt, 0, GROUPS * 4);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int chain = 48, replays = 100;
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
        for (int i = 0; i < chain; ++i) {
            if (variant == 0) hipLaunchKernelGGL(partial_only, dim3(GROUPS * PER), dim3(256), 0, st, part);
            else if (variant == 1) hipLaunchKernelGGL(partial_fused, dim3(GROUPS * PER), dim3(256), 0, st, part, out, cnt);
            else {
                hipLaunchKernelGGL(partial_only, dim3(GROUPS * PER), dim3(256), 0, st, part);
                hipLaunchKernelGGL(merge_only, dim3(GROUPS), dim3(256), 0, st, part, out);
            }
        }
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < replays; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const char* names[3] = {"partials only", "partials + last-arriver merge (fused)", "partials kernel + merge kernel"};
        printf("{\"variant\": \"%s\", \"us_per_step\": %.3f}\n", names[variant], ms * 1000.0f / (chain * replays));
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    unsigned h[GROUPS]; (void)hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
    unsigned bad = 0; for (unsigned v : h) bad |= v;
    printf("{\"counters_back_to_zero\": %s}\n", bad ? "false" : "true");
    return 0;
}
