// Micro-benchmark for DESIGN.md section 10 item 1b: what does a hand-rolled grid barrier cost on MI355X?
// All workgroups are co-resident (grid <= 3 per CU), one arrives with an agent-scope atomic after a release fence,
// everyone spins (bounded) on the counter, acquire fence.  Prints microseconds per barrier for several grid sizes,
// with and without a global store per thread before the barrier (the release then has dirty lines to write back).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier_probe scripts/probes/grid_barrier_probe.hip && /tmp/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void probe(unsigned* counter, int* err, float* scratch, int iters, int with_store) {
    const int tid = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (with_store) scratch[(size_t)blockIdx.x * 256 + tid] = (float)it;
        __syncthreads();
        if (tid == 0) {
            __threadfence();                                              // release: data of this phase visible device-wide
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * gridDim.x;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22)) { *err = 1; break; }             // never hang the box
            }
            __threadfence();                                              // acquire
        }
        __syncthreads();
        if (*err) return;
    }
}

int main() {
    unsigned* counter; int* err; float* scratch;
    hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&scratch, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int with_store = 0; with_store < 2; ++with_store)
        for (int grid : {64, 256, 512, 768}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, counter, err, scratch, iters, with_store);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int h_err = 0; hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
            printf("{\"grid\": %d, \"store_before_barrier\": %d, \"us_per_barrier\": %.3f, \"spin_bound_hit\": %d}\n", grid, with_store,
                   best * 1000.0f / iters, h_err);
        }
    return 0;
}
