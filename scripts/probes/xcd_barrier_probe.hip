// Grid-barrier latency on MI355X: flat single-counter barrier (what round 1 measured: 15 us at 256 workgroups) vs the
// XCD-hierarchical barrier of MI355X_MICROARCH.md (price list row `barrier-xcd`: per-XCC counter, the last arriver of
// an XCC is its leader and arrives on the top counter, the last leader publishes a per-XCC generation word; every
// waiter polls only its own XCC's word with relaxed loads + s_sleep, then ONE agent-scope acquire).
// Every spin is bounded (a stuck barrier sets an error word and lets the kernel finish).
//   hipcc --offload-arch=gfx950 -O3 -o xcd_barrier_probe xcd_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct BarState {            // every word on its own 128-byte line
    unsigned xcc_count[8 * 32];
    unsigned xcc_gen[8 * 32];
    unsigned census[8 * 32];
    unsigned top[32];
    unsigned flat[32];
    unsigned start[32];
    unsigned error[32];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}

constexpr unsigned SPIN_LIMIT = 4000000u;

__device__ __forceinline__ bool spin_until_ge(unsigned* word, unsigned want, unsigned* err) {
    unsigned spins = 0;
    while (__hip_atomic_load((gu32*)word, RLX_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT || __hip_atomic_load((gu32*)err, RLX_AGENT) != 0) {
            __hip_atomic_store((gu32*)err, 1u, RLX_AGENT);
            return false;
        }
    }
    return true;
}

// flat: one monotonic counter
__device__ __forceinline__ void barrier_flat(BarState* s, unsigned epoch, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add((gu32*)&s->flat[0], 1u, RLX_AGENT);
        spin_until_ge(&s->flat[0], epoch * nwg, &s->error[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// hierarchical: xcc counter -> top counter -> per-xcc generation
__device__ __forceinline__ void barrier_xcd(BarState* s, unsigned epoch, unsigned xcc, unsigned n_on_xcc, unsigned n_xcc, bool publish) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (publish) {   // this workgroup wrote data other workgroups will read after the barrier
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned old = __hip_atomic_fetch_add((gu32*)&s->xcc_count[xcc * 32], 1u, RLX_AGENT);
        if (old == epoch * n_on_xcc - 1u) {
            const unsigned old2 = __hip_atomic_fetch_add((gu32*)&s->top[0], 1u, RLX_AGENT);
            if (old2 == epoch * n_xcc - 1u) {
#pragma unroll
                for (int x = 0; x < 8; ++x) __hip_atomic_store((gu32*)&s->xcc_gen[x * 32], epoch, RLX_AGENT);
            }
        }
        spin_until_ge(&s->xcc_gen[xcc * 32], epoch, &s->error[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void barrier_loop(BarState* s, int iters, float* payload, int publish) {
    const unsigned nwg = gridDim.x;
    const unsigned xcc = xcc_id();
    __shared__ unsigned sh_n, sh_nx;
    // census: who is on which XCC (flat start barrier; all workgroups must be co-resident)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add((gu32*)&s->census[xcc * 32], 1u, RLX_AGENT);
        __hip_atomic_fetch_add((gu32*)&s->start[0], 1u, RLX_AGENT);
        spin_until_ge(&s->start[0], nwg, &s->error[0]);
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += __hip_atomic_load((gu32*)&s->census[x * 32], RLX_AGENT) > 0 ? 1u : 0u;
        sh_n = __hip_atomic_load((gu32*)&s->census[xcc * 32], RLX_AGENT);
        sh_nx = nx;
    }
    __syncthreads();
    const unsigned n_on_xcc = sh_n, n_xcc = sh_nx;
    float acc = 0.f;
    for (int it = 1; it <= iters; ++it) {
        if (publish) {      // a 128-byte record per workgroup, re-read from the neighbour after the barrier
            if (threadIdx.x < 32) payload[blockIdx.x * 32 + threadIdx.x] = (float)it;
        }
        if (MODE == 0) barrier_flat(s, (unsigned)it, nwg);
        else barrier_xcd(s, (unsigned)it, xcc, n_on_xcc, n_xcc, publish != 0);
        if (publish && threadIdx.x < 32) {
            const float v = payload[((blockIdx.x + 1) % nwg) * 32 + threadIdx.x];
            // the neighbour may already have entered iteration it + 1 and rewritten its record (one barrier per
            // iteration): it + 1 is fresh too; anything OLDER than `it` is a stale read
            if (v != (float)it && v != (float)(it + 1)) __hip_atomic_store((gu32*)&s->error[1], (unsigned)it, RLX_AGENT);
            acc += v;
        }
    }
    if (acc == -1.f) payload[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    BarState* s;
    float* payload;
    CHECK(hipMalloc(&s, sizeof(BarState)));
    CHECK(hipMalloc(&payload, 4096 * 32 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 200;
    for (int publish = 0; publish <= 1; ++publish)
        for (int mode = 0; mode <= 1; ++mode)
            for (int nwg : {64, 256, 512, 1024}) {
                float best = 1e9f;
                unsigned err[2] = {0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipMemset(s, 0, sizeof(BarState)));
                    CHECK(hipMemset(payload, 0, 4096 * 32 * sizeof(float)));
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(barrier_loop<0>, dim3(nwg), dim3(256), 0, 0, s, iters, payload, publish);
                    else hipLaunchKernelGGL(barrier_loop<1>, dim3(nwg), dim3(256), 0, 0, s, iters, payload, publish);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipDeviceSynchronize());
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    BarState h;
                    CHECK(hipMemcpy(&h, s, sizeof(BarState), hipMemcpyDeviceToHost));
                    err[0] |= h.error[0];
                    err[1] |= h.error[1];
                    if (rep == 0 && mode == 1 && nwg == 256 && publish == 0) {
                        printf("census (workgroups per XCC): ");
                        for (int x = 0; x < 8; ++x) printf("%u ", h.census[x * 32]);
                        printf("\n");
                    }
                }
                printf("%-22s %-8s workgroups %4d: %7.2f us per barrier%s%s\n", mode == 0 ? "flat counter" : "xcd-hierarchical",
                       publish ? "publish" : "bare", nwg, best * 1000.f / iters, err[0] ? "  [SPIN TIMEOUT]" : "", err[1] ? "  [STALE READ]" : "");
            }
    return 0;
}
