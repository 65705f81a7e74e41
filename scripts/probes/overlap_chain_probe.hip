// Does launching the NEXT weight-streaming kernel early - on a second queue, beside its predecessor - hide the ~2.7 us that
// every dependent launch of the B = 1 decode layer costs?  ("partial fusion ... behind ready counters
// rather than a grid barrier".)
//
// Stand-in layer as in persistent_chain_probe.hip (five GEMVs over rows of 1536 floats with the real byte volumes and the real
// all-to-all dependency: every phase reads the WHOLE output vector of the previous one).  Variants:
//   (A) one stream, graph-replayed, hardware kernel boundaries only (the launches baseline, same kernel body)
//   (E1) two streams, eager: even kernels on stream 0, odd kernels on stream 1.  Stream order gives K(n) -> K(n+2); the data
//        dependency K(n) -> K(n+1) is a per-XCD-sharded arrival counter: every workgroup of K(n) stores its outputs
//        write-through (sc1), drains them, and ONE lane adds 1 to shard blockIdx % 8; K(n+1) issues its first weight rows,
//        THEN one wave polls the 8 shards (relaxed agent-scope loads, s_sleep) and the input vector is read with sc1 loads.
//        So K(n+1) is resident, with its weights in flight, while K(n) still runs.
//   (E2) the same two chains captured into ONE hipGraph (fork / join events) and replayed.
// Deadlock freedom: a kernel never waits for a YOUNGER kernel, and every kernel fits twice on the chip (512 workgroups of
// <= 4 waves, <= 128 VGPRs: 2 per CU, 4 of them resident per CU), so whichever of K(n), K(n+1) the dispatcher places first
// the other still fits.  Spins are bounded; a give-up sets an error word, is reported, and never hangs the GPU.
//   hipcc --offload-arch=gfx950 -O3 -o overlap_chain_probe overlap_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) float gf32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int K = 1536, NPH = 5, NL = 24, XLEN = 8192, NWG = 512, SHARDS = 8, SHARD_STRIDE = 32;
static const int h_rows[NPH] = {4608, 8192, 1536, 6144, 6144};     // qkv, K/V stream at context ~4096, out_proj, fc1, fc2
static const int h_nw[NPH] = {3, 4, 3, 4, 4};                       // waves per workgroup: rows / 512 / nw = 3, 4, 1, 3, 3 rows per wave
constexpr unsigned SPIN_LIMIT = 400000u;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float squash(float v) { return v / (1.0f + fabsf(v)); }

// grid = 512 workgroups x NW waves; wave (b, w) owns rows (b * NW + w) * RPW .. + RPW, one row (6 x 1 KiB loads) per step,
// two row buffers in flight.  wait_cnt == nullptr: no software dependency (variant A).
template <int NW, int RPW>
__global__ __launch_bounds__(64 * NW, 4) void chain_kernel(const float* __restrict__ W, const float* xin, float* yout,
                                                           const unsigned* wait_cnt, unsigned* my_cnt, unsigned* err) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long row0 = (long long)(blockIdx.x * NW + wid) * RPW;
    const f32x4* wr = reinterpret_cast<const f32x4*>(W + row0 * K) + lane;
    f32x4 wa[6], wb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) wa[j] = __builtin_nontemporal_load(wr + j * 64);
    if (RPW > 1) {
#pragma unroll
        for (int j = 0; j < 6; ++j) wb[j] = __builtin_nontemporal_load(wr + (K / 4) + j * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wait_cnt != nullptr) {
        if (wid == 0) {
            unsigned spins = 0;
            for (;;) {
                const unsigned c = lane < SHARDS ? __hip_atomic_load((const gu32*)(wait_cnt + lane * SHARD_STRIDE), RLX_AGENT) : (unsigned)(NWG / SHARDS);
                if (__all(c >= (unsigned)(NWG / SHARDS))) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT || __hip_atomic_load((const gu32*)err, RLX_AGENT) != 0u) {
                    if (lane == 0) __hip_atomic_store((gu32*)err, 1u, RLX_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
    // the input vector: written write-through by the predecessor, read past this CU's L1 (sc1)
    f32x4 x[6];
    {
        const float* p0 = xin + lane * 4;
        const float* p1 = p0 + 3 * 256;
        asm volatile(
            "global_load_dwordx4 %0, %6, off sc1\n\t"
            "global_load_dwordx4 %1, %6, off offset:1024 sc1\n\t"
            "global_load_dwordx4 %2, %6, off offset:2048 sc1\n\t"
            "global_load_dwordx4 %3, %7, off sc1\n\t"
            "global_load_dwordx4 %4, %7, off offset:1024 sc1\n\t"
            "global_load_dwordx4 %5, %7, off offset:2048 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5])
            : "v"(p0), "v"(p1)
            : "memory");
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        f32x4 (&cur)[6] = (r & 1) ? wb : wa;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            s = fmaf(cur[j].x, x[j].x, s); s = fmaf(cur[j].y, x[j].y, s); s = fmaf(cur[j].z, x[j].z, s); s = fmaf(cur[j].w, x[j].w, s);
        }
        s = wave_sum(s);
        if (lane == 0) __hip_atomic_store((gf32*)(yout + row0 + r), squash(s), RLX_AGENT);
        if (r + 2 < RPW) {
#pragma unroll
            for (int j = 0; j < 6; ++j) cur[j] = __builtin_nontemporal_load(wr + (long long)(r + 2) * (K / 4) + j * 64);
        }
    }
    if (my_cnt != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*)(my_cnt + (blockIdx.x % SHARDS) * SHARD_STRIDE), 1u, RLX_AGENT);
    }
}

static void launch_phase(int p, const float* W, const float* xin, float* yout, const unsigned* wait_cnt, unsigned* my_cnt, unsigned* err,
                         hipStream_t st) {
    switch (p) {
        case 0: hipLaunchKernelGGL((chain_kernel<3, 3>), dim3(NWG), dim3(192), 0, st, W, xin, yout, wait_cnt, my_cnt, err); break;
        case 1: hipLaunchKernelGGL((chain_kernel<4, 4>), dim3(NWG), dim3(256), 0, st, W, xin, yout, wait_cnt, my_cnt, err); break;
        case 2: hipLaunchKernelGGL((chain_kernel<3, 1>), dim3(NWG), dim3(192), 0, st, W, xin, yout, wait_cnt, my_cnt, err); break;
        default: hipLaunchKernelGGL((chain_kernel<4, 3>), dim3(NWG), dim3(256), 0, st, W, xin, yout, wait_cnt, my_cnt, err); break;
    }
}

static void fill(std::vector<float>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 8388608.0f - 1.0f) * scale; }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    long long per_layer = 0;
    for (int p = 0; p < NPH; ++p) {
        per_layer += (long long)h_rows[p] * K;
        if (h_rows[p] % (NWG * h_nw[p]) != 0) { printf("bad shape\n"); return 1; }
    }
    printf("layer stand-in: %.1f MB of weights per layer, %d layers (%.2f GB)\n", per_layer * 4 / 1e6, NL, per_layer * 4.0 * NL / 1e9);
    float* W;
    CHECK(hipMalloc(&W, per_layer * NL * sizeof(float)));
    {
        std::vector<float> h(per_layer);
        for (int l = 0; l < NL; ++l) {
            fill(h, 1234u + l, 0.03f);
            CHECK(hipMemcpy(W + l * per_layer, h.data(), per_layer * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    std::vector<float> x0(XLEN);
    fill(x0, 77u, 1.0f);
    float *xa, *xb;
    CHECK(hipMalloc(&xa, XLEN * sizeof(float)));
    CHECK(hipMalloc(&xb, XLEN * sizeof(float)));
    const int nk = NL * NPH;
    unsigned* cnt;      // [nk][SHARDS][SHARD_STRIDE] + error word
    const size_t cnt_words = (size_t)nk * SHARDS * SHARD_STRIDE + 32;
    CHECK(hipMalloc(&cnt, cnt_words * sizeof(unsigned)));
    unsigned* err = cnt + (size_t)nk * SHARDS * SHARD_STRIDE;
    hipStream_t s0, s1;
    CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int reps = 20;
    auto reset_x = [&]() {
        CHECK(hipMemcpy(xa, x0.data(), XLEN * sizeof(float), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(xb, x0.data(), XLEN * sizeof(float), hipMemcpyHostToDevice));
    };
    // chained = software dependencies between consecutive kernels; two = alternate the streams
    auto enqueue = [&](bool chained, hipStream_t sa, hipStream_t sb) {
        int gp = 0;
        for (int l = 0; l < NL; ++l) {
            long long off = 0;
            for (int p = 0; p < NPH; ++p, ++gp) {
                const float* Wp = W + l * per_layer + off;
                const float* xin = (gp & 1) ? xb : xa;
                float* yout = (gp & 1) ? xa : xb;
                const unsigned* wc = (chained && gp > 0) ? cnt + (size_t)(gp - 1) * SHARDS * SHARD_STRIDE : nullptr;
                unsigned* mc = chained ? cnt + (size_t)gp * SHARDS * SHARD_STRIDE : nullptr;
                launch_phase(p, Wp, xin, yout, wc, mc, err, (gp & 1) ? sb : sa);
                off += (long long)h_rows[p] * K;
            }
        }
    };
    std::vector<float> refA(XLEN), got(XLEN);

    // ---- (A) one stream, graph, hardware boundaries
    {
        hipGraph_t graph;
        hipGraphExec_t gexec;
        CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        enqueue(false, s0, s0);
        CHECK(hipStreamEndCapture(s0, &graph));
        CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        reset_x();
        CHECK(hipGraphLaunch(gexec, s0));
        CHECK(hipStreamSynchronize(s0));
        CHECK(hipMemcpy(refA.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));
        CHECK(hipEventRecord(e0, s0));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(gexec, s0));
        CHECK(hipEventRecord(e1, s0));
        CHECK(hipStreamSynchronize(s0));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps / NL;
        printf("(A)  one stream, graph, 5 launches per layer        : %7.2f us per layer  (%.2f TB/s)\n", us, per_layer * 4 / us / 1e6);
    }
    // ---- (A2) the same chain WITH the counters on one stream (the price of arrive + poll when nothing overlaps)
    {
        hipGraph_t graph;
        hipGraphExec_t gexec;
        CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        CHECK(hipMemsetAsync(cnt, 0, cnt_words * sizeof(unsigned), s0));
        enqueue(true, s0, s0);
        CHECK(hipStreamEndCapture(s0, &graph));
        CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        reset_x();
        CHECK(hipGraphLaunch(gexec, s0));
        CHECK(hipStreamSynchronize(s0));
        CHECK(hipMemcpy(got.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));
        const bool ok = memcmp(got.data(), refA.data(), 6144 * sizeof(float)) == 0;
        CHECK(hipEventRecord(e0, s0));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(gexec, s0));
        CHECK(hipEventRecord(e1, s0));
        CHECK(hipStreamSynchronize(s0));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps / NL;
        unsigned herr = 0;
        CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("(A2) one stream, graph, counters armed              : %7.2f us per layer  (%.2f TB/s)  %s%s\n", us, per_layer * 4 / us / 1e6,
               ok ? "bit-identical" : "DIFFERS", herr ? "  [SPIN TIMEOUT]" : "");
    }
    const int mode = argc > 1 ? atoi(argv[1]) : 3;     // bit 0: eager two-stream, bit 1: two-chain graph
    // ---- (E1) two streams, eager
    if (mode & 1) {
        float best = 1e9f;
        bool ok = true;
        unsigned herr = 0;
        for (int r = 0; r < 6 && !herr; ++r) {
            reset_x();
            CHECK(hipMemsetAsync(cnt, 0, cnt_words * sizeof(unsigned), s0));
            CHECK(hipEventRecord(e0, s0));
            CHECK(hipEventRecord(fork, s0));
            CHECK(hipStreamWaitEvent(s1, fork, 0));
            enqueue(true, s0, s1);
            CHECK(hipEventRecord(join, s1));
            CHECK(hipStreamWaitEvent(s0, join, 0));
            CHECK(hipEventRecord(e1, s0));
            CHECK(hipStreamSynchronize(s0));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(got.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));
            ok &= memcmp(got.data(), refA.data(), 6144 * sizeof(float)) == 0;
        }
        const double us = best * 1000.0 / NL;
        printf("(E1) two streams, eager, early launch + counters    : %7.2f us per layer  (%.2f TB/s)  %s%s\n", us, per_layer * 4 / us / 1e6,
               ok ? "bit-identical" : "DIFFERS", herr ? "  [SPIN TIMEOUT]" : "");
    }
    // ---- (E2) two chains in one graph
    if (mode & 2) {
        hipGraph_t graph;
        hipGraphExec_t gexec;
        CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        CHECK(hipMemsetAsync(cnt, 0, cnt_words * sizeof(unsigned), s0));
        CHECK(hipEventRecord(fork, s0));
        CHECK(hipStreamWaitEvent(s1, fork, 0));
        enqueue(true, s0, s1);
        CHECK(hipEventRecord(join, s1));
        CHECK(hipStreamWaitEvent(s0, join, 0));
        CHECK(hipStreamEndCapture(s0, &graph));
        CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        reset_x();
        CHECK(hipGraphLaunch(gexec, s0));
        CHECK(hipStreamSynchronize(s0));
        CHECK(hipMemcpy(got.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));
        bool ok = memcmp(got.data(), refA.data(), 6144 * sizeof(float)) == 0;
        unsigned herr = 0;
        CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        double us = 0;
        if (!herr) {
            CHECK(hipEventRecord(e0, s0));
            for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(gexec, s0));
            CHECK(hipEventRecord(e1, s0));
            CHECK(hipStreamSynchronize(s0));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            us = ms * 1000.0 / reps / NL;
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(got.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));
            ok &= memcmp(got.data(), refA.data(), 6144 * sizeof(float)) == 0;
        }
        printf("(E2) two chains in ONE graph, early launch + counters: %7.2f us per layer  (%.2f TB/s)  %s%s\n", us, us > 0 ? per_layer * 4 / us / 1e6 : 0.0,
               ok ? "bit-identical" : "DIFFERS", herr ? "  [SPIN TIMEOUT]" : "");
    }
    return 0;
}
