// Decode-layer stand-in as (A) five graph-replayed weight-streaming kernels per layer vs (B) ONE persistent launch for
// all 24 layers with an XCD-hierarchical grid barrier per phase and the NEXT phase's weights prefetched into registers
// BEFORE the barrier wait (the experiment VERDICT r1 asked to redo with the guide's barrier instead of the single-counter
// probe).  (C) is (B) with the prefetch issued after the barrier, isolating what the prefetch buys.
// (D) [round 2's staged variant: the barrier replaced by a data-tagged all-gather of 8-byte {value, phase tag} granules] was run in
// round 3 - 74.6 us per layer, the slowest of all (profiles/r03_persistent_chain_probe_D.log) - and removed from this file.
// (F) replaces the barrier tree of (B) by flat arrival counters sharded per XCD that every CU polls itself, the next phase's
// input read with sc1 loads - no fence at all: 56.2 us per layer (profiles/r03_persistent_chain_probe_F.log): the poll and the
// input read queue behind the CU's own prefetched weight stream.
//
// The layer keeps the real byte volumes and the real all-to-all dependency structure of the B = 1 fp32 decode layer
// (every phase needs the WHOLE output vector of the previous one), with simplified arithmetic: five GEMVs over rows of
// 1536 floats, N = 4608 (qkv, 28.3 MB), 8100 (the K/V stream at context 4050, 49.8 MB), 1536 (out_proj, 9.4 MB),
// 6144 (fc1, 37.7 MB), 6144 (fc2's 37.7 MB), x_next = tanh-squashed first 1536 outputs.  Same per-lane fmaf chain and
// wave reduction in all variants -> results must be bit-identical.
//   hipcc --offload-arch=gfx950 -O3 -o persistent_chain_probe persistent_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) float gf32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int K = 1536, NPH = 5, NL = 24, XLEN = 8192;
__constant__ int c_rows[NPH] = {4608, 8100, 1536, 6144, 6144};
static const int h_rows[NPH] = {4608, 8100, 1536, 6144, 6144};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float squash(float v) { return v / (1.0f + fabsf(v)); }
__device__ __forceinline__ float dot_row(const f32x4 (&w)[6], const f32x4 (&x)[6]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        s = fmaf(w[j].x, x[j].x, s); s = fmaf(w[j].y, x[j].y, s); s = fmaf(w[j].z, x[j].z, s); s = fmaf(w[j].w, x[j].w, s);
    }
    return wave_sum(s);
}

// ---------------------------------------------------------------- (A) one kernel per phase, 4 waves x RW rows
template <int RW>
__global__ __launch_bounds__(256) void phase_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wid) * RW;
    f32x4 w[RW][6];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + (long long)min(row0 + r, N - 1) * K);
#pragma unroll
        for (int j = 0; j < 6; ++j) w[r][j] = __builtin_nontemporal_load(wr + j * 64 + lane);
    }
    f32x4 x[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = reinterpret_cast<const f32x4*>(xin)[j * 64 + lane];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float s = dot_row(w[r], x);
        if (lane == 0 && row0 + r < N) yout[row0 + r] = squash(s);
    }
}

// ---------------------------------------------------------------- (B)/(C) persistent
struct BarState {
    unsigned xcc_count[8 * 32];
    unsigned xcc_gen[8 * 32];
    unsigned census[8 * 32];
    unsigned top[32];
    unsigned start[32];
    unsigned error[32];
    unsigned shard[8 * 32];    // (F) flat arrival counters, one 128-byte line per shard (cu % 8)
};
constexpr unsigned SPIN_LIMIT = 4000000u;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ void spin_until_ge(unsigned* word, unsigned want, unsigned* err) {
    unsigned spins = 0;
    while (__hip_atomic_load((gu32*)word, RLX_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || __hip_atomic_load((gu32*)err, RLX_AGENT) != 0) {
            __hip_atomic_store((gu32*)err, 1u, RLX_AGENT);
            return;
        }
    }
}

constexpr int PW = 16, CW = PW - 1, PT = PW * 64, RMAX = 3;   // waves per workgroup, compute waves, threads, rows per wave per phase

// grid = one workgroup per CU (the 100 KB dynamic LDS request forces one per CU).  Wave PW-1 is the "sync" wave: it
// publishes the workgroup's outputs (write-through sc1 stores), drains them, runs the grid barrier and the acquire; the
// 15 compute waves meanwhile have the next phase's weight rows in flight and park on the workgroup barrier.
// MODE 0: barrier, prefetch before it (B); 1: barrier, prefetch after it (C); 3: flat sharded counters + sc1 reads (F)
template <int MODE>
__global__ __launch_bounds__(PT) void persistent_kernel(const float* __restrict__ Wall, float* xbuf0, float* xbuf1, BarState* s, int layers,
                                                        unsigned long long* gran0, unsigned long long* gran1) {
    constexpr bool PREFETCH = MODE != 1;
    constexpr bool GATHER = false;
    constexpr bool FLAT = MODE == 3;           // (F): sharded arrival counters polled by every CU, x read with sc1 loads - no fence, no barrier tree
    extern __shared__ float ylds[];            // [64] outputs of this workgroup in the current phase; [256 ..] the gathered x (MODE 2)
    float* xs = ylds + 256;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cu = blockIdx.x, ncu = gridDim.x;
    const unsigned xcc = xcc_id();
    __shared__ unsigned sh_n, sh_nx;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add((gu32*)&s->census[xcc * 32], 1u, RLX_AGENT);
        __hip_atomic_fetch_add((gu32*)&s->start[0], 1u, RLX_AGENT);
        spin_until_ge(&s->start[0], (unsigned)ncu, &s->error[0]);
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += __hip_atomic_load((gu32*)&s->census[x * 32], RLX_AGENT) > 0 ? 1u : 0u;
        sh_n = __hip_atomic_load((gu32*)&s->census[xcc * 32], RLX_AGENT);
        sh_nx = nx;
    }
    __syncthreads();
    const unsigned n_on_xcc = sh_n, n_xcc = sh_nx;

    long long layer_off = 0;
    for (int p = 0; p < NPH; ++p) layer_off += (long long)c_rows[p] * K;
    f32x4 w[RMAX][6];
    auto issue = [&](int gp) {                 // rows of global phase gp for this wave: local row lr = wid + CW*i -> row cu + ncu*lr
        const int l = gp / NPH, p = gp - l * NPH;
        const int N = c_rows[p];
        long long off = l * layer_off;
        for (int q = 0; q < p; ++q) off += (long long)c_rows[q] * K;
        const float* Wp = Wall + off;
        const int nloc = (N - cu + ncu - 1) / ncu;             // rows of this phase that live on this CU
#pragma unroll
        for (int i = 0; i < RMAX; ++i) {
            if (wid + CW * i < nloc) {                         // wave-uniform
                const f32x4* wr = reinterpret_cast<const f32x4*>(Wp + (long long)(cu + ncu * (wid + CW * i)) * K);
#pragma unroll
                for (int j = 0; j < 6; ++j) w[i][j] = __builtin_nontemporal_load(wr + j * 64 + lane);
            }
        }
    };
    const int total = layers * NPH;
    if (wid < CW) issue(0);
    if (GATHER || FLAT) {                      // phase 0 reads the host-written vector
        for (int i = threadIdx.x; i < K; i += PT) xs[i] = xbuf0[i];
        __syncthreads();
    }
    unsigned epoch = 0;
    for (int gp = 0; gp < total; ++gp) {
        const int p = gp % NPH;
        const int N = c_rows[p];
        const float* xin = (gp & 1) ? xbuf1 : xbuf0;
        float* yout = (gp & 1) ? xbuf0 : xbuf1;
        if (wid < CW) {
            if (!PREFETCH && gp > 0) issue(gp);
            f32x4 x[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) x[j] = (GATHER || FLAT) ? reinterpret_cast<const f32x4*>(xs)[j * 64 + lane] : reinterpret_cast<const f32x4*>(xin)[j * 64 + lane];
            const int nloc = (N - cu + ncu - 1) / ncu;
#pragma unroll
            for (int i = 0; i < RMAX; ++i) {
                const int lr = wid + CW * i;
                if (lr < nloc) {
                    const float v = dot_row(w[i], x);
                    if (lane == 0) ylds[lr] = squash(v);
                }
            }
            if (PREFETCH && gp + 1 < total) issue(gp + 1);     // in flight across the grid barrier
        }
        __syncthreads();                                        // ylds complete
        if (wid == CW && FLAT) {
            // (F) publish write-through, drain, ONE arrival on shard cu % 8, poll the 8 shards, fetch x past L1 (sc1) into LDS
            if (lane < 48) {
                const int row = cu + ncu * lane;
                if (row < N) __hip_atomic_store((gf32*)(yout + row), ylds[lane], RLX_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (gp + 1 < total) {
                if (lane == 0) __hip_atomic_fetch_add((gu32*)&s->shard[(cu & 7) * 32], 1u, RLX_AGENT);
                const unsigned want = (unsigned)(gp + 1) * (unsigned)((ncu + 7 - (lane & 7)) / 8);   // arrivals of shard (lane & 7) so far
                unsigned spins = 0;
                for (;;) {
                    const unsigned c = __hip_atomic_load((gu32*)&s->shard[(lane & 7) * 32], RLX_AGENT);
                    if (__all(c >= want)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT / 16 || __hip_atomic_load((gu32*)&s->error[0], RLX_AGENT) != 0) {
                        __hip_atomic_store((gu32*)&s->error[0], 1u, RLX_AGENT);
                        break;
                    }
                }
                f32x4 t[6];
                const float* p0 = yout + lane * 4;
                const float* p1 = p0 + 3 * 256;
                asm volatile(
                    "global_load_dwordx4 %0, %6, off sc1\n\t"
                    "global_load_dwordx4 %1, %6, off offset:1024 sc1\n\t"
                    "global_load_dwordx4 %2, %6, off offset:2048 sc1\n\t"
                    "global_load_dwordx4 %3, %7, off sc1\n\t"
                    "global_load_dwordx4 %4, %7, off offset:1024 sc1\n\t"
                    "global_load_dwordx4 %5, %7, off offset:2048 sc1\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5])
                    : "v"(p0), "v"(p1)
                    : "memory");
#pragma unroll
                for (int j = 0; j < 6; ++j) reinterpret_cast<f32x4*>(xs)[j * 64 + lane] = t[j];
            }
        } else if (wid == CW) {
            // publish this workgroup's rows (write-through), drain, grid barrier, acquire
            if (lane < 48) {
                const int row = cu + ncu * lane;
                if (row < N) __hip_atomic_store((gf32*)(yout + row), ylds[lane], RLX_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && gp + 1 < total) {
                ++epoch;
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
            } else if (gp + 1 < total) {
                ++epoch;
            }
        }
        __syncthreads();                                        // next phase's x is visible to this CU
    }
}

static void fill(std::vector<float>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 8388608.0f - 1.0f) * scale; }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    long long per_layer = 0;
    for (int p = 0; p < NPH; ++p) per_layer += (long long)h_rows[p] * K;
    printf("layer stand-in: %.1f MB of weights per layer, %d layers (%.2f GB)\n", per_layer * 4 / 1e6, NL, per_layer * 4.0 * NL / 1e9);
    float* W;
    CHECK(hipMalloc(&W, per_layer * NL * sizeof(float)));
    {   // fill on the host in chunks (values ~ U(-0.03, 0.03))
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
    BarState* bs;
    CHECK(hipMalloc(&bs, sizeof(BarState)));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;

    // ---- (A) graph of 24 x 5 kernels
    auto enqueue_layers = [&](hipStream_t s_) {
        int gp = 0;
        for (int l = 0; l < NL; ++l) {
            long long off = 0;
            for (int p = 0; p < NPH; ++p, ++gp) {
                const float* Wp = W + l * per_layer + off;
                const float* xin = (gp & 1) ? xb : xa;
                float* yout = (gp & 1) ? xa : xb;
                const int N = h_rows[p];
                if (N >= 6144) hipLaunchKernelGGL(phase_kernel<2>, dim3((N + 7) / 8), dim3(256), 0, s_, Wp, xin, yout, N);
                else hipLaunchKernelGGL(phase_kernel<1>, dim3((N + 3) / 4), dim3(256), 0, s_, Wp, xin, yout, N);
                off += (long long)N * K;
            }
        }
    };
    hipGraph_t graph;
    hipGraphExec_t gexec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    enqueue_layers(st);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    std::vector<float> refA(XLEN), got(XLEN);
    auto reset_x = [&]() {
        CHECK(hipMemcpy(xa, x0.data(), XLEN * sizeof(float), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(xb, x0.data(), XLEN * sizeof(float), hipMemcpyHostToDevice));
    };
    reset_x();
    CHECK(hipGraphLaunch(gexec, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(refA.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));   // 120 phases: last output in xa
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(gexec, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float msA;
    CHECK(hipEventElapsedTime(&msA, e0, e1));
    const double usA = msA * 1000.0 / reps / NL;
    printf("(A) 5 graph-replayed kernels per layer : %7.2f us per layer  (%.2f TB/s)\n", usA, per_layer * 4 / usA / 1e6);

    // ---- (B)/(C) persistent
    const size_t lds = 100 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&persistent_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&persistent_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&persistent_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned long long *g0 = nullptr, *g1 = nullptr;
    const int vmask = argc > 1 ? atoi(argv[1]) : 11;      // bit v selects variant v: 1 = (B), 2 = (C), 8 = (F)  [4 = (D), removed]
    for (int variant = 0; variant < 4; ++variant) {
        if (!(vmask & (1 << variant)) || variant == 2) continue;
        const bool prefetch = variant != 1;
        float best = 1e9f;
        bool ok = true, timeout = false;
        for (int r = 0; r < 6; ++r) {
            reset_x();
            CHECK(hipMemsetAsync(bs, 0, sizeof(BarState), st));
            CHECK(hipEventRecord(e0, st));
            if (variant == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(ncu), dim3(PT), lds, st, W, xa, xb, bs, NL, g0, g1);
            else if (variant == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(ncu), dim3(PT), lds, st, W, xa, xb, bs, NL, g0, g1);
            else if (variant == 2) continue;
            else hipLaunchKernelGGL(persistent_kernel<3>, dim3(ncu), dim3(PT), lds, st, W, xa, xb, bs, NL, g0, g1);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
            BarState h;
            CHECK(hipMemcpy(&h, bs, sizeof(BarState), hipMemcpyDeviceToHost));
            timeout |= h.error[0] != 0;
            CHECK(hipMemcpy(got.data(), xa, XLEN * sizeof(float), hipMemcpyDeviceToHost));
            ok &= memcmp(got.data(), refA.data(), 6144 * sizeof(float)) == 0;
            if (timeout) break;
        }
        const double us = best * 1000.0 / NL;
        printf("(%c) persistent, %s, prefetch %-6s: %7.2f us per layer  (%.2f TB/s)  result %s%s\n", "BCDF"[variant],
               variant == 2 ? "tagged all-gather" : (variant == 3 ? "flat sharded counters + sc1 reads" : "xcd barrier"), prefetch ? "before" : "after", us, per_layer * 4 / us / 1e6,
               ok ? "bit-identical to (A)" : "DIFFERS from (A)", timeout ? "  [SPIN TIMEOUT]" : "");
    }
    return 0;
}
