// Does a run-ahead PREFETCH WALKER on a second stream shorten the B = 1 decode launch chain?
//
// The decode layer is a chain of five weight-streaming launches whose only run-ahead-able inputs are the weights and the KV
// cache (addresses known long before the activations are).  profiles/r03_launch_chain_floor_probe.log: the chain of plain
// streaming kernels runs at 4.8 TB/s (fp32 volumes) / 3.6 TB/s (fp16 volumes) - HBM idles through every launch boundary,
// ramp and tail.  The persistent engine (one launch, loader wave per CU) measured 0.98x: its all-to-all edges cost what
// the boundaries cost.  This probe keeps the cheap boundaries and adds the run-ahead from OUTSIDE the chain: a one-wave-per-CU
// walker kernel on a second stream touches one dword of every 128-byte line of the data of launch k + 1 .. k + LEAD while
// launch k runs, paced by a progress word that workgroup 0 of every chain kernel bumps at entry (one no-return atomic).
// Lines land in the L2 of the XCD that will read them (block b of a chain kernel runs on XCD b % 8; walker workgroup j runs
// on XCD j % 8 and takes the blocks of its own XCD) and in the Infinity Cache.  Variants: affine (own XCD), mis-affine (XCD + 4:
// Infinity Cache only), LEAD 1 / 2 / 3, one or two walker waves per CU.  Spins are bounded; a walker that times out exits.
//   hipcc --offload-arch=gfx950 -O3 -o prefetch_stream_probe prefetch_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NPH = 5, NL = 24, NK = NPH * NL;
static const int h_rows[NPH] = {4608, 8192, 1536, 6144, 6144};
static const char* h_name[NPH] = {"qkv", "kv", "out", "fc1", "fc2"};

struct Region { const char* base; int block_bytes; int nblocks; int pad; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

// J 16-byte loads per lane and row (row = J KiB); NW waves x RW rows per workgroup: block b reads rows [b NW RW, (b + 1) NW RW)
template <int J, int NW, int RW, bool NT = true>
__global__ __launch_bounds__(64 * NW) void plain_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N,
                                                       unsigned* prog, unsigned* census) {
    if (prog && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (census && threadIdx.x == 0) atomicAdd(census + (blockIdx.x & 7) * 8 + xcc_id(), 1u);      // diagnostic pass only
    constexpr int KF = J * 256;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * NW + wid) * RW;
    f32x4 w[RW][J];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + (long long)min(row0 + r, N - 1) * KF);
#pragma unroll
        for (int j = 0; j < J; ++j) w[r][j] = NT ? __builtin_nontemporal_load(wr + j * 64 + lane) : wr[j * 64 + lane];
    }
    f32x4 x[J];
#pragma unroll
    for (int j = 0; j < J; ++j) x[j] = reinterpret_cast<const f32x4*>(xin)[j * 64 + lane];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            s = fmaf(w[r][j].x, x[j].x, s); s = fmaf(w[r][j].y, x[j].y, s); s = fmaf(w[r][j].z, x[j].z, s); s = fmaf(w[r][j].w, x[j].w, s);
        }
        s = wave_sum(s);
        if (lane == 0 && row0 + r < N) yout[row0 + r] = s / (1.0f + fabsf(s));
    }
}

constexpr int SPIN_MAX = 4000;     // x (s_sleep + one poll round trip) ~ 5 ms: a walker whose chain never comes gives up

// One wave per workgroup.  stats[0] regions prefetched (summed over workgroups), [1] skipped (chain already there), [2] give-ups,
// [3] workgroups not on XCD j % 8, [4] polls while waiting
// res_of_xcc: 8 x 4 bits, nibble x = the residue (block % 8) of the chain blocks that run on XCC x (measured by the census pass)
__global__ __launch_bounds__(64) void walker_kernel(const Region* __restrict__ tab, int nk, unsigned seq0, int lead, unsigned* prog, int xshift,
                                                    unsigned* stats, unsigned res_of_xcc, int keep_mod) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const unsigned my = xcc_id();
    const int xcd = (int)((res_of_xcc >> (4 * ((my + xshift) & 7))) & 7), slot = j >> 3, nslots = gridDim.x >> 3;     // (walker workgroups j, j + 8, ... share an XCC: checked below)
    if (lane == 0) atomicAdd(stats + 16 + (j & 7) * 8 + my, 1u);           // walker census: [j % 8][XCC]
    unsigned p = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned done = 0, skipped = 0, polls = 0;
    bool gave_up = false;
    for (int m = 0; m < nk; ++m) {
        const unsigned seq = seq0 + m;                 // chain kernel `seq` has started once prog > seq
        int it = 0;
        while (p <= seq && p + lead <= seq) {          // not started and more than `lead` launches ahead of the chain: wait
            __builtin_amdgcn_s_sleep(8);
            p = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++polls;
            if (++it > SPIN_MAX) { gave_up = true; break; }
        }
        if (gave_up) break;
        if (p > seq) { ++skipped; continue; }
        const Region r = tab[m];
        for (int b = xcd + 8 * slot; b < r.nblocks; b += 8 * nslots) {
            if (keep_mod > 1 && ((b >> 3) / nslots) % keep_mod != 0) continue;
            const char* q = r.base + (long long)b * r.block_bytes + lane * 128;
            for (int off = lane * 128; off < r.block_bytes; off += 64 * 128, q += 64 * 128) {
                unsigned junk;
                asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(q) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++done;
    }
    if (lane == 0) {
        atomicAdd(stats + 0, done);
        atomicAdd(stats + 1, skipped);
        if (gave_up) atomicAdd(stats + 2, 1u);
        atomicAdd(stats + 4, polls);
    }
}

template <int J>
struct Chain {
    const float* W; float *xa, *xb; unsigned* prog; unsigned* census; long long per_layer;
    int fat_kv;
    void enqueue(int p, const float* Wp, const float* xin, float* yout, hipStream_t st, bool armed, bool with_census = false) const {
        const int N = h_rows[p];
        unsigned* pg = armed ? prog : nullptr;
        unsigned* census = with_census ? this->census + p * 64 : nullptr;
        if (N == 8192 && fat_kv) hipLaunchKernelGGL((plain_kernel<J, 16, 2>), dim3(256), dim3(1024), 0, st, Wp, xin, yout, N, pg, census);
        else if (N >= 6144) hipLaunchKernelGGL((plain_kernel<J, 4, 2>), dim3((N + 7) / 8), dim3(256), 0, st, Wp, xin, yout, N, pg, census);
        else hipLaunchKernelGGL((plain_kernel<J, 4, 1>), dim3((N + 3) / 4), dim3(256), 0, st, Wp, xin, yout, N, pg, census);
    }
    int block_rows(int p) const {
        const int N = h_rows[p];
        if (N == 8192 && fat_kv) return 32;
        return N >= 6144 ? 8 : 4;
    }
};

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; %d graph replays of %d launches per timing\n", prop.name, prop.multiProcessorCount, reps, NK);
    long long per_layer = 0;
    for (int p = 0; p < NPH; ++p) per_layer += (long long)h_rows[p] * 1536;
    float* W;
    CHECK(hipMalloc(&W, per_layer * NL * sizeof(float)));
    {
        std::vector<float> h(per_layer);
        unsigned s = 1234u;
        for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.03f; }
        for (int l = 0; l < NL; ++l) CHECK(hipMemcpy(W + l * per_layer, h.data(), per_layer * sizeof(float), hipMemcpyHostToDevice));
    }
    float *xa, *xb;
    unsigned *prog, *stats, *census;
    Region* tab;
    CHECK(hipMalloc(&xa, 8192 * sizeof(float)));
    CHECK(hipMalloc(&xb, 8192 * sizeof(float)));
    CHECK(hipMemset(xa, 0, 8192 * sizeof(float)));
    CHECK(hipMemset(xb, 0, 8192 * sizeof(float)));
    CHECK(hipMalloc(&prog, 256));
    CHECK(hipMalloc(&stats, 512));
    CHECK(hipMalloc(&census, NPH * 64 * 4));
    CHECK(hipMalloc(&tab, NK * sizeof(Region)));
    hipStream_t st, st2;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    unsigned res_of_xcc = 0;
    auto run_all = [&](auto chain, int J, const char* label) {
        const long long KF = J * 256;
        long long pl = 0;
        for (int p = 0; p < NPH; ++p) pl += (long long)h_rows[p] * KF;
        // region table of the chain, in launch order
        std::vector<Region> h_tab(NK);
        {
            int gp = 0;
            for (int l = 0; l < NL; ++l) {
                long long off = 0;
                for (int p = 0; p < NPH; ++p, ++gp) {
                    const int br = chain.block_rows(p);
                    h_tab[gp].base = reinterpret_cast<const char*>(W + l * pl + off);
                    h_tab[gp].block_bytes = br * (int)KF * 4;
                    h_tab[gp].nblocks = (h_rows[p] + br - 1) / br;
                    h_tab[gp].pad = 0;
                    off += (long long)h_rows[p] * KF;
                }
            }
        }
        CHECK(hipMemcpy(tab, h_tab.data(), NK * sizeof(Region), hipMemcpyHostToDevice));
        hipGraph_t graph;
        hipGraphExec_t gexec;
        auto capture = [&](bool armed) {
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            int gp = 0;
            for (int l = 0; l < NL; ++l) {
                long long off = 0;
                for (int p = 0; p < NPH; ++p, ++gp) {
                    chain.enqueue(p, W + l * pl + off, (gp & 1) ? xb : xa, (gp & 1) ? xa : xb, st, armed);
                    off += (long long)h_rows[p] * KF;
                }
            }
            CHECK(hipStreamEndCapture(st, &graph));
            CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        };
        auto time_it = [&](int lead, int xshift, int nwg, int keep_mod = 1) {       // lead == 0: no walker
            CHECK(hipMemset(prog, 0, 256));
            CHECK(hipMemset(stats, 0, 512));
            CHECK(hipDeviceSynchronize());
            // warm-up replay (with its walker) + timed replays
            for (int r = 0; r <= reps; ++r) {
                if (r == 1) CHECK(hipEventRecord(e0, st));
                if (lead > 0) hipLaunchKernelGGL(walker_kernel, dim3(nwg), dim3(64), 0, st2, tab, NK, (unsigned)(r * NK), lead, prog, xshift, stats, res_of_xcc, keep_mod);
                CHECK(hipGraphLaunch(gexec, st));
            }
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipStreamSynchronize(st2));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hs[80];
            CHECK(hipMemcpy(hs, stats, sizeof(hs), hipMemcpyDeviceToHost));
            hs[3] = 0;                                   // walker workgroups j, j + 8, ... not all on one XCC
            for (int r8 = 0; r8 < 8; ++r8) { unsigned mx = 0, tt = 0; for (int x = 0; x < 8; ++x) { tt += hs[16 + r8 * 8 + x]; if (hs[16 + r8 * 8 + x] > mx) mx = hs[16 + r8 * 8 + x]; } hs[3] += tt - mx; }
            const double us = ms * 1000.0 / reps / NL;
            if (lead == 0) printf("  %-44s: %6.2f us per layer (%.2f TB/s)\n", "chain alone (progress word armed)", us, pl * 4.0 / 1e6 / us);
            else {
                const double tot = (double)nwg * NK * (reps + 1);
                printf("  walker lead %d, XCC shift %d%s, 1/%d of the blocks, %3d waves: %6.2f us per layer (%.2f TB/s)  regions prefetched %.0f%% skipped %.0f%% give-ups %u walkers off their XCC group %u polls/region %.1f\n",
                       lead, xshift, xshift ? " (wrong L2)" : " (consumer's L2)", keep_mod, nwg, us, pl * 4.0 / 1e6 / us, 100.0 * hs[0] / tot,
                       100.0 * hs[1] / tot, hs[2], hs[3], hs[4] / tot);
            }
            return us;
        };
        printf("%s: %.1f MB per layer\n", label, pl * 4.0 / 1e6);
        // unarmed baseline
        capture(false);
        {
            CHECK(hipGraphLaunch(gexec, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(gexec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %-44s: %6.2f us per layer (%.2f TB/s)\n", "chain alone (no progress word)", ms * 1000.0 / reps / NL, pl * 4.0 / 1e6 / (ms * 1000.0 / reps / NL));
        }
        CHECK(hipGraphExecDestroy(gexec));
        CHECK(hipGraphDestroy(graph));
        capture(true);
        time_it(0, 0, 0);
        time_it(1, 0, 256);
        time_it(1, 0, 256, 2);
        time_it(1, 0, 256, 4);
        time_it(2, 0, 256);
        time_it(2, 0, 256, 2);
        for (int sh = 1; sh < 8; ++sh) time_it(1, sh, 256, 2);
        time_it(1, 0, 512, 2);
        time_it(0, 0, 0);
        CHECK(hipGraphExecDestroy(gexec));
        CHECK(hipGraphDestroy(graph));
    };

    // where do the chain's workgroups run?  one eager pass of a layer with the census on: rows = block % 8, columns = XCC id
    {
        Chain<6> c6{W, xa, xb, prog, census, per_layer, 1};
        CHECK(hipMemset(census, 0, NPH * 64 * 4));
        long long off = 0;
        for (int p = 0; p < NPH; ++p) { c6.enqueue(p, W + off, xa, xb, st, false, true); off += (long long)h_rows[p] * 1536; }
        CHECK(hipStreamSynchronize(st));
        std::vector<unsigned> hc(NPH * 64);
        CHECK(hipMemcpy(hc.data(), census, NPH * 64 * 4, hipMemcpyDeviceToHost));
        for (int p = 0; p < NPH; ++p) {
            printf("census %-3s: block %% 8 -> XCC:", h_name[p]);
            for (int r = 0; r < 8; ++r) {
                int best = 0;
                unsigned tot = 0;
                for (int x = 0; x < 8; ++x) { tot += hc[p * 64 + r * 8 + x]; if (hc[p * 64 + r * 8 + x] > hc[p * 64 + r * 8 + best]) best = x; }
                printf(" %d->%d (%u/%u)", r, best, hc[p * 64 + r * 8 + best], tot);
                if (p == 0) res_of_xcc |= (unsigned)r << (4 * best);
            }
            printf("\n");
        }
        printf("residue table (nibble x = block %% 8 of the chain blocks on XCC x): 0x%08x\n", res_of_xcc);
    }
    for (int fat = 1; fat < 2; ++fat) {
        Chain<6> c6{W, xa, xb, prog, census, per_layer, fat};
        run_all(c6, 6, fat ? "fp32 volumes, kv as 256 x 16-wave workgroups" : "fp32 volumes, 768 / 1024-workgroup shapes");
        Chain<3> c3{W, xa, xb, prog, census, per_layer, fat};
        run_all(c3, 3, fat ? "fp16 volumes, kv as 256 x 16-wave workgroups" : "fp16 volumes, 768 / 1024-workgroup shapes");
    }
    return 0;
}
