// Round 6: IN-KERNEL TAIL PREFETCH next to the launch chain.  The round-5 walker (prefetch_stream_probe.hip) ran on
// a second stream and lost in every variant; here the prefetch is issued by the chain kernels themselves: behind its own weight loads
// every wave touches one dword of each 128-byte line of the first PF KB of the weight blocks that workgroups of the NEXT launch on
// the SAME XCD will read (workgroup j of a launch runs on XCD j % 8 and every grid is a multiple of 8: workgroup b covers the next
// launch's workgroups b, b + G, ...), so that those lines are in that XCD's L2 when the next launch starts.  Built on
// launch_chain_floor_probe.hip:
// the floor of a launch chain at the byte volumes of the B = 1 decode layer: five graph-replayed plain GEMV kernels per layer
// (no LayerNorm, no softmax, no merge - just "all loads in flight, dot, store"), at the fp32 volumes (J = 6 sixteen-byte loads
// per lane and row: 163 MB per layer) and at the fp16 volumes (J = 3: 81.5 MB per layer), with the workgroup shapes the real
// kernels use (768 workgroups of 4 waves x 2 rows) and with one fat workgroup per CU.  What the real layer loses against these
// numbers is arithmetic structure (prologues, reductions, merges), what these numbers lose against bytes / 6.3 TB/s is the
// launch chain itself.
//   hipcc --offload-arch=gfx950 -O3 -o tail_prefetch_probe tail_prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NPH = 5, NL = 24;
static const int h_rows[NPH] = {4608, 8192, 1536, 6144, 6144};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// J 16-byte loads per lane and row (row = J KiB); NW waves x RW rows per workgroup
struct Next { const float* W; int nwg; int wg_floats; int pf_lines; };     // next launch: weights, workgroups, floats per workgroup block, lines to touch per block

template <int J, int NW, int RW>
__global__ __launch_bounds__(64 * NW) void plain_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N, Next nx) {
    constexpr int KF = J * 256;                 // floats per row
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * NW + wid) * RW;
    f32x4 w[RW][J];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + (long long)min(row0 + r, N - 1) * KF);
#pragma unroll
        for (int j = 0; j < J; ++j) w[r][j] = __builtin_nontemporal_load(wr + j * 64 + lane);
    }
    f32x4 x[J];
#pragma unroll
    for (int j = 0; j < J; ++j) x[j] = reinterpret_cast<const f32x4*>(xin)[j * 64 + lane];
    // tail prefetch: queued BEHIND this wave's own loads (loads return in order); line l of the next workgroup's block by thread l
    float pf = 0.f;
    if (nx.pf_lines > 0) {
        for (int jn = blockIdx.x; jn < nx.nwg; jn += gridDim.x)
            for (int l = threadIdx.x; l < nx.pf_lines; l += 64 * NW) {
                float t;
                asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(nx.W + (long long)jn * nx.wg_floats + l * 32) : "memory");
                pf = t;
            }
    }
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
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(pf) : "memory");       // (the wave ends with nothing in flight)
}

static int wg_rows(int shape, int N) { return shape == 0 ? (N >= 6144 ? 8 : 4) : (N == 4608 ? 18 : N == 8192 ? 32 : N == 1536 ? 6 : 24); }

template <int J>
static double run(int shape, const float* W, float* xa, float* xb, hipStream_t st, double* per_kernel, int pf_kb) {
    const long long KF = J * 256;
    auto enqueue_one = [&](int p, const float* Wp, const float* xin, float* yout, const float* Wnext, int pn) {
        const int N = h_rows[p];
        Next nx{Wnext, 0, 0, 0};
        if (Wnext && pf_kb > 0) {
            const int rows = wg_rows(shape, h_rows[pn]);
            nx.nwg = (h_rows[pn] + rows - 1) / rows;
            nx.wg_floats = rows * (int)KF;
            nx.pf_lines = (int)std::min<long long>(pf_kb * 8, (long long)rows * KF / 32);
        }
        if (shape == 0) {          // the real kernels' shapes: 4 waves, 2 rows per wave for the wide matrices, 1 otherwise
            if (N >= 6144) hipLaunchKernelGGL((plain_kernel<J, 4, 2>), dim3((N + 7) / 8), dim3(256), 0, st, Wp, xin, yout, N, nx);
            else hipLaunchKernelGGL((plain_kernel<J, 4, 1>), dim3((N + 3) / 4), dim3(256), 0, st, Wp, xin, yout, N, nx);
        } else {                   // one workgroup per CU
            if (N == 4608) hipLaunchKernelGGL((plain_kernel<J, 9, 2>), dim3(256), dim3(576), 0, st, Wp, xin, yout, N, nx);
            else if (N == 8192) hipLaunchKernelGGL((plain_kernel<J, 16, 2>), dim3(256), dim3(1024), 0, st, Wp, xin, yout, N, nx);
            else if (N == 1536) hipLaunchKernelGGL((plain_kernel<J, 6, 1>), dim3(256), dim3(384), 0, st, Wp, xin, yout, N, nx);
            else hipLaunchKernelGGL((plain_kernel<J, 12, 2>), dim3(256), dim3(768), 0, st, Wp, xin, yout, N, nx);
        }
    };
    long long per_layer = 0;
    for (int p = 0; p < NPH; ++p) per_layer += (long long)h_rows[p] * KF;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    auto time_graph = [&](int only) {       // only < 0: the whole layer chain; else 24 launches of phase `only`
        hipGraph_t graph;
        hipGraphExec_t gexec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int gp = 0;
        for (int l = 0; l < NL; ++l) {
            long long off = 0;
            for (int p = 0; p < NPH; ++p, ++gp) {
                const long long noff = off + (long long)h_rows[p] * KF;
                const bool last = p == NPH - 1;
                const float* Wn = only >= 0 ? nullptr : last ? (l + 1 < NL ? W + (l + 1) * per_layer : nullptr) : W + l * per_layer + noff;
                if (only < 0 || only == p) enqueue_one(p, W + l * per_layer + off, (gp & 1) ? xb : xa, (gp & 1) ? xa : xb, Wn, last ? 0 : p + 1);
                off = noff;
            }
        }
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(gexec, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(gexec, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipGraphExecDestroy(gexec));
        CHECK(hipGraphDestroy(graph));
        return ms * 1000.0 / reps / NL;
    };
    const double us = time_graph(-1);
    for (int p = 0; p < NPH; ++p) per_kernel[p] = time_graph(p);
    return us;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
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
    CHECK(hipMalloc(&xa, 8192 * sizeof(float)));
    CHECK(hipMalloc(&xb, 8192 * sizeof(float)));
    CHECK(hipMemset(xa, 0, 8192 * sizeof(float)));
    CHECK(hipMemset(xb, 0, 8192 * sizeof(float)));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int pfs[] = {0, 4, 8, 16, 32, 0, 16};
    for (int shape = 0; shape < 2; ++shape)
        for (int pf : pfs) {
            double pk[NPH];
            const double u32 = run<6>(shape, W, xa, xb, st, pk, pf);
            const double u16 = run<3>(shape, W, xa, xb, st, pk, pf);
            printf("%s  tail prefetch %2d KB per next workgroup:  fp32 volumes %6.2f us per layer (%.2f TB/s)   fp16 volumes %6.2f us per layer (%.2f TB/s)\n",
                   shape ? "256 fat workgroups  " : "768-workgroup shapes", pf, u32, 163.6 / u32, u16, 81.8 / u16);
            fflush(stdout);
        }
    return 0;
}
