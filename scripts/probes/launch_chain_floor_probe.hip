// The floor of a launch chain at the byte volumes of the B = 1 decode layer: five graph-replayed plain GEMV kernels per layer
// (no LayerNorm, no softmax, no merge - just "all loads in flight, dot, store"), at the fp32 volumes (J = 6 sixteen-byte loads
// per lane and row: 163 MB per layer) and at the fp16 volumes (J = 3: 81.5 MB per layer), with the workgroup shapes the real
// kernels use (768 workgroups of 4 waves x 2 rows) and with one fat workgroup per CU.  What the real layer loses against these
// numbers is arithmetic structure (prologues, reductions, merges), what these numbers lose against bytes / 6.3 TB/s is the
// launch chain itself.
//   hipcc --offload-arch=gfx950 -O3 -o launch_chain_floor_probe launch_chain_floor_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
template <int J, int NW, int RW>
__global__ __launch_bounds__(64 * NW) void plain_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N) {
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

template <int J>
static double run(int shape, const float* W, float* xa, float* xb, hipStream_t st, double* per_kernel) {
    const long long KF = J * 256;
    auto enqueue_one = [&](int p, const float* Wp, const float* xin, float* yout) {
        const int N = h_rows[p];
        if (shape == 0) {          // the real kernels' shapes: 4 waves, 2 rows per wave for the wide matrices, 1 otherwise
            if (N >= 6144) hipLaunchKernelGGL((plain_kernel<J, 4, 2>), dim3((N + 7) / 8), dim3(256), 0, st, Wp, xin, yout, N);
            else hipLaunchKernelGGL((plain_kernel<J, 4, 1>), dim3((N + 3) / 4), dim3(256), 0, st, Wp, xin, yout, N);
        } else {                   // one workgroup per CU
            if (N == 4608) hipLaunchKernelGGL((plain_kernel<J, 9, 2>), dim3(256), dim3(576), 0, st, Wp, xin, yout, N);
            else if (N == 8192) hipLaunchKernelGGL((plain_kernel<J, 16, 2>), dim3(256), dim3(1024), 0, st, Wp, xin, yout, N);
            else if (N == 1536) hipLaunchKernelGGL((plain_kernel<J, 6, 1>), dim3(256), dim3(384), 0, st, Wp, xin, yout, N);
            else hipLaunchKernelGGL((plain_kernel<J, 12, 2>), dim3(256), dim3(768), 0, st, Wp, xin, yout, N);
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
                if (only < 0 || only == p) enqueue_one(p, W + l * per_layer + off, (gp & 1) ? xb : xa, (gp & 1) ? xa : xb);
                off += (long long)h_rows[p] * KF;
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
    for (int shape = 0; shape < 2; ++shape) {
        double pk[NPH];
        double us = run<6>(shape, W, xa, xb, st, pk);
        printf("fp32 volumes (163.6 MB / layer), %s: %6.2f us per layer (%.2f TB/s); alone: qkv %.2f  kv %.2f  out %.2f  fc1 %.2f  fc2 %.2f\n",
               shape ? "256 fat workgroups   " : "768-workgroup shapes ", us, 163.6 / us, pk[0], pk[1], pk[2], pk[3], pk[4]);
        us = run<3>(shape, W, xa, xb, st, pk);
        printf("fp16 volumes ( 81.8 MB / layer), %s: %6.2f us per layer (%.2f TB/s); alone: qkv %.2f  kv %.2f  out %.2f  fc1 %.2f  fc2 %.2f\n",
               shape ? "256 fat workgroups   " : "768-workgroup shapes ", us, 81.8 / us, pk[0], pk[1], pk[2], pk[3], pk[4]);
    }
    return 0;
}
