// Do two branches of one hipGraph run CONCURRENTLY on MI355X / ROCm 7.2, and what does a fork + join cost?
//
// The decode layer's k / v projection (18.9 MB of fp32 weights) does not feed the attention over the OLD keys: only the new key does,
// and that one could enter the merge as a 17th partial.  If a graph ran [q projection -> attention] and [k / v projection] as parallel
// branches, the layer's critical path would lose ~3 us (of 37.9).  This probe measures exactly that shape with the plain streaming
// kernels of launch_chain_floor_probe.hip at the fp32 volumes:
//   serial : q (9.4 MB) -> kv (18.9 MB) -> attention (50.3 MB) -> out (9.4) -> fc1 (37.7) -> fc2 (37.7)
//   forked : [q -> attention] || [kv]  -> join -> out -> fc1 -> fc2        (stream capture with two streams and events)
// and the same with the kv branch EMPTY (fork + join overhead alone).
//   hipcc --offload-arch=gfx950 -O3 -o graph_fork_probe graph_fork_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NL = 24;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int J, int NW, int RW>
__global__ __launch_bounds__(64 * NW) void plain_kernel(const float* __restrict__ W, const float* __restrict__ xin, float* __restrict__ yout, int N) {
    constexpr int KF = J * 256;
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
        for (int j = 0; j < J; ++j) { s = fmaf(w[r][j].x, x[j].x, s); s = fmaf(w[r][j].y, x[j].y, s); s = fmaf(w[r][j].z, x[j].z, s); s = fmaf(w[r][j].w, x[j].w, s); }
        s = wave_sum(s);
        if (lane == 0 && row0 + r < N) yout[row0 + r] = s / (1.0f + fabsf(s));
    }
}

int main() {
    // rows of 1536 floats: q 1536, kv 3072, attention stand-in 8192, out 1536, fc1 6144, fc2 6144
    const int rows[6] = {1536, 3072, 8192, 1536, 6144, 6144};
    long long per_layer = 0;
    for (int r : rows) per_layer += (long long)r * 1536;
    float* W;
    CHECK(hipMalloc(&W, per_layer * NL * sizeof(float)));
    CHECK(hipMemset(W, 0, per_layer * NL * sizeof(float)));
    float* x[8];
    for (auto& p : x) { CHECK(hipMalloc(&p, 8192 * 4)); CHECK(hipMemset(p, 0, 8192 * 4)); }
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    auto launch = [&](int p, const float* Wp, const float* xin, float* yout, hipStream_t st) {
        const int N = rows[p];
        if (N == 8192) hipLaunchKernelGGL((plain_kernel<6, 16, 2>), dim3(256), dim3(1024), 0, st, Wp, xin, yout, N);
        else if (N >= 6144) hipLaunchKernelGGL((plain_kernel<6, 4, 2>), dim3((N + 7) / 8), dim3(256), 0, st, Wp, xin, yout, N);
        else hipLaunchKernelGGL((plain_kernel<6, 4, 1>), dim3((N + 3) / 4), dim3(256), 0, st, Wp, xin, yout, N);
    };
    // mode 0: serial; 1: forked; 2: forked with an EMPTY side branch (kv stays in the main chain: fork/join overhead only); 3: serial without kv (lower bound)
    auto time_mode = [&](int mode) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeRelaxed));
        for (int l = 0; l < NL; ++l) {
            const float* Wl = W + l * per_layer;
            long long off[6], o = 0;
            for (int p = 0; p < 6; ++p) { off[p] = o; o += (long long)rows[p] * 1536; }
            if (mode == 0 || mode == 3) {
                launch(0, Wl + off[0], x[0], x[1], s1);
                if (mode == 0) launch(1, Wl + off[1], x[0], x[2], s1);
                launch(2, Wl + off[2], x[1], x[3], s1);
            } else {
                CHECK(hipEventRecord(ef, s1));
                CHECK(hipStreamWaitEvent(s2, ef, 0));
                launch(0, Wl + off[0], x[0], x[1], s1);
                if (mode == 2) launch(1, Wl + off[1], x[0], x[2], s1);
                launch(2, Wl + off[2], x[1], x[3], s1);
                if (mode == 1) launch(1, Wl + off[1], x[0], x[2], s2);
                else hipLaunchKernelGGL((plain_kernel<6, 4, 1>), dim3(1), dim3(256), 0, s2, Wl + off[1], x[0], x[7], 4);      // a token kernel on the side branch
                CHECK(hipEventRecord(ej, s2));
                CHECK(hipStreamWaitEvent(s1, ej, 0));
            }
            launch(3, Wl + off[3], x[3], x[4], s1);
            launch(4, Wl + off[4], x[4], x[5], s1);
            launch(5, Wl + off[5], x[5], x[0], s1);
        }
        CHECK(hipStreamEndCapture(s1, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge, s1));
        CHECK(hipStreamSynchronize(s1));
        const int reps = 20;
        CHECK(hipEventRecord(e0, s1));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, s1));
        CHECK(hipEventRecord(e1, s1));
        CHECK(hipStreamSynchronize(s1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
        const char* names[4] = {"serial chain of six launches", "forked: [q -> attention] || [kv], join", "forked with a token side branch (fork + join cost)", "serial WITHOUT the kv launch (lower bound)"};
        printf("  %-56s %7.2f us per layer\n", names[mode], ms * 1000.0 / reps / NL);
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int mode : {0, 1, 2, 3}) time_mode(mode);
    return 0;
}
