// Round 6: how fast can ONE CU pull operand tiles into LDS?  The streamed GEMM (k_gemm_stream.h) measured 34 GB/s per CU of LDS-DMA with
// four loader waves - the bound of its k-step.  This probe prices the delivery path by itself: W loader waves per workgroup (one
// workgroup per CU, 160 KB of LDS), each moving 1 KB pieces (64 lanes x 16 B, eight 128-byte rows) of a tile stream into a ring, with
// two k-tiles' worth of pieces in flight per wave, in four forms:
//   0  global_load_lds_dwordx4, 64-bit per-lane address, M0 saved / set / restored around every piece (gm_glds16 of k_gemm.h)
//   1  the same, M0 written once per FOUR pieces: the instruction offset (+1024 j) moves the LDS destination and the global address
//      alike, so the per-lane pointer of piece j is pre-decremented by 1024 j
//   2  global_load_dwordx4 into registers, then ds_write_b128 (register staging by the loader wave)
//   3  form 1 with the scalar-base + 32-bit-lane-offset address form
// source: L2-resident (every workgroup re-reads a 256 KB window) or streaming (distinct data per workgroup and step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dma_rate_probe dma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PIECES = 8;            // per wave and step (the streamed GEMM: 32 pieces per k-tile over 4 loaders)

template <int MODE, int W>
__global__ __launch_bounds__(64 * W) void dma_kernel(const char* src, long long wg_stride, long long step_stride, int steps, int window_steps, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[160 * 1024];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;
    // ring: W waves x 4 slots x 8 KB  (W <= 4: 128 KB)
    const unsigned mybase = lds0 + (unsigned)wid * 4u * (PIECES * 1024u);
    // the source is read like the A operand of a K = 1024 fp16 GEMM: rows 2 KB apart, wave w owns rows 64 w .. 64 w + 63 of a 256-row
    // panel (512 KB), a step reads one 128-byte k-slice of them (16 steps per panel), piece j = rows 8 j .. 8 j + 7 of the wave's 64
    constexpr int PITCH = 2048;
    const char* base = src + (long long)blockIdx.x * wg_stride + (long long)wid * 64 * PITCH;
    const int lrow = lane >> 3, lslot = lane & 7;
    const long long lane_off = (long long)lrow * PITCH + lslot * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
        const char* p = base + (long long)((s % window_steps) >> 4) * step_stride + ((s & 15) << 7) + lane_off;
        const unsigned dst = mybase + (unsigned)(s & 3) * (PIECES * 1024u);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                const char* pj = p + (long long)j * 8 * PITCH;        // next eight rows
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(pj), "s"(dst + j * 1024u) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int h = 0; h < PIECES / 4; ++h) {
                const char* p0 = p + (long long)(4 * h + 0) * 8 * PITCH, *p1 = p + (long long)(4 * h + 1) * 8 * PITCH - 1024,
                           *p2 = p + (long long)(4 * h + 2) * 8 * PITCH - 2048, *p3 = p + (long long)(4 * h + 3) * 8 * PITCH - 3072;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(dst + h * 4096u) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if constexpr (MODE == 3) {
            const char* sb = base + (long long)((s % window_steps) >> 4) * step_stride + ((s & 15) << 7);       // wave-uniform
#pragma unroll
            for (int h = 0; h < PIECES / 4; ++h) {
                const unsigned o0 = (unsigned)lane_off + (4 * h + 0) * 8 * PITCH, o1 = (unsigned)lane_off + (4 * h + 1) * 8 * PITCH - 1024,
                               o2 = (unsigned)lane_off + (4 * h + 2) * 8 * PITCH - 2048, o3 = (unsigned)lane_off + (4 * h + 3) * 8 * PITCH - 3072;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, %6\n\tglobal_load_lds_dwordx4 %2, %6 offset:1024\n\t"
                             "global_load_lds_dwordx4 %3, %6 offset:2048\n\tglobal_load_lds_dwordx4 %4, %6 offset:3072\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dst + h * 4096u), "s"(sb) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            f32x4 v[PIECES];
#pragma unroll
            for (int j = 0; j < PIECES; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (long long)j * 8 * PITCH);
            char* d = lds + (mybase - lds0) + (s & 3) * (PIECES * 1024) + lane * 16;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) *reinterpret_cast<f32x4*>(d + j * 1024) = v[j];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *reinterpret_cast<const f32x4*>(lds + (threadIdx.x * 16) % (128 * 1024));
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc.y;
}

template <int MODE, int W>
static void run(const char* src, int stream, hipStream_t st, float* sink, size_t bytes_total) {
    const int steps = 256;
    const long long step_bytes = 256LL * 2048;                            // one 256-row panel (16 steps)
    // L2-resident: every workgroup re-reads the same panel; streaming: 16 distinct panels per workgroup
    // stream = 2: SHARED stream - every workgroup walks the same 16 fresh panels in lockstep, as the tiles of a GEMM that share an
    // operand panel do: every line is a compulsory L2 miss for whoever asks first and a hit-on-miss for the others
    const long long wg_stride = stream == 1 ? (long long)(steps / 16) * step_bytes : 0;
    const int window = stream ? steps : 16;
    if ((unsigned long long)(255 * wg_stride + (steps / 16) * step_bytes) > bytes_total) { printf("buffer too small\n"); exit(1); }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((dma_kernel<MODE, W>), dim3(256), dim3(64 * W), 0, st, src, wg_stride, step_bytes, steps, window, sink);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((dma_kernel<MODE, W>), dim3(256), dim3(64 * W), 0, st, src, wg_stride, step_bytes, steps, window, sink);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * W * steps * PIECES * 1024.0;
    const double us = ms * 1e3 / reps;
    const char* names[4] = {"lds-dma, M0 per piece", "lds-dma, M0 per 4 pieces", "global_load + ds_write", "lds-dma saddr, M0 per 4"};
    printf("%-26s %d loader wave(s)  %-9s  %8.1f us   %6.1f GB/s per CU   %5.2f TB/s chip   %5.0f cycles per piece and wave (2.4 GHz)\n", names[MODE], W,
           stream == 1 ? "streaming" : stream == 2 ? "shared" : "L2-hot", us, bytes / 256 / us / 1e3, bytes / us / 1e6, us * 2400.0 / (steps * PIECES));
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const size_t bytes = (size_t)256 * 16 * 256 * 2048 + (1 << 20);         // 256 workgroups x 16 panels x 512 KB = 2 GB
    char* src;
    float* sink;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc(&sink, 4096));
    for (int stream = 0; stream < 3; ++stream) {
        run<0, 1>(src, stream, st, sink, bytes); run<0, 2>(src, stream, st, sink, bytes); run<0, 4>(src, stream, st, sink, bytes);
        run<1, 1>(src, stream, st, sink, bytes); run<1, 2>(src, stream, st, sink, bytes); run<1, 4>(src, stream, st, sink, bytes);
        run<3, 1>(src, stream, st, sink, bytes); run<3, 2>(src, stream, st, sink, bytes); run<3, 4>(src, stream, st, sink, bytes);
        run<2, 1>(src, stream, st, sink, bytes); run<2, 2>(src, stream, st, sink, bytes); run<2, 4>(src, stream, st, sink, bytes);
    }
    return 0;
}
