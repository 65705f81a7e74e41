// Round 6: where the DiT self-attention kernel (k_flash_attn.h flash_attn_hh_kernel: 2 x 16 heads, 2048 queries x 2048 keys, head_dim 64;
// 22 % of a guided DiT forward) spends its cycles.  The library kernel with s_memtime stamps (-DFA_TIMELINE): per key tile of 64 keys
// and wave - counted wait + barrier + next tile's LDS-DMA issue | S^T = K Q^T (8 MFMAs) | softmax | O^T += V^T P^T (8 MFMAs) - next to
// the launch's wall time.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DFA_TIMELINE -I ../../edgerunner_amd/csrc -o flash_hh_timeline_probe flash_hh_timeline_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "er_common.h"
#include "k_flash_attn.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace er;

int main() {
    const int B = 2, H = 16, N = 2048, D = 64, C = H * D;
    for (int M : {2048, 320}) {       // self-attention; cross-attention to 257 condition tokens padded to 320
        _Float16 *qkv, *vt, *o;
        CHECK(hipMalloc(&qkv, (size_t)B * N * 3 * C * 2));
        CHECK(hipMalloc(&vt, (size_t)B * H * 64 * N * 2));
        CHECK(hipMalloc(&o, (size_t)B * N * C * 2));
        std::vector<_Float16> h((size_t)B * N * 3 * C);
        unsigned x = 11u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((float)(x >> 8) / 8388608.0f - 1.0f) * 1.5f); }
        CHECK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(vt, h.data(), (size_t)B * H * 64 * N * 2, hipMemcpyHostToDevice));
        FlashHArgs fh{};
        fh.Q = qkv; fh.K = qkv + C; fh.Vt = vt; fh.O16 = o; fh.N = N; fh.M = M == 2048 ? N : 257;
        fh.ldq = fh.ldk = 3 * C; fh.ldvt = M == 2048 ? N : 320; fh.ldo = C;
        fh.qs_b = fh.ks_b = (long long)N * 3 * C; fh.vts_h = 64LL * fh.ldvt; fh.vts_b = (long long)H * 64 * fh.ldvt; fh.os_b = (long long)N * C;
        fh.head_stride = D; fh.scale = 0.125f;
        const int nwg = (N / 128) * H * B;
        float* dbg;
        CHECK(hipMalloc(&dbg, (size_t)nwg * 16 * 4));
        CHECK(hipMemset(dbg, 0, (size_t)nwg * 16 * 4));
#ifdef FA_TIMELINE
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(fa_timeline_out), &dbg, sizeof(dbg)));
#endif
        hipStream_t st;
        CHECK(hipStreamCreate(&st));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(launch_flash_attn_hh(fh, H, B, st));
        CHECK(hipStreamSynchronize(st));
        const int reps = 20;
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHECK(launch_flash_attn_hh(fh, H, B, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> d((size_t)nwg * 16);
        CHECK(hipMemcpy(d.data(), dbg, d.size() * 4, hipMemcpyDeviceToHost));
        double m[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < d.size(); ++i) m[i & 3] += d[i] / (d.size() / 4);
        const double us = ms * 1e3 / reps, flop = 4.0 * B * H * (double)N * fh.M * D;
        const int ntiles = (fh.M + 63) / 64;

        printf("%d keys: %7.1f us per launch (%6.1f TFLOP/s);  cycles per key tile and wave: wait + barrier + DMA issue %5.0f | S = K Q^T %5.0f | softmax %5.0f | "
               "P V %5.0f | sum %5.0f x %d tiles = %.0f cycles -> clock ~%.2f GHz if the loop is the launch\n", fh.M, us, flop / us / 1e6, m[0], m[1], m[2], m[3],
               m[0] + m[1] + m[2] + m[3], ntiles, (m[0] + m[1] + m[2] + m[3]) * ntiles, (m[0] + m[1] + m[2] + m[3]) * ntiles / (us - 2.0) / 1e3);
        hipFree(qkv); hipFree(vt); hipFree(o); hipFree(dbg);
    }
    return 0;
}
