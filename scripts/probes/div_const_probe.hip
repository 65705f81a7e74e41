// Is x / c, for the two score divisors of the exact path (c = sqrtf(96), sqrtf(64)), reproduced BIT FOR BIT by the three-instruction
// sequence  q0 = x * y;  r = fma(-q0, c, x);  q = fma(r, y, q0)  with y = 1.0f / c (Markstein's correction step)?  hipcc lowers a
// float division to ~10 instructions (v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup); the exact-mode attention kernels
// divide every score by sqrt(D) as the reference does (core/transformer/attention.py:52).  Exhaustive over all 2^32 bit patterns of x;
// NaN inputs must give NaN, everything else the same bits.  Reports the mismatches by class (the sequence is NOT exact for results in
// the subnormal range or for |x| near overflow: the kernels' scores are O(1..100), so the range actually needed is printed too).
//   hipcc --offload-arch=gfx950 -O3 -o div_const_probe div_const_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__global__ void check(float c, float y, unsigned long long* bad_all, unsigned long long* bad_range, unsigned* first_bad) {
    const unsigned long long n = 1ull << 32;
    unsigned long long ba = 0, br = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float want = x / c;
        const float q0 = x * y;
        const float r = __builtin_fmaf(-q0, c, x);
        const float got = __builtin_fmaf(r, y, q0);
        const bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want != want && got != got);
        if (!same) {
            ++ba;
            const float ax = fabsf(x);
            if (ax >= 1e-30f && ax <= 1e30f) { ++br; atomicMin(first_bad, (unsigned)i); }
        }
    }
    atomicAdd(bad_all, ba);
    atomicAdd(bad_range, br);
}

int main() {
    unsigned long long *d, h[2];
    unsigned *fb, hfb;
    hipMalloc(&d, 16);
    hipMalloc(&fb, 4);
    for (int D : {96, 64}) {
        const float c = sqrtf((float)D), y = 1.0f / c;
        hipMemset(d, 0, 16);
        hipMemset(fb, 0xff, 4);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, c, y, d, d + 1, fb);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        hipMemcpy(&hfb, fb, 4, hipMemcpyDeviceToHost);
        printf("c = sqrtf(%d) = %.9g, y = %.9g: %llu of 2^32 inputs differ from x / c; %llu of them with 1e-30 <= |x| <= 1e30", D, c, y, h[0], h[1]);
        if (h[1]) printf(" (first: bits 0x%08x)", hfb);
        printf("\n");
    }
    return 0;
}
