// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define ER_WAVE 64
#define ER_WG 256          // threads per workgroup used by every kernel here
#define ER_NWAVES 4        // ER_WG / ER_WAVE

namespace er {

typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: usable with nontemporal builtins / MFMA

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide (4 waves) reductions through a 4-float LDS scratch.  Every thread
// returns the same value (partials are combined in a fixed order).
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// Same sum (identical order) through a caller-chosen 4-float slot and WITHOUT the trailing barrier: successive
// reductions use different slots, so one barrier per reduction is enough.
template <int NW = ER_NWAVES>
__device__ __forceinline__ float block_sum_slot(float v, float* slot, bool writer = true) {   // writer: wave-uniform, false for waves beyond NW
    static_assert(NW == 3 || NW == 4, "3- or 4-wave reductions");
    v = wave_sum(v);
    if (writer && (threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    if (NW == 3) return (slot[0] + slot[1]) + slot[2];
    return (slot[0] + slot[1]) + (slot[2] + slot[3]);
}

// NB such sums behind ONE barrier (slot of sum b: slots + stride * b); same per-sum order as block_sum_slot
template <int NW, int NB>
__device__ __forceinline__ void block_sum_slots(float (&v)[NB], float* slots, int stride, bool writer) {
    static_assert(NW == 3 || NW == 4, "3- or 4-wave reductions");
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        v[b] = wave_sum(v[b]);
        if (writer && (threadIdx.x & 63) == 0) slots[stride * b + (threadIdx.x >> 6)] = v[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float* s = slots + stride * b;
        v[b] = NW == 3 ? (s[0] + s[1]) + s[2] : (s[0] + s[1]) + (s[2] + s[3]);
    }
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
}

// v of lane (l + N) mod 16 inside the lane's own row of 16 (DPP row_ror: a plain VALU move, no LDS crossbar trip like
// ds_bpermute).  Summing a value with its rotations by 8 (and 4) adds the lanes that sit 8 (4) apart in the row.
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}

// DPP row broadcast (GFX9 wave reductions): CTRL 0x142 = row_bcast:15 (lane 15 of every row -> all lanes of the NEXT row),
// 0x143 = row_bcast:31 (lane 31 -> all lanes of rows 2 and 3); lanes of rows outside ROW_MASK get `fill`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float row_bcast(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

}  // namespace er
