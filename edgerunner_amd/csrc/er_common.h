// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

// gfx950 only: v_permlane32/16_swap, global_load_lds_dwordx4, 160 KB LDS per CU (k_gemm.h's 128 x 128 fp32 tile holds 66.8 KB)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "edgerunner_hip is written for gfx950 (MI355X): build with --offload-arch=gfx950 (edgerunner_amd/build.py)"
#endif

#define ER_WAVE 64
#define ER_WG 256          // threads per workgroup used by every kernel here
#define ER_NWAVES 4        // ER_WG / ER_WAVE

namespace er {

typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: usable with nontemporal builtins / MFMA

// ---- xor-butterfly steps without the LDS crossbar.  hipcc lowers __shfl_xor to ds_bpermute_b32 (index arithmetic + an LDS-pipe round
// trip + s_waitcnt lgkmcnt(0) per step: a 64-lane sum is six DEPENDENT trips, ~0.25 us at the tail of every GEMV and twice in every
// softmax).  Every step below is one or two VALU instructions: lanes 32 / 16 apart meet through gfx950's v_permlane32_swap /
// v_permlane16_swap (both inputs the same register: the two results are {own half, partner half} in one lane half and {partner, own} in
// the other, and + / max do not care about the order), lanes 8 apart through DPP row_ror:8, 4 apart through row_half_mirror (l ^ 7)
// followed by quad_perm [3,2,1,0] (l ^ 3), 2 and 1 apart through quad_perm.  Each lane still computes op(v[l], v[l ^ OFF]): the
// association of the butterfly - and with it every bit of every result - is unchanged.
typedef unsigned er_u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// v of lane l ^ OFF for OFF = 8, 4, 2, 1
template <int OFF>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "DPP partners inside a row of 16 lanes");
    if constexpr (OFF == 8) return dpp_mov<0x128>(v);                     // row_ror:8
    else if constexpr (OFF == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));   // row_half_mirror, then quad_perm [3,2,1,0]
    else if constexpr (OFF == 2) return dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]
    else return dpp_mov<0xB1>(v);                                         // quad_perm [1,0,3,2]
}
template <int OFF>
__device__ __forceinline__ float xor_sum(float v) {                       // v[l] + v[l ^ OFF] in every lane
    if constexpr (OFF == 32) {
        const er_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r.x) + __uint_as_float(r.y);
    } else if constexpr (OFF == 16) {
        const er_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r.x) + __uint_as_float(r.y);
    } else {
        return v + lane_xor<OFF>(v);
    }
}
template <int OFF>
__device__ __forceinline__ float xor_max(float v) {                       // max(v[l], v[l ^ OFF]) in every lane
    if constexpr (OFF == 32) {
        const er_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    } else if constexpr (OFF == 16) {
        const er_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    } else {
        return fmaxf(v, lane_xor<OFF>(v));
    }
}

// v of lane l ^ OFF for any power of two (the 32 / 16 swaps need the lane's own side to pick the partner's copy); bit pattern
// preserving, for values that are not reduced with a commutative op (arg-max pairs, reduce-scatter)
template <int OFF>
__device__ __forceinline__ unsigned lane_xor_bits(unsigned v, int lane) {
    if constexpr (OFF == 32) {
        const er_u32x2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r.x : r.y;
    } else if constexpr (OFF == 16) {
        const er_u32x2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r.x : r.y;
    } else {
        return __float_as_uint(lane_xor<OFF>(__uint_as_float(v)));
    }
}
template <int OFF>
__device__ __forceinline__ float lane_xor_any(float v, int lane) { return __uint_as_float(lane_xor_bits<OFF>(__float_as_uint(v), lane)); }

// the same with the offset as a value that constant-folds after unrolling (32 >> k in a reduction loop)
__device__ __forceinline__ float lane_xor_pow2(float v, int off, int lane) {
    switch (off) {
        case 32: return lane_xor_any<32>(v, lane);
        case 16: return lane_xor_any<16>(v, lane);
        case 8: return lane_xor_any<8>(v, lane);
        case 4: return lane_xor_any<4>(v, lane);
        case 2: return lane_xor_any<2>(v, lane);
        default: return lane_xor_any<1>(v, lane);
    }
}

// sum over the LPK (4 or 8) adjacent lanes that share a key row: offsets 1, 2 (, 4) in that order
template <int LPK>
__device__ __forceinline__ float lane_group_sum(float v) {
    static_assert(LPK == 4 || LPK == 8, "4 or 8 lanes per row");
    v = xor_sum<1>(v); v = xor_sum<2>(v);
    if constexpr (LPK == 8) v = xor_sum<4>(v);
    return v;
}
// sum over the 64 / LPK lane groups of a wave (lanes LPK, 2 LPK, ... 32 apart, in that order)
template <int LPK>
__device__ __forceinline__ float across_groups_sum(float v) {
    static_assert(LPK == 4 || LPK == 8, "4 or 8 lanes per row");
    if constexpr (LPK == 4) v = xor_sum<4>(v);
    v = xor_sum<8>(v); v = xor_sum<16>(v); v = xor_sum<32>(v);
    return v;
}

// 64-lane butterflies (offsets 32, 16, 8, 4, 2, 1: the order and association of the former __shfl_xor loops)
__device__ __forceinline__ float wave_sum(float v) {
    v = xor_sum<32>(v); v = xor_sum<16>(v); v = xor_sum<8>(v); v = xor_sum<4>(v); v = xor_sum<2>(v); v = xor_sum<1>(v);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    v = xor_max<32>(v); v = xor_max<16>(v); v = xor_max<8>(v); v = xor_max<4>(v); v = xor_max<2>(v); v = xor_max<1>(v);
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

// NB such sums behind ONE barrier and WITHOUT a trailing one (successive reductions use different slots; slot of sum b: slots +
// stride * b); same per-sum order as block_sum
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
