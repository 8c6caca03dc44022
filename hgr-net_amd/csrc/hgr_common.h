// Shared device/host helpers for libhgr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/hgr.h"

#define HGR_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// ---- per-dtype traits: 16-bit MFMA input element ------------------------------------------------
template <int DT> struct T16;
template <> struct T16<HGR_BF16> {
    typedef __bf16 elem;
    typedef bf16x8 vec8;
    typedef bf16x4 vec4;
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct T16<HGR_F16> {
    typedef _Float16 elem;
    typedef f16x8 vec8;
    typedef f16x4 vec4;
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// plain casts: hipcc emits v_cvt_pk_bf16_f32 / v_cvt_f16_f32 (round-to-nearest-even, NaN-preserving)
template <int DT> __device__ __forceinline__ typename T16<DT>::vec4 cvt4(float a, float b, float c, float d) {
    typedef typename T16<DT>::elem E;
    typename T16<DT>::vec4 r;
    r[0] = (E)a; r[1] = (E)b; r[2] = (E)c; r[3] = (E)d;
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane of the row; fixed order.
// Must be executed by all lanes of the wave's rows it concerns (no divergence inside a row).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// ---- host-side error plumbing -------------------------------------------------------------------
int hgr_set_error(int code, const char *fmt, ...);

#define HGR_REQUIRE(cond, ...)                                            \
    do {                                                                  \
        if (!(cond)) return hgr_set_error(HGR_EINVAL, __VA_ARGS__);       \
    } while (0)

#define HGR_CHECK_LAUNCH(name)                                                                      \
    do {                                                                                            \
        hipError_t e__ = hipGetLastError();                                                         \
        if (e__ != hipSuccess) return hgr_set_error(HGR_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline bool hgr_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
