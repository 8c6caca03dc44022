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

// ---- the residual stream's 16-bit-plus-8-bit PAIR (round 4) ----------------------------------------------------------------------
// x (fp32) is kept as hi - x in the MFMA type (f16 / bf16: it IS the next GEMM's A operand) - and ONE byte q, both cut out of x's own
// bit pattern.  With S = 13 (f16) / 16 (bf16) mantissa bits dropped by hi:
//     t  = bits(x) + 2^(S-1)          (round the magnitude half up at hi's last place; float bit patterns are monotonic integers)
//     hi = t with its low S bits cleared  (exactly representable in the MFMA type)      q = bits S-1 .. S-8 of t
//     decode:  bits(x') = bits(hi) + ((q - 128) << (S - 8))  =  bits(x) with its low S - 8 bits cleared
// i.e. x to 8 more mantissa bits than hi alone (19 / 16 significant bits; |x - x'| < ulp(hi) / 256, towards zero) in 3 bytes, for ~2
// integer operations per element each way - rounds 2 - 3 kept lo = f16(x - hi): 22 bits in 4 bytes.  The residual producers are
// bound by this stream's read-modify-write; a byte less each way is 39 MB per launch at ViT-B/32 batch 512.  (A first round-4 form -
// q = rint((x - hi) 254 / ulp(hi)) through v_ldexp / fma, ~13 floating-point operations per element - saved the bytes and lost the
// same time to VALU: 5.01 vs 5.01 ms per step.)  hi differs from round-to-nearest-EVEN on exact ties only (half away from zero).
// Values below the f16 normal range (|x| < 2^-14) keep an absolute error <= 2^-25 instead of a relative one; inf / NaN are the range
// guard's business.  Every kernel that touches the pair uses these two functions.
template <int DT> __device__ __forceinline__ void pair_split(float x, typename T16<DT>::elem &hi, unsigned &q) {
    constexpr int S = DT == HGR_F16 ? 13 : 16;
    const unsigned t = __float_as_uint(x) + (1u << (S - 1));
    q = (t >> (S - 8)) & 255u;
    if (DT == HGR_F16) {
        // below the f16 normal range the conversion is not a bit copy (it rounds into subnormals, or to zero): no extra bits there -
        // q = 128 decodes to hi itself (|x - hi| <= 2^-25); a stale q < 128 on hi = 0 would decode to a NaN pattern
        if ((t & 0x7FFFFFFFu) < 0x38800000u) q = 128u;
        hi = (typename T16<DT>::elem)__uint_as_float(t & ~((1u << S) - 1u));
    } else hi = __builtin_bit_cast(typename T16<DT>::elem, (unsigned short)(t >> 16));
}
template <int DT> __device__ __forceinline__ float pair_dec(typename T16<DT>::elem hi, unsigned q) {
    constexpr int S = DT == HGR_F16 ? 13 : 16;
    return __uint_as_float(__float_as_uint((float)hi) + (unsigned)(((int)q - 128) * (1 << (S - 8))));
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
