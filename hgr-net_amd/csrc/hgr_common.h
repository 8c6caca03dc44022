// Shared device/host helpers for libhgr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/hgr.h"

#define HGR_WAVE 64

// Ablation switches (HGR_GEMM_DBG, HGR_WS_DBG, HGR_LS_DBG, HGR_LE_DBG: parts of a kernel left out for timing experiments - WRONG
// results) exist in `make lab` builds only (-DHGR_LAB, ../lib/libhgr_lab.so).  In libhgr.so HGR_LAB_ON() is the constant false: the
// branches fold away and no environment variable can switch a product kernel into a wrong-result mode.
#ifdef HGR_LAB
#define HGR_LAB_ON(expr) (expr)
static inline int hgr_lab_env(const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; }
#else
#define HGR_LAB_ON(expr) false
static inline int hgr_lab_env(const char *) { return 0; }
#endif

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

// (a0 b0 + c0, a1 b1 + c1) as a packed pair of the MFMA type, first value in the low half.  f16: v_fma_mixlo_f16 / v_fma_mixhi_f16 - the
// fp32 FMA rounded ONCE to f16, and two instructions where fma + fma + convert were three; bf16 (no such instruction): fp32 FMA, then
// v_cvt_pk_bf16_f32.  Written as inline asm because hipcc makes this choice by context (round 5: one kernel form fused the product
// with its conversion and its twin did not; round 6: without the SLP vectoriser the choice flipped in gemm_nt_duo and not in
// qkv_attn): every kernel that rounds a folded-LayerNorm output - or its QuickGELU - to 16 bits goes through these two functions.
template <int DT> __device__ __forceinline__ unsigned fma_pack16(float a0, float b0, float c0, float a1, float b1, float c1) {
    if (DT == HGR_F16) {
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a0), "v"(b0), "v"(c0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(d) : "v"(a1), "v"(b1), "v"(c1));
        return d;
    }
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    b2 r;
    r[0] = (__bf16)__builtin_fmaf(a0, b0, c0); r[1] = (__bf16)__builtin_fmaf(a1, b1, c1);
    return __builtin_bit_cast(unsigned, r);
}
template <int DT> __device__ __forceinline__ unsigned mul_pack16(float a0, float b0, float a1, float b1) {
    if (DT == HGR_F16) {
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(a0), "v"(b0));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(a1), "v"(b1));
        return d;
    }
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    b2 r;
    r[0] = (__bf16)(a0 * b0); r[1] = (__bf16)(a1 * b1);
    return __builtin_bit_cast(unsigned, r);
}

// one product rounded ONCE to the MFMA type (scalar form of mul_pack16: the compiler packs the halves); conv16: a plain conversion that
// the compiler cannot merge with the arithmetic in front of it (it is handed the value through an opaque register copy)
template <int DT> __device__ __forceinline__ typename T16<DT>::elem mul16(float a, float b) {
    if (DT == HGR_F16) {
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
        return __builtin_bit_cast(typename T16<DT>::elem, (unsigned short)d);
    }
    return (typename T16<DT>::elem)(a * b);
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
// The same split for TWO values at a time as the hot epilogues use it (gemm_nt_duo's residual producer: ~8 -> ~5 vector instructions
// per element, round 6): returns the two hi halves packed in one dword (x0 low) and the rounded bit patterns tq0 / tq1 whose bits
// S-1 .. S-8 are the two q bytes - for f16 already replaced by a pattern with q = 128 below the f16 normal range.  Same bits as
// pair_split(): f16 hi = ONE v_cvt_pk_f16_f32 of the masked patterns; the range test is a float compare with |.| as a source modifier.
template <int DT> __device__ __forceinline__ unsigned pair_split2(float x0, float x1, unsigned &tq0, unsigned &tq1) {
    constexpr int S = DT == HGR_F16 ? 13 : 16;
    const unsigned t0 = __float_as_uint(x0) + (1u << (S - 1)), t1 = __float_as_uint(x1) + (1u << (S - 1));
    if (DT == HGR_F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        h2 h;
#if defined(HGR_PAIR_RTZ) && HGR_PAIR_RTZ       // experiment: truncating conversion of t instead of mask + exact conversion (differs below the f16 normal range only)
        h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(t0), __uint_as_float(t1)));
#else
        h[0] = (_Float16)__uint_as_float(t0 & ~((1u << S) - 1u));
        h[1] = (_Float16)__uint_as_float(t1 & ~((1u << S) - 1u));
#endif
        tq0 = __builtin_fabsf(__uint_as_float(t0)) < 6.103515625e-05f ? (128u << (S - 8)) : t0;
        tq1 = __builtin_fabsf(__uint_as_float(t1)) < 6.103515625e-05f ? (128u << (S - 8)) : t1;
        return __builtin_bit_cast(unsigned, h);
    }
    tq0 = t0; tq1 = t1;
    return __builtin_amdgcn_perm(t1, t0, 0x07060302u);        // (t1 & 0xFFFF0000) | (t0 >> 16)
}
// byte BYTE of q4 = bits S-1 .. S-8 of tq, the other bytes kept (BYTE 0: cleared): ONE SDWA shift that writes a single destination
// byte, where shift + mask + or (and the compiler's per-element compare / select on the shifted byte) were ~4 instructions
template <int DT, int BYTE> __device__ __forceinline__ void pair_put_q(unsigned &q4, unsigned tq) {
    constexpr int S = DT == HGR_F16 ? 13 : 16;
    const unsigned sh = S - 8;
    if (BYTE == 0) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(q4) : "v"(sh), "v"(tq));
    else if (BYTE == 1) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(q4) : "v"(sh), "v"(tq));
    else if (BYTE == 2) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(q4) : "v"(sh), "v"(tq));
    else asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(q4) : "v"(sh), "v"(tq));
}
// pair_dec() of byte BYTE of q4 (four q bytes as they are stored): the byte select rides in the shift (SDWA), the - 128 in a 3-input add
template <int DT, int BYTE> __device__ __forceinline__ float pair_dec4(typename T16<DT>::elem hi, unsigned q4) {
    constexpr int S = DT == HGR_F16 ? 13 : 16;
    const unsigned sh = S - 8;
    unsigned qs;
    if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(qs) : "v"(sh), "v"(q4));
    else if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(qs) : "v"(sh), "v"(q4));
    else if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(qs) : "v"(sh), "v"(q4));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(qs) : "v"(sh), "v"(q4));
    return __uint_as_float(__float_as_uint((float)hi) + qs + (unsigned)(-(128 << (S - 8))));
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
