// Shared pieces of the NT GEMM family (hgr_gemm_128 / _256 / _duo .hip + the host entry points in hgr_gemm.hip):
// the argument block, the epilogue helpers and the wait / barrier macros of the counted-vmcnt pipelines.
//
// C[M,N] = epilogue(A[M,K] . W[N,K]^T) on the gfx950 matrix cores.  Both operands are K-contiguous (activations row-major,
// nn.Linear weights [out, in]), so a lane's MFMA fragment (8 consecutive k of one row) is one 16-byte LDS read for either operand.
// W is fed as the MFMA "A" operand and the activations as "B", i.e. a wave computes C^T tiles: the 4 accumulator registers of a
// lane are 4 CONSECUTIVE n of one output row m, so bias / residual / stores are 8- or 16-byte vector accesses.
// Staging is LDS-DMA (global_load_lds_dwordx4) into a lane-linear image of 128-byte rows; the bank-conflict swizzle
// chunk' = chunk ^ (row & 7) is applied on the per-lane SOURCE address and again on the ds_read address (cdna_hip_programming.md
// rule 21).  Block -> tile maps are XCD-aware (blocks b, b + 8 share an L2) with a grouped raster.
#pragma once
#include "hgr_common.h"
#include <stdlib.h>
#include <type_traits>

namespace hgr_gemm {

constexpr int BK = 64;

struct GemmArgs {
    const char *A; int64_t lda;
    const char *W; int64_t ldw;
    void *C; int64_t ldc;
    const float *bias;
    const float *res; int64_t ldr;
    int M, N, K;
    int tiles_m, tiles_n;
    int m_fastest;   // 1: consecutive tile ids walk M first (W panel shared), 0: walk N first
    int vec_ok;      // C / residual rows allow 4-element vector access
    int dbg;         // diagnostics only (HGR_GEMM_DBG): 1 = skip MFMAs, 2 = skip LDS-DMA issue, 3 = skip epilogue
    // implicit-GEMM 3x3 convolution (CONV kernels only): A is an NHWC image [B, H, W, C], pad 1
    int cH, cW, cC, cStride, cHo, cWo;
    unsigned cMagic;  // ceil(2^32 / cC): __umulhi(k, cMagic) == k / cC for every k < 9 * cC + 64 (k * cC < 2^32)
    int cUni;         // cC % 64 == 0: a 64-deep K-tile lies inside ONE tap, so tap / kernel offset are wave-uniform per K-tile (scalar ALU)
    // split-K (gemm_nt_128 only): blockIdx.y = split s works on K columns [s * kc, min(K, (s + 1) * kc)) and writes its own
    // fp32 partial C + s * csplit elements; 0 = off
    int kc; int64_t csplit;
    // LayerNorm folded into the GEMMs around it (gemm_nt_duo only, LN template parameter):
    //   producer (LN = 1, x += A W^T + b): the residual stream is kept as a 16-bit pair (hi, lo) with x = hi + lo - 4 bytes per
    //            element like fp32, and hi IS the next GEMM's A operand; per row and 64-column slot it also emits the partial
    //            (sum, sum of squares) of the new values -> ln_stats [M][ln_slots][2]
    //   consumer (LN = 2, y = LN(x) W^T + b): A is the un-normalised 16-bit x, W the gamma-folded weight,
    //            y = rstd_m (acc - mean_m ln_s[n]) + ln_c[n] with row statistics from ln_stats (K = row width)
    float *ln_stats; int ln_slots; float ln_eps;
    void *ln_xh, *ln_xl; int64_t ln_ldx;      // producer: the residual stream as a 16-bit PAIR, x = hi + lo (hi in the MFMA type, lo f16)
    const float *ln_s, *ln_c;
    // producer range guard: when a 64-column slot's sum of squares exceeds ln_guard (or is inf / NaN) its bit pattern is
    // atomicMax'ed into *ln_flag (0 = the stream stayed in range); null = no guard
    unsigned *ln_flag; float ln_guard;
    int group;       // gemm_nt_duo: row (or column) panels per raster group (HGR_GEMM_GROUP, default 4)
    // gemm_nt_duo tail plan (duo_plan): blocks [0, nbig) = full 256 x 128 tiles on row panels [0, big_panels), the remaining blocks =
    // 128 x 128 half tiles on the rows behind them (tiles_m_half panels of 128 rows).  No tail: nbig = grid, big_panels = tiles_m.
    int nbig, big_panels, tiles_m_half;
    // HGR_EPI_QGELU_GRAD16 on gemm_nt_duo only (hgr_gemm_nt_qgelu_grad_colsum): column sums of the ROUNDED 16-bit outputs per 64-row
    // unit, colsum[unit][n] for unit < colsum_units = ceil(M / 64) - the bias gradient of the layer below without a second pass over C
    float *colsum = nullptr; int colsum_units = 0;
    int total = 0;   // gemm_nt_duo, persistent form: number of virtual blocks (0 = gridDim.x, one tile per workgroup); set by launch_duo
};


// 16 zero bytes every out-of-bounds conv tap (and the K padding) is loaded from; one copy per translation unit (no -fgpu-rdc)
static __device__ __attribute__((aligned(16))) unsigned int hgr_zero_page[4] = {0u, 0u, 0u, 0u};

// x * sigmoid(1.702 x) with v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE fp32 divide costs ~10 VALU ops per
// element and, at 128 elements per lane, dominated the c_fc epilogue (measured 88 us of a 206 us launch).
__device__ __forceinline__ float quick_gelu(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
}


// QuickGELU to 16 bits, the product rounded once (mul16: hipcc merges a product with its conversion in some kernel forms and not in
// others - every form goes through these)
template <int DT> __device__ __forceinline__ typename T16<DT>::elem quick_gelu16(float v) {
    return mul16<DT>(v, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v)));
}
// the expression hgr_quickgelu16 (hgr_train.hip) evaluates, operation for operation: the dual-output forward must give its bits
template <int DT> __device__ __forceinline__ typename T16<DT>::elem quick_gelu_train16(float x) {
    const float z = 1.702f * x;
    return mul16<DT>(x, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)));
}
// d/dx of quick_gelu: s (1 + 1.702 x (1 - s)), s = sigmoid(1.702 x) - the expression of hgr_quickgelu16's backward
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}
// The folded LayerNorm's arithmetic, written with explicit fused multiply-adds: every kernel that finalises row statistics or applies
// them (the consumer epilogues of gemm_nt_duo, hgr_gemm_nt_ln_mha) must round identically, and left to -ffp-contract the compiler is
// free to contract `s2 * inv - mean * mean` and `acc - mean * s` differently from one kernel to the next (the in_proj + attention kernel
// differed from gemm_nt_ln + hgr_mha in 578 of 19.7 M f16 outputs by one ulp before this).
__device__ __forceinline__ float2 ln_finalize(float s1, float s2, float inv_k, float eps) {
    const float mean = s1 * inv_k;
    const float msq = mean * mean;
    return make_float2(mean, rsqrtf(fmaxf(__builtin_fmaf(s2, inv_k, -msq), 0.f) + eps));
}
// rstd * (acc - mean * s_n) + c_n
__device__ __forceinline__ float ln_apply(float2 mr, float acc, float s_n, float c_n) {
    return __builtin_fmaf(mr.y, __builtin_fmaf(-mr.x, s_n, acc), c_n);
}

// four consecutive outputs of a folded-LayerNorm consumer, rounded to the MFMA type: rstd (acc - mean s_n) + c_n [through QuickGELU]
template <int DT, bool GELU> __device__ __forceinline__ u32x2 ln_out16(float2 mr, f32x4 acc, f32x4 s4, f32x4 c4);

// The folded LayerNorm's output through QuickGELU in 4 plain + 2 quarter-rate vector instructions per element (round 6; ln_apply +
// quick_gelu: 5 + 2).  With k = -1.702 log2(e) and the row's rstd and the column's c pre-multiplied by k (once per row / column):
//     w = k v  straight out of the second FMA,   e = 2^w,   v sigmoid(1.702 v) = v / (1 + e) = w / (k + k e)
// Every folded-LayerNorm consumer with QuickGELU (gemm_nt_duo, gemm_nt_ws, gemm_nt_p8) goes through this function, so they keep
// rounding alike.  -DHGR_QGELU_W=0 restores ln_apply + quick_gelu (A/B builds).
#ifndef HGR_QGELU_W
#define HGR_QGELU_W 1
#endif
constexpr float QG_K = -1.702f * 1.4426950408889634f;
__device__ __forceinline__ float ln_apply_gelu(float2 mr, float acc, float s_n, float c_n) {
#if HGR_QGELU_W
    const float w = __builtin_fmaf(mr.y * QG_K, __builtin_fmaf(-mr.x, s_n, acc), c_n * QG_K);
    return w * __builtin_amdgcn_rcpf(__builtin_fmaf(QG_K, __builtin_amdgcn_exp2f(w), QG_K));
#else
    return quick_gelu(ln_apply(mr, acc, s_n, c_n));
#endif
}

template <int DT, bool GELU> __device__ __forceinline__ u32x2 ln_out16(float2 mr, f32x4 acc, f32x4 s4, f32x4 c4) {
    u32x2 o;
    if (GELU) {
#if HGR_QGELU_W
        const float rk = mr.y * QG_K;
        float w[4], r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w[e] = __builtin_fmaf(rk, __builtin_fmaf(-mr.x, s4[e], acc[e]), c4[e] * QG_K);
            r[e] = __builtin_amdgcn_rcpf(__builtin_fmaf(QG_K, __builtin_amdgcn_exp2f(w[e]), QG_K));
        }
        o[0] = mul_pack16<DT>(w[0], r[0], w[1], r[1]);
        o[1] = mul_pack16<DT>(w[2], r[2], w[3], r[3]);
#else
        float v[4], r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = ln_apply(mr, acc[e], s4[e], c4[e]);
            r[e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v[e]));
        }
        o[0] = mul_pack16<DT>(v[0], r[0], v[1], r[1]);
        o[1] = mul_pack16<DT>(v[2], r[2], v[3], r[3]);
#endif
    } else {
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = __builtin_fmaf(-mr.x, s4[e], acc[e]);
        o[0] = fma_pack16<DT>(mr.y, u[0], c4[0], mr.y, u[1], c4[1]);
        o[1] = fma_pack16<DT>(mr.y, u[2], c4[2], mr.y, u[3], c4[3]);
    }
    return o;
}

__host__ __device__ constexpr bool epi_has_bias(int epi) { return epi != HGR_EPI_NONE && epi != HGR_EPI_ACCUM && epi != HGR_EPI_QGELU_GRAD16; }
__host__ __device__ constexpr bool epi_has_idn16(int epi) { return epi == HGR_EPI_BIAS_ADD16_RELU || epi == HGR_EPI_QGELU_GRAD16; }

// One lane's 4 consecutive outputs C[m][n .. n+3] of an accumulator tile: bias / QuickGELU / residual,
// then a 16-byte (fp32) or 8-byte (16-bit) store; scalar tail only at the N edge or for odd strides.
template <int DT, int EPI, bool OUT32>
__device__ __forceinline__ void store_quad(const GemmArgs &p, f32x4 v, int m, int n) {
    typedef typename T16<DT>::elem E;
    if (n + 3 < p.N && p.vec_ok) {
        if (epi_has_bias(EPI)) v += *(const f32x4 *)(p.bias + n);
        if (EPI == HGR_EPI_ACCUM) v += *(const f32x4 *)((const float *)p.C + (int64_t)m * p.ldc + n);
        const f32x4 v0 = v;                 // before the last product (QuickGELU / its gradient): the 16-bit forms round that product once
        f32x4 gq = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        if (EPI == HGR_EPI_BIAS_RESIDUAL) v += *(const f32x4 *)(p.res + (int64_t)m * p.ldr + n);
        if (epi_has_idn16(EPI)) {
            const typename T16<DT>::vec4 idn = *(const typename T16<DT>::vec4 *)((const E *)(const void *)p.res + (int64_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (EPI == HGR_EPI_QGELU_GRAD16) { gq[e] = quick_gelu_grad((float)idn[e]); v[e] = v[e] * gq[e]; }
                else v[e] = v[e] + (float)idn[e];
            }
        }
        if (EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (OUT32) *(f32x4 *)((float *)p.C + (int64_t)m * p.ldc + n) = v;
        else if (EPI == HGR_EPI_BIAS_QUICKGELU || EPI == HGR_EPI_QGELU_GRAD16) {
            typename T16<DT>::vec4 o;                      // the last product and the rounding as ONE operation (v0 = its first factor)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = EPI == HGR_EPI_BIAS_QUICKGELU ? quick_gelu16<DT>(v0[e]) : mul16<DT>(v0[e], gq[e]);
            *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = o;
        } else *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = cvt4<DT>(v[0], v[1], v[2], v[3]);
        return;
    }
    for (int e = 0; e < 4 && n + e < p.N; ++e) {
        float x = v[e];
        if (epi_has_bias(EPI)) x += p.bias[n + e];
        if (EPI == HGR_EPI_ACCUM) x += ((const float *)p.C)[(int64_t)m * p.ldc + n + e];
        const float x0 = x;
        float gx = 0.f;
        if (EPI == HGR_EPI_BIAS_QUICKGELU) x = quick_gelu(x);
        if (EPI == HGR_EPI_BIAS_RESIDUAL) x += p.res[(int64_t)m * p.ldr + n + e];
        if (EPI == HGR_EPI_BIAS_ADD16_RELU) x += (float)((const E *)(const void *)p.res)[(int64_t)m * p.ldr + n + e];
        if (EPI == HGR_EPI_QGELU_GRAD16) { gx = quick_gelu_grad((float)((const E *)(const void *)p.res)[(int64_t)m * p.ldr + n + e]); x *= gx; }
        if (EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) x = fmaxf(x, 0.f);
        if (OUT32) ((float *)p.C)[(int64_t)m * p.ldc + n + e] = x;
        else if (EPI == HGR_EPI_BIAS_QUICKGELU) ((E *)p.C)[(int64_t)m * p.ldc + n + e] = quick_gelu16<DT>(x0);
        else if (EPI == HGR_EPI_QGELU_GRAD16) ((E *)p.C)[(int64_t)m * p.ldc + n + e] = mul16<DT>(x0, gx);
        else ((E *)p.C)[(int64_t)m * p.ldc + n + e] = (E)x;
    }
}

// Interior-tile forms (the caller guarantees vec_ok and that the whole quad is inside C).  The quad's second addend
// (fp32 residual, 16-bit identity or the old C for ACCUM) is loaded by load_addend() and handed to store_quad_full():
// callers fetch a batch of addends, then store the batch.  Interleaved load / store pairs serialise completely -
// residual and C may alias, so hipcc keeps every load behind the previous store with a vmcnt(0) between them.
template <int DT, int EPI>
__device__ __forceinline__ f32x4 load_addend(const GemmArgs &p, int m, int n) {
    typedef typename T16<DT>::elem E;
    if (EPI == HGR_EPI_ACCUM) return *(const f32x4 *)((const float *)p.C + (int64_t)m * p.ldc + n);
    if (EPI == HGR_EPI_BIAS_RESIDUAL) return *(const f32x4 *)(p.res + (int64_t)m * p.ldr + n);
    if (epi_has_idn16(EPI)) {
        const typename T16<DT>::vec4 idn = *(const typename T16<DT>::vec4 *)((const E *)(const void *)p.res + (int64_t)m * p.ldr + n);
        return (f32x4){(float)idn[0], (float)idn[1], (float)idn[2], (float)idn[3]};
    }
    return (f32x4){0.f, 0.f, 0.f, 0.f};
}

template <int DT, int EPI, bool OUT32>
__device__ __forceinline__ void store_quad_full(const GemmArgs &p, f32x4 v, f32x4 bq, f32x4 addend, int m, int n) {
    typedef typename T16<DT>::elem E;
    if (epi_has_bias(EPI)) v += bq;
    if (!OUT32 && (EPI == HGR_EPI_BIAS_QUICKGELU || EPI == HGR_EPI_QGELU_GRAD16)) {
        typename T16<DT>::vec4 o;                          // the last product rounded once (mul16)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = EPI == HGR_EPI_BIAS_QUICKGELU ? quick_gelu16<DT>(v[e]) : mul16<DT>(v[e], quick_gelu_grad(addend[e]));
        *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = o;
        return;
    }
    if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
    }
    if (EPI == HGR_EPI_ACCUM || EPI == HGR_EPI_BIAS_RESIDUAL || EPI == HGR_EPI_BIAS_ADD16_RELU) v += addend;
    if (EPI == HGR_EPI_QGELU_GRAD16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad(addend[e]);
    }
    if (EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (OUT32) *(f32x4 *)((float *)p.C + (int64_t)m * p.ldc + n) = v;
    else *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = cvt4<DT>(v[0], v[1], v[2], v[3]);
}


// LDS-DMA in its BUFFER form (buffer_load_dwordx4 ... offen lds): the per-lane source is a 32-bit byte offset into a buffer resource
// (128-bit descriptor in SGPRs: base, no stride, 4 GB of records, raw dword format) and the K-tile advance rides in the scalar offset
// operand - one address VGPR per instruction and no VALU, where the global form (global_load_lds_dwordx4) needs a 64-bit per-lane
// pointer, i.e. a 64-bit add (two VALU) and two address VGPRs per instruction.  -DHGR_DMA_GLOBAL=1 builds the global form (A/B runs).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, const char *base, unsigned voff, int soff, char *lds_dst) {
#if defined(HGR_DMA_GLOBAL) && HGR_DMA_GLOBAL
    __builtin_amdgcn_global_load_lds((const AS1 void *)(base + soff + voff), (AS3 void *)lds_dst, 16, 0, 0);
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void *)lds_dst, 16, (int)voff, soff, 0, 0);
#endif
}

// R interval end: my share of the piece the NEXT read interval needs has landed (counted vmcnt), my own
// ds_reads are complete (so the slot they read may be refilled), then the barrier.  M interval end: barrier.
#define HGR_RWAIT(N) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HGR_RBAR() do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HGR_MBAR() do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)


// 16 MFMAs of a phase interleaved with NR LDS reads, NW LDS writes and NV vector-memory instructions (LDS-DMAs) that do not depend on
// them: the scheduler is told to issue them in the gaps between MFMAs instead of in a block in front (gemm_nt_ws, -DHGR_DUO_PF=1)
template <int NR, int NW, int NV, int M_, int NM = 16>
__device__ __forceinline__ void mfma16_interleave_step() {
    if constexpr (M_ < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int DR = (M_ + 1) * NR / NM - M_ * NR / NM, DW = (M_ + 1) * NW / NM - M_ * NW / NM, DV = (M_ + 1) * NV / NM - M_ * NV / NM;
        if constexpr (DR > 0) __builtin_amdgcn_sched_group_barrier(0x100, DR, 0);
        if constexpr (DW > 0) __builtin_amdgcn_sched_group_barrier(0x200, DW, 0);
        if constexpr (DV > 0) __builtin_amdgcn_sched_group_barrier(0x020, DV, 0);
        mfma16_interleave_step<NR, NW, NV, M_ + 1, NM>();
    }
}
template <int NR, int NW, int NV>
__device__ __forceinline__ void mfma16_interleave() { mfma16_interleave_step<NR, NW, NV, 0>(); }
// the same over 8 MFMAs (one k half of a phase)
template <int NR, int NW, int NV>
__device__ __forceinline__ void mfma8_interleave() { mfma16_interleave_step<NR, NW, NV, 0, 8>(); }

// ---- launchers, one per translation unit (the kernels are templates; host code selects by value) ------------------------
enum { V128_PLAIN = 0, V128_TALL = 1, V128_CONV = 2, V128_CONV_TALL = 3 };
// gemm_nt_128 family.  PLAIN: epi any (K == 64 with 16-bit bias epilogues takes the one-stage variant); TALL: 256 x 64 tiles,
// epi = BIAS_RELU, 16-bit out; CONV / CONV_TALL: implicit-GEMM 3x3 convolution, epi = BIAS_RELU or NONE, 16-bit out.
void launch_128(const GemmArgs &a, int dtype, int epi, bool out32, int variant, dim3 grid, hipStream_t s);
// gemm_nt_256: plain (any epi) or the implicit-GEMM convolution (BIAS_RELU, 16-bit out)
void launch_256(const GemmArgs &a, int dtype, int epi, bool out32, bool conv, dim3 grid, hipStream_t s);
// gemm_nt_duo: ln = 0 plain (any epi), 1 LayerNorm producer, 2 LayerNorm consumer (epi BIAS / BIAS_QUICKGELU), 4 dual output
// (pre-activation + QuickGELU), 5 the 3 x 3 convolution
void launch_duo(const GemmArgs &a, int dtype, int epi, bool out32, int ln, dim3 grid, hipStream_t s);
int duo_set_persist(int enabled);     // hgr_gemm_set_persist
// gemm_nt_ws (hgr_gemm_ws.hip): 256 x 128 tiles, ONE persistent workgroup per CU of 4 matrix waves + 4 helper waves; the epilogue of
// tile i runs in the helper waves under the MFMAs of tile i + 1.  Same bits as gemm_nt_duo.  mode WS_PLAIN: 16-bit C = act(A W^T
// [+ bias]), act 0 none / 1 QuickGELU / 2 ReLU; WS_LNC: the folded-LayerNorm consumer (act 0 / 1); WS_LNP: the residual producer.
enum { WS_PLAIN = 0, WS_LNC = 1, WS_LNP = 2 };
bool ws_covers(int M, int N, int K, int mode);      // whole tiles, an even number >= 12 of K-tiles, at least one tile per CU
void launch_ws(const GemmArgs &a, int dtype, int mode, int act, bool hasb, hipStream_t s);
int ws_enabled();                     // HGR_WS (default 0: an experiment that did not beat gemm_nt_duo, kept bit-identical and tested)
int ws_set(int enabled);              // hgr_gemm_set_ws
// hgr_gemm_p8.hip: the LayerNorm-folded consumer as one persistent 512-thread workgroup per CU on 256 x 256 tiles
bool p8_covers(int M, int N, int K);
void launch_p8(const GemmArgs &a, int dtype, int act, hipStream_t s);
bool p8_wanted(int M, int N, int K);  // by shape, or as forced by HGR_P8 / hgr_gemm_set_p8
int p8_set(int mode);                 // hgr_gemm_set_p8: 0 never, 1 wherever it covers, 2 by shape (default)

// first stage of hgr_logits_eval (hgr_logits_slab.hip): 512-row x 96-column tiles, one per CU, evaluation consumers in the epilogue
struct SlabArgs {
    const char *A; int64_t lda;          // features [M, K] 16-bit
    const char *W; int64_t ldw;          // level-sorted class matrix [Np, K] 16-bit, Np % 96 == 0
    int M, K, Np;
    const int *tpos, *epos;              // [Np] train / test position of every permuted column (-1 = not in the subset / padding)
    unsigned long long *ev_key;          // [M][S]      S = Np / 32
    float *ev_tmax, *ev_m2; int *ev_p1;  // [M][S][2]
    int S;
    int dbg;                             // HGR_LS_DBG (timing experiments only, wrong results): 1 skip the MFMAs, 2 skip the feature DMAs, 4 skip the class-row DMAs, 8 skip the epilogue, 16 skip the warm-up touches
};
void launch_logits_slab(const SlabArgs &a, int dtype, hipStream_t s);

}  // namespace hgr_gemm
