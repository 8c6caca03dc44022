// hgr_gemm_nt: C[M,N] = epilogue(A[M,K] . W[N,K]^T) on the gfx950 matrix cores.
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out, in]), so a lane's
// MFMA fragment (8 consecutive k of one row) is one 16-byte LDS read for either operand.
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2(M) x 2(N), each wave 64 x 64 =
// 4 x 4 tiles of v_mfma_f32_16x16x32.  W is fed as the MFMA "A" operand and the activations as
// "B", i.e. the wave computes C^T tiles: the 4 accumulator registers of a lane are then 4
// CONSECUTIVE n of one output row m, so bias/residual/stores are 8- or 16-byte vector accesses.
//
// Staging: global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) into two LDS stages of 32 KB; the LDS
// image is lane-linear ([row][8 chunks of 16 B], 128-B rows), the bank-conflict swizzle
// chunk' = chunk ^ (row & 7) is applied on the per-lane SOURCE address and again on the ds_read
// address (cdna_hip_programming.md rule 21).  With it every ds_read_b128 of a fragment is
// conflict-free: a 16-lane read group touches 16 distinct 16-B slots of the 256-B bank row.
// Loop: issue stage t+1, compute stage t, vmcnt(0) + one barrier per K-tile ("minimum 2-phase").
//
// Block -> tile map is XCD-aware: consecutive tile ids (which share an operand panel) are dealt to
// ONE XCD (blocks b and b+8 share an XCD's L2), and the fastest-varying tile index is chosen on the
// host so that the larger operand is read from HBM once (see hgr_gemm_nt below).
#include "hgr_common.h"
#include <stdlib.h>
#include <type_traits>

int hgr_conv3x3_c32_launch(const void *x, const void *w, const float *bias, void *out, int B, int H, int W, int Cout, int Kp,
                           int dtype, int relu, void *stream, int pool);      // hgr_conv_direct.hip

namespace {

int hgr_gemm_force_tile();

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT = 256;

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
    int group;       // gemm_nt_duo: row (or column) panels per raster group (HGR_GEMM_GROUP, default 4)
    // fused evaluation (LN = 3, hgr_logits_eval): W = level-sorted, level-aligned class matrix, so every 64-column slice (one
    // wave's share of a tile) lies inside ONE hierarchy level.  Nothing of C is written; per (row, slice) the epilogue emits
    // the best train column as an orderable key and the largest value over the test columns.
    unsigned long long *ev_key; float *ev_tmax, *ev_m2; int *ev_p1; const int *ev_tpos, *ev_epos; int ev_slices;
};

// 16 zero bytes every out-of-bounds conv tap (and the K padding) is loaded from
__device__ __attribute__((aligned(16))) unsigned int hgr_zero_page[4] = {0u, 0u, 0u, 0u};

// x * sigmoid(1.702 x) with v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE fp32 divide costs ~10 VALU ops per
// element and, at 128 elements per lane, dominated the c_fc epilogue (measured 88 us of a 206 us launch).
__device__ __forceinline__ float quick_gelu(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
}


// the expression hgr_quickgelu16 (hgr_train.hip) evaluates, operation for operation: the dual-output forward must give its bits
__device__ __forceinline__ float quick_gelu_train(float x) {
    const float z = 1.702f * x;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
// d/dx of quick_gelu: s (1 + 1.702 x (1 - s)), s = sigmoid(1.702 x) - the expression of hgr_quickgelu16's backward
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
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
        if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        if (EPI == HGR_EPI_BIAS_RESIDUAL) v += *(const f32x4 *)(p.res + (int64_t)m * p.ldr + n);
        if (epi_has_idn16(EPI)) {
            const typename T16<DT>::vec4 idn = *(const typename T16<DT>::vec4 *)((const E *)(const void *)p.res + (int64_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = EPI == HGR_EPI_QGELU_GRAD16 ? v[e] * quick_gelu_grad((float)idn[e]) : v[e] + (float)idn[e];
        }
        if (EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (OUT32) *(f32x4 *)((float *)p.C + (int64_t)m * p.ldc + n) = v;
        else *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = cvt4<DT>(v[0], v[1], v[2], v[3]);
        return;
    }
    for (int e = 0; e < 4 && n + e < p.N; ++e) {
        float x = v[e];
        if (epi_has_bias(EPI)) x += p.bias[n + e];
        if (EPI == HGR_EPI_ACCUM) x += ((const float *)p.C)[(int64_t)m * p.ldc + n + e];
        if (EPI == HGR_EPI_BIAS_QUICKGELU) x = quick_gelu(x);
        if (EPI == HGR_EPI_BIAS_RESIDUAL) x += p.res[(int64_t)m * p.ldr + n + e];
        if (EPI == HGR_EPI_BIAS_ADD16_RELU) x += (float)((const E *)(const void *)p.res)[(int64_t)m * p.ldr + n + e];
        if (EPI == HGR_EPI_QGELU_GRAD16) x *= quick_gelu_grad((float)((const E *)(const void *)p.res)[(int64_t)m * p.ldr + n + e]);
        if (EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) x = fmaxf(x, 0.f);
        if (OUT32) ((float *)p.C)[(int64_t)m * p.ldc + n + e] = x;
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

// TALL = the 4 waves stacked along M: a 256 (M) x 64 (N) tile for outputs at most 64 wide (the 64-channel stages of the
// ResNet towers), where the square tile would spend half of its MFMAs and LDS traffic on columns that do not exist.
// ONEK = K is a single 64-wide tile (the 1x1 convolutions out of 64-channel stages): no second LDS stage, so the
// workgroup needs 32 KB instead of 64 KB and 3-4 of them (VGPR-limited) share a CU - these launches are HBM-bound
// streams whose only latency hiding is other workgroups.
template <int DT, int EPI, bool OUT32, bool CONV = false, bool TALL = false, bool ONEK = false>
__global__ __launch_bounds__(NT) void gemm_nt_128(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    constexpr int BM = TALL ? 256 : 128, BN = TALL ? 64 : 128;
    constexpr int PA = BM / 32, PW = BN / 32;              // 4 KB LDS-DMA pieces (32 rows of 128 B) per operand: one per wave each
    constexpr int TILE_A = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
    __shared__ __attribute__((aligned(1024))) char smem[(ONEK ? 1 : 2) * STAGE_BYTES];
    if (p.kc) {                                   // split-K: this workgroup's slice of the reduction, its own partial output
        const int sp = blockIdx.y;
        p.A += (int64_t)sp * p.kc * 2; p.W += (int64_t)sp * p.kc * 2;
        p.C = (char *)p.C + (int64_t)sp * p.csplit * 4;
        p.K = min(p.kc, p.K - sp * p.kc);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = TALL ? wave : wave >> 1, wn = TALL ? 0 : wave & 1;
    const int r = lane & 15, g = lane >> 4;

    // XCD-aware, bijective remap of the 1-D grid (cdna_hip_programming.md T1)
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    // Grouped raster inside the XCD's range: GROUP panels of the big operand stay L2-resident while
    // the walk slides over the other operand's panels, so ~GROUP concurrently running tiles share every
    // panel either way (64 tiles in flight per XCD = 8 x 8 panels of 128 x K: ~3 MB at K = 768 < 4 MB L2).
    constexpr int GROUP = 8;
    int tm, tn;
    if (p.m_fastest) {
        const int per = GROUP * p.tiles_m, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_n - first), loc = wg - grp * per;
        tn = first + loc % gs; tm = loc / gs;
    } else {
        const int per = GROUP * p.tiles_n, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_m - first), loc = wg - grp * per;
        tm = first + loc % gs; tn = loc / gs;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // per-lane source rows of the PA + PW LDS-DMA pieces this thread issues per stage
    const char *srcA[PA], *srcW[PW];
    int cchunk[PA];             // CONV: this lane's logical 16-B chunk (8 input channels) inside a K-tile
    unsigned vmask[PA];         // CONV: bit t set = tap t of this output pixel is inside the image
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int id = (i * 4 + wave) * 64 + lane;    // 16-B chunk id inside the BM x 64 tile
        const int row = id >> 3, c = (id & 7) ^ (row & 7);
        const int gm = min(m0 + row, p.M - 1);        // edge rows: load a valid row, never store it
        if (i < PW) {
            const int gn = min(n0 + row, p.N - 1);
            srcW[i] = p.W + ((int64_t)gn * p.ldw + c * 8) * 2;
        }
        if (CONV) {
            // output pixel (b, ho, wo) of row gm; tap (ky, kx) reads input pixel (ho*s - 1 + ky, wo*s - 1 + kx)
            const int wo = gm % p.cWo, t1 = gm / p.cWo, ho = t1 % p.cHo, b = t1 / p.cHo;
            const int hi0 = ho * p.cStride - 1, wi0 = wo * p.cStride - 1;
            unsigned vm = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hi = hi0 + t / 3, wi = wi0 + t % 3;
                if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW) vm |= 1u << t;
            }
            vmask[i] = vm;
            cchunk[i] = c;
            srcA[i] = p.A + (((int64_t)b * p.cH + hi0) * p.cW + wi0) * p.cC * 2;     // tap (0,0), channel 0
        } else {
            srcA[i] = p.A + ((int64_t)gm * p.lda + c * 8) * 2;
        }
    }

    auto stage = [&](int buf, int kt) {
        char *sA = smem + buf * STAGE_BYTES;
        char *sW = sA + TILE_A;
        const int64_t koff = (int64_t)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const char *src;
            if (CONV) {
                // implicit im2col: K index = tap * C + channel (C a multiple of 8), 8 channels per chunk
                const int kq = kt * BK + cchunk[i] * 8;
                const int tap = (int)__umulhi((unsigned)kq, p.cMagic), cin = kq - tap * p.cC;
                const int ky = (tap * 11) >> 5, kx = tap - ky * 3;               // tap / 3, tap % 3 for tap < 9
                const bool ok = tap < 9 && ((vmask[i] >> tap) & 1u);
                src = ok ? srcA[i] + ((int64_t)(ky * p.cW + kx) * p.cC + cin) * 2 : (const char *)hgr_zero_page;
            } else {
                src = srcA[i] + koff;
            }
            __builtin_amdgcn_global_load_lds((const AS1 void *)src, (AS3 void *)(sA + (i * 4 + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void *)(srcW[i] + koff),
                                             (AS3 void *)(sW + (i * 4 + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = ONEK ? 1 : p.K / BK;
    stage(0, 0);
    // 16-bit outputs of interior tiles leave through LDS ("wide"): the wave's 64 x 64 tile is written as fp32 quads into its
    // own 8 KB (two passes of 32 rows, 256-byte rows, 16-byte chunk ^= row & 15) and read back 8 columns per lane, so the
    // 16-bit identity of the ResNet residual is LOADED and the result is STORED as full 128-byte lines (16 bytes per lane,
    // 8 lanes per row) instead of 8-byte quads that touch 32-byte fragments of 16 different rows per instruction
    // (1x1 conv 64 -> 256 over 1.6 M pixels, bias epilogue: 273 -> 188 us = 5.5 TB/s; with identity + ReLU 468 -> 390 us).
    // The arithmetic is unchanged: bias and identity are added in fp32, one rounding at the end.
    // ONEK: the whole epilogue input (bias quads, first half of the identity) is requested together with the operands, so
    // a workgroup waits for memory once instead of three times (operands -> bias -> identity).
    typedef typename T16<DT>::elem E;
    constexpr bool HAS_BIAS = epi_has_bias(EPI);
    constexpr bool HAS_IDN = epi_has_idn16(EPI);
    const bool full = p.vec_ok && m0 + BM <= p.M && n0 + BN <= p.N;
    const bool wide = !OUT32 && full && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
                      (!HAS_IDN || ((p.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0));
    const int och = lane & 7, orr = lane >> 3;          // output chunk (8 columns) and row-in-group of this lane
    const E *idn_src = (const E *)(const void *)p.res + (int64_t)(m0 + wm * 64 + orr) * p.ldr + n0 + wn * 64 + och * 8;
    f32x4 pbq[ONEK ? 4 : 1];
    u32x4 pidn[ONEK && HAS_IDN ? 4 : 1];
    if (ONEK && wide) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pbq[i] = HAS_BIAS ? *(const f32x4 *)(p.bias + n0 + wn * 64 + i * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (HAS_IDN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pidn[q] = *(const u32x4 *)(idn_src + (int64_t)q * 8 * p.ldr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment read offsets: row & 7 == r & 7 for every fragment of this lane
    const int offW = (wn * 64 + r) * 128;
    const int offA = (wm * 64 + r) * 128;
    const int sw0 = ((0 + g) ^ (r & 7)) * 16;
    const int sw1 = ((4 + g) ^ (r & 7)) * 16;

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (!ONEK && kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char *sA = smem + cur * STAGE_BYTES;
        const char *sW = sA + TILE_A;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk ? sw1 : sw0;
            vec8 wf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = *(const vec8 *)(sW + offW + t * 2048 + sw);
                xf[t] = *(const vec8 *)(sA + offA + t * 2048 + sw);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T16<DT>::mfma16(wf[i], xf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane holds, for tile (i, j), C[m][n .. n+3] with
    //   m = m0 + wm*64 + j*16 + r,   n = n0 + wn*64 + i*16 + g*4
    if (wide) {
        f32x4 bq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ONEK) bq[i] = pbq[i];
            else bq[i] = HAS_BIAS ? *(const f32x4 *)(p.bias + n0 + wn * 64 + i * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        u32x4 idn[2][HAS_IDN ? 4 : 1];
        if (HAS_IDN) {           // all identity loads are issued before the first store (C and the identity may alias)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ONEK) idn[0][q] = pidn[q];
                else idn[0][q] = *(const u32x4 *)(idn_src + (int64_t)q * 8 * p.ldr);
                idn[1][q] = *(const u32x4 *)(idn_src + (int64_t)(32 + q * 8) * p.ldr);
            }
        }
        char *my = smem + wave * 8192;
        E *dst = (E *)p.C + (int64_t)(m0 + wm * 64 + orr) * p.ldc + n0 + wn * 64 + och * 8;
        if (!HAS_IDN) {
            // no second addend: round first, stage the 16-bit tile in one pass (64 rows of 128 B, chunk ^= row & 7)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = acc[i][j] + bq[i];
                    if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                    }
                    if (EPI == HGR_EPI_BIAS_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    const int row = j * 16 + r;
                    *(typename T16<DT>::vec4 *)(my + row * 128 + (((i * 2 + (g >> 1)) ^ (row & 7)) * 16) + (g & 1) * 8) = cvt4<DT>(v[0], v[1], v[2], v[3]);
                }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = q * 8 + orr;
                *(u32x4 *)(dst + (int64_t)q * 8 * p.ldc) = *(const u32x4 *)(my + row * 128 + ((och ^ (row & 7)) * 16));
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = acc[i][h * 2 + jj] + bq[i];
                    if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                    }
                    const int row = jj * 16 + r;
                    *(f32x4 *)(my + row * 256 + (((i * 4 + g) ^ (row & 15)) * 16)) = v;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = q * 8 + orr;
                const f32x4 lo = *(const f32x4 *)(my + row * 256 + (((2 * och) ^ (row & 15)) * 16));
                const f32x4 hi = *(const f32x4 *)(my + row * 256 + (((2 * och + 1) ^ (row & 15)) * 16));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (HAS_IDN) {
                    const typename T16<DT>::vec8 iv = __builtin_bit_cast(typename T16<DT>::vec8, idn[h][q]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = EPI == HGR_EPI_QGELU_GRAD16 ? v[e] * quick_gelu_grad((float)iv[e]) : v[e] + (float)iv[e];
                }
                if (EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                typename T16<DT>::vec8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (E)v[e];
                *(u32x4 *)(dst + (int64_t)(h * 32 + q * 8) * p.ldc) = __builtin_bit_cast(u32x4, o);
            }
        }
        return;
    }
    if (!ONEK && full) {     // (ONEK keeps only the wide and the guarded epilogue: this one costs it 55 registers = an occupancy step)
        // interior tile: unguarded loads and stores, bias quads fetched once
        f32x4 bq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (epi_has_bias(EPI)) bq[i] = *(const f32x4 *)(p.bias + n0 + wn * 64 + i * 16 + g * 4);
        }
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {       // batches of 8 quads: all addend loads first, then the stores
            f32x4 ad[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) ad[j][i] = load_addend<DT, EPI>(p, m0 + wm * 64 + (jh * 2 + j) * 16 + r, n0 + wn * 64 + i * 16 + g * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    store_quad_full<DT, EPI, OUT32>(p, acc[i][jh * 2 + j], bq[i], ad[j][i], m0 + wm * 64 + (jh * 2 + j) * 16 + r, n0 + wn * 64 + i * 16 + g * 4);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + g * 4;
            if (n < p.N) store_quad<DT, EPI, OUT32>(p, acc[i][j], m, n);
        }
    }
}


// =================================================================================================
// 256 x 256 x 64 tile, 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns 128 (M) x 64 (N) =
// 8 x 4 MFMA tiles = 128 accumulator registers.  128 flop per staged byte (2x the 128^2 kernel), which
// is what the short-K tower GEMMs (K = 768: only 12 K-tiles) need: they are bound by bytes in flight.
//
// A K-tile is computed in 4 phases of 16 MFMAs per wave (one quadrant of the wave's output each):
//     ph1: Q(m-half 0, n-half 0)   reads W n-half 0 (4 x ds_read_b128) and A m-half 0 (8 x)
//     ph2: Q(0, 1)                 reads W n-half 1 (4 x)
//     ph3: Q(1, 1)                 reads A m-half 1 (8 x, same registers as m-half 0)
//     ph4: Q(1, 0)                 reads nothing (W n-half 0 is still in registers)
// The K-tile is staged as 4 LDS-DMA "pieces" of 16 KB cut BY PHASE, not by row range:
//     A0 = the m-half-0 rows of both wave rows (A rows 0-63,128-191)   first read in ph1
//     W0 = the n-half-0 rows of all four wave columns                  first read in ph1
//     W1 = the n-half-1 rows                                           first read in ph2
//     A1 = the m-half-1 rows (A rows 64-127,192-255)                   first read in ph3
// so a piece's LDS slot is free one phase after its first read and can be refilled for K-tile t+2
// immediately: every phase issues exactly one piece (2 x global_load_lds_dwordx4 per thread) about
// 6 phases before it is read, and 5 pieces (80 KB per CU) stay in flight across every wait:
//     (t,1) issues A1(t+1)   (t,2) A0(t+2)   (t,3) W0(t+2)   (t,4) W1(t+2)
// Waits are counted: vmcnt(10) = "all but my 10 youngest DMA instructions landed" = the piece the
// NEXT phase reads has landed; then one raw s_barrier (every wave's share landed / every wave's reads
// of the slot about to be refilled are done).  The last two K-tiles use the exact smaller counts.
// Two LDS buffers x 4 pieces = 128 KB, one workgroup per CU.  (cdna_hip_programming.md section 5:
// "Pipelining across barriers", counted vmcnt + raw s_barrier, all LDS in one array.)
// =================================================================================================
constexpr int NT256 = 512;
constexpr int PIECE = 16384;

// R interval end: my share of the piece the NEXT read interval needs has landed (counted vmcnt), my own
// ds_reads are complete (so the slot they read may be refilled), then the barrier.  M interval end: barrier.
#define HGR_RWAIT(N) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HGR_RBAR() do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HGR_MBAR() do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int DT, int EPI, bool OUT32, bool CONV = false>
__global__ __launch_bounds__(NT256) void gemm_nt_256(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    if (p.kc) {                                   // split-K (see gemm_nt_128)
        const int sp = blockIdx.y;
        p.A += (int64_t)sp * p.kc * 2; p.W += (int64_t)sp * p.kc * 2;
        p.C = (char *)p.C + (int64_t)sp * p.csplit * 4;
        p.K = min(p.kc, p.K - sp * p.kc);
    }
    __shared__ __attribute__((aligned(1024))) char smem[2 * 4 * PIECE + 8 * 2048];   // + epilogue row padding

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;      // waves w and w+4 share a SIMD: wm is also the ping-pong group
    const int r = lane & 15, g = lane >> 4;

    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    constexpr int GROUP = 4;      // 32 tiles in flight per XCD = 4 x 8 panels of 256 x K
    int tm, tn;
    if (p.m_fastest) {
        const int per = GROUP * p.tiles_m, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_n - first), loc = wg - grp * per;
        tn = first + loc % gs; tm = loc / gs;
    } else {
        const int per = GROUP * p.tiles_n, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_m - first), loc = wg - grp * per;
        tm = first + loc % gs; tn = loc / gs;
    }
    const int m0 = tm * 256, n0 = tn * 256;

    // piece kinds in issue order: 0 = A0, 1 = W0, 2 = W1, 3 = A1.  Each thread moves 2 x 16 B per piece.
    const char *src[4][2];
    int cchunk[2];              // CONV: this lane's logical 16-B chunk (8 input channels) inside a K-tile
    unsigned vmask[2][2];       // CONV: [A0 / A1][j]: bit t set = tap t of that output pixel is inside the image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = (j * 8 + wave) * 64 + lane;       // 16-B chunk id inside the 128-row piece
        const int pr = id >> 3, c = (id & 7) ^ (pr & 7); // piece row, swizzled source chunk
        const int ra0 = pr + (pr >= 64 ? 64 : 0);        // A row of piece row (m-half 0 of wave row pr/64)
        const int rw0 = (pr >> 5) * 64 + (pr & 31);      // W row of piece row (n-half 0 of wave col pr/32)
        src[1][j] = p.W + ((int64_t)min(n0 + rw0, p.N - 1) * p.ldw + c * 8) * 2;
        src[2][j] = p.W + ((int64_t)min(n0 + rw0 + 32, p.N - 1) * p.ldw + c * 8) * 2;
        if (CONV) {
            cchunk[j] = c;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int gm = min(m0 + ra0 + half * 64, p.M - 1);
                const int wo = gm % p.cWo, t1 = gm / p.cWo, ho = t1 % p.cHo, b = t1 / p.cHo;
                const int hi0 = ho * p.cStride - 1, wi0 = wo * p.cStride - 1;
                unsigned vm = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hi = hi0 + t / 3, wi = wi0 + t % 3;
                    if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW) vm |= 1u << t;
                }
                vmask[half][j] = vm;
                src[half ? 3 : 0][j] = p.A + (((int64_t)b * p.cH + hi0) * p.cW + wi0) * p.cC * 2;   // tap (0,0), channel 0
            }
        } else {
            src[0][j] = p.A + ((int64_t)min(m0 + ra0, p.M - 1) * p.lda + c * 8) * 2;
            src[3][j] = p.A + ((int64_t)min(m0 + ra0 + 64, p.M - 1) * p.lda + c * 8) * 2;
        }
    }
    const bool do_mma = p.dbg != 1, do_ld = p.dbg != 2;
    auto issue = [&](int kind, int t) {
        if (!do_ld) return;
        char *dst = smem + (t & 1) * (4 * PIECE) + kind * PIECE + wave * 1024;
        const int64_t koff = (int64_t)t * 128;
        if (CONV && (kind == 0 || kind == 3)) {
            // implicit im2col (see gemm_nt_128): K index = tap * C + channel, 8 channels per 16-B chunk, OOB taps -> zero page
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kq = t * 64 + cchunk[j] * 8;
                const int tap = (int)__umulhi((unsigned)kq, p.cMagic), cin = kq - tap * p.cC;
                const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                const bool ok = tap < 9 && ((vmask[kind == 3][j] >> tap) & 1u);
                const char *sp = ok ? src[kind][j] + ((int64_t)(ky * p.cW + kx) * p.cC + cin) * 2 : (const char *)hgr_zero_page;
                __builtin_amdgcn_global_load_lds((const AS1 void *)sp, (AS3 void *)(dst + j * 8192), 16, 0, 0);
            }
            return;
        }
        __builtin_amdgcn_global_load_lds((const AS1 void *)(src[kind][0] + koff), (AS3 void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const AS1 void *)(src[kind][1] + koff), (AS3 void *)(dst + 8192), 16, 0, 0);
    };

    f32x4 acc[2][2][4][2];      // [m-half][n-half][m tile][n tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;    // >= 2 (host guarantees)
    // prologue: K-tile 0 complete, K-tile 1 without A1 (sequence numbers 0..6)
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    issue(0, 1); issue(1, 1); issue(2, 1);
    HGR_RWAIT(10);              // A0(0), W0(0) landed
    if (wm) HGR_MBAR();         // ping-pong: group 1 runs one barrier interval behind group 0

    const int offA = (wm * 64 + r) * 128;      // + m tile * 2048, within pieces A0 / A1
    const int offW = (wn * 32 + r) * 128;      // + n tile * 2048, within pieces W0 / W1
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;

    vec8 af[4][2], wf0[2][2], wf1[2][2];

    // MODE 0: steady state, 1: second-last K-tile (nothing left to issue after ph1), 2: last K-tile.
    // Every phase = R interval (reads, 1 piece issued, wait) | barrier | M interval (16 MFMAs) | barrier.
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char *buf = smem + (t & 1) * (4 * PIECE);
        // ---- ph1: Q(0,0) ----------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf0[j][0] = *(const vec8 *)(buf + 1 * PIECE + offW + j * 2048 + sw0);
            wf0[j][1] = *(const vec8 *)(buf + 1 * PIECE + offW + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(buf + 0 * PIECE + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(buf + 0 * PIECE + offA + i * 2048 + sw1);
        }
        if (MODE <= 1) issue(3, t + 1);
        if (MODE <= 1) HGR_RWAIT(10); else HGR_RWAIT(2);          // W1(t) landed
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[0][0][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph2: Q(0,1) ----------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf1[j][0] = *(const vec8 *)(buf + 2 * PIECE + offW + j * 2048 + sw0);
            wf1[j][1] = *(const vec8 *)(buf + 2 * PIECE + offW + j * 2048 + sw1);
        }
        if (MODE == 0) issue(0, t + 2);
        if (MODE == 0) HGR_RWAIT(10); else if (MODE == 1) HGR_RWAIT(8); else HGR_RWAIT(0);   // A1(t) landed
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[0][1][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph3: Q(1,1) ----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(buf + 3 * PIECE + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(buf + 3 * PIECE + offA + i * 2048 + sw1);
        }
        if (MODE == 0) issue(1, t + 2);
        HGR_RBAR();                                               // ph4 reads nothing new
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[1][1][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph4: Q(1,0) ----------------------------------------------------------------------
        if (MODE == 0) issue(2, t + 2);
        if (MODE == 0) HGR_RWAIT(10); else if (MODE == 1) HGR_RWAIT(4); else HGR_RBAR();     // A0(t+1), W0(t+1) landed
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[1][0][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
    };

    for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
    ktile(nk - 2, std::integral_constant<int, 1>());
    ktile(nk - 1, std::integral_constant<int, 2>());
    if (!wm) HGR_MBAR();
    if (p.dbg == 3) { if (acc[0][0][0][0][0] == 123.456f) ((float *)p.C)[0] = 1.f; return; }
    // group 0 waits for group 1's last interval: every LDS read is done, no DMA in flight

    // tile (a, b, i, j) of this lane holds C[m][n .. n+3],
    //   m = m0 + wm*128 + a*64 + i*16 + r,   n = n0 + wn*64 + b*32 + j*16 + g*4
    if (!OUT32 && p.vec_ok && (p.ldc & 7) == 0 && (EPI <= HGR_EPI_BIAS_QUICKGELU || EPI == HGR_EPI_BIAS_RELU)) {
        // 16-bit output: transpose the wave's 128 x 64 tile through its private LDS slice (rows of 128 B
        // + 16 B pad) and write full 128-byte lines with 16-byte stores (8 lanes per row, 8 rows per
        // instruction) instead of 32-byte fragments of 16 different lines per instruction.
        constexpr int RS = 144;
        char *my = smem + wave * (128 * RS);
        // FULL = the whole 256 x 256 tile is inside C: no guards, so the 4 bias loads and the 16 stores are independent
        // (guarded loads make hipcc wait vmcnt(0) behind every one of them: 4 serial L2 round trips per tile)
        auto staged = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            f32x4 bq[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                bq[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (epi_has_bias(EPI)) {
                    if (FULL || n + 3 < p.N) bq[b][j] = *(const f32x4 *)(p.bias + n);
                    else
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) bq[b][j][e] = p.bias[n + e];
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[a][b][i][j] + bq[b][j];
                if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                }
                if (EPI == HGR_EPI_BIAS_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *(typename T16<DT>::vec4 *)(my + (a * 64 + i * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 2) = cvt4<DT>(v[0], v[1], v[2], v[3]);
            }
            // LDS ops of one wave complete in order, and the slice is private to the wave: no barrier needed
            const int ch = lane & 7, rr = lane >> 3;
            const int nb = n0 + wn * 64 + ch * 8;
            E *dst0 = (E *)p.C + (int64_t)(m0 + wm * 128 + rr) * p.ldc + nb;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = q * 8 + rr;
                const int m = m0 + wm * 128 + row;
                const u32x4 v = *(const u32x4 *)(my + row * RS + ch * 16);
                if (p.dbg == 4) { if (v[0] == 0x12345678u) ((float *)p.C)[0] = 1.f; continue; }
                E *dst = dst0 + (int64_t)q * 8 * p.ldc;
                if (FULL) *(u32x4 *)dst = v;
                else if (m < p.M) {
                    if (nb + 7 < p.N) *(u32x4 *)dst = v;
                    else {
                        const E *ve = (const E *)&v;
                        for (int e = 0; e < 8 && nb + e < p.N; ++e) dst[e] = ve[e];
                    }
                }
            }
        };
        if (m0 + 256 <= p.M && n0 + 256 <= p.N) staged(std::true_type());
        else staged(std::false_type());
        return;
    }
    if (p.vec_ok && m0 + 256 <= p.M && n0 + 256 <= p.N) {
        f32x4 bq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bq[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (epi_has_bias(EPI)) bq[b][j] = *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {       // batches of 8 quads: all addend loads first, then the stores
            f32x4 ad[2][2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ad[i][b][j] = load_addend<DT, EPI>(p, m0 + wm * 128 + a * 64 + (ih * 2 + i) * 16 + r, n0 + wn * 64 + b * 32 + j * 16 + g * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                store_quad_full<DT, EPI, OUT32>(p, acc[a][b][ih * 2 + i][j], bq[b][j], ad[i][b][j],
                                                m0 + wm * 128 + a * 64 + (ih * 2 + i) * 16 + r, n0 + wn * 64 + b * 32 + j * 16 + g * 4);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + a * 64 + i * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
            if (n < p.N) store_quad<DT, EPI, OUT32>(p, acc[a][b][i][j], m, n);
        }
    }
}


// =================================================================================================
// gemm_nt_duo: 256 (M) x 128 (N) x 64 tile, 256 threads = 4 waves as 2 (M) x 2 (N); a wave owns 128 x 64, the register
// tile of gemm_nt_256 (128 accumulators, 4 quadrant phases per K-tile).  TWO workgroups share a CU: 80 KB of LDS and
// <= 256 registers per wave each, one wave of each workgroup per SIMD.
//
// Why: at one workgroup per CU nothing overlaps a tile's prologue (first operand pieces) and epilogue (bias / fp32
// residual loads, stores): ~14 us of a ~35 us tile at K = 768, and a whole round of 256 tiles stores (and, for the
// residual epilogue, re-reads) its 33 - 67 MB in one burst.  A wave's stores and its LDS-DMA loads share one in-order
// vmcnt, so a single persistent workgroup cannot hide them either (finding 3).  Two independent
// workgroups per CU can: while one is in its epilogue or waits for operands, the other one's waves own the matrix
// pipes of the same SIMDs.  No ping-pong groups inside a workgroup, so ONE barrier per phase (4 per K-tile).
//
// LDS: A0 (m-half-0 rows of both wave rows, 16 KB) and A1 (m-half 1) double-buffered, W0 / W1 (n-half 0 / 1 rows of both
// wave columns, 8 KB each) single-buffered: 2 x 32 + 16 = 80 KB.  A piece's slot is refilled in the phase after the
// barrier that follows its last read:
//     ph1 reads W0(t), A0(t)   issues A1(t+1) x4           waits vmcnt(8)  : W1(t) landed
//     ph2 reads W1(t)          issues W0(t+1) x2           waits vmcnt(14) : A1(t) landed
//     ph3 reads A1(t)          issues W1(t+1) x2, A0(t+2) first half x2    (ph4 reads nothing new)
//     ph4                      issues A0(t+2) second half x2   waits vmcnt(6) : W0(t+1), A0(t+1) landed
// (xN = global_load_lds_dwordx4 instructions per thread; the counts are "my N youngest may still be in flight").
// The activations are prefetched 5 - 7 phases ahead, the weights (L2 / MALL resident panels shared by every row panel)
// 3 phases ahead; if a piece is late the partner workgroup's MFMAs fill the gap.
// =================================================================================================
constexpr int NTD = 256;
constexpr int DUO_A0 = 0, DUO_A1 = 32768, DUO_W0 = 65536, DUO_W1 = 73728, DUO_LDS = 81920;

// 16-byte store that does not keep the line in the XCD's L2 (sc1: write-through, line dropped): a tile's output is never
// re-read by this launch, and 64 tiles in flight per XCD write as many bytes as the L2 holds (experiment: HGR_GEMM_DBG=8)
__device__ __forceinline__ void store16_sc1(void *ptr, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

template <int DT, int EPI, bool OUT32, int LN = 0>
__global__ __launch_bounds__(NTD, 2) void gemm_nt_duo(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    if (p.kc) {                                   // split-K (see gemm_nt_128)
        const int sp = blockIdx.y;
        p.A += (int64_t)sp * p.kc * 2; p.W += (int64_t)sp * p.kc * 2;
        p.C = (char *)p.C + (int64_t)sp * p.csplit * 4;
        p.K = min(p.kc, p.K - sp * p.kc);
    }
    __shared__ __attribute__((aligned(1024))) char smem[DUO_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, g = lane >> 4;

    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int GROUP = p.group;    // default 4: 64 tiles in flight per XCD = 4 row panels of 256 x 16 column panels of 128
    int tm, tn;
    if (p.m_fastest) {
        const int per = GROUP * p.tiles_m, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_n - first), loc = wg - grp * per;
        tn = first + loc % gs; tm = loc / gs;
    } else {
        const int per = GROUP * p.tiles_n, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_m - first), loc = wg - grp * per;
        tm = first + loc % gs; tn = loc / gs;
    }
    const int m0 = tm * 256, n0 = tn * 128;

    // per-lane source offsets (bytes from A / W; operands are < 4 GB, checked on the host) of the LDS-DMA instructions of
    // one K-tile: 4 per A piece (32 piece rows each), 2 per W piece.  Piece row pr of A0 = tile row (pr / 64) * 128 + pr % 64,
    // of W0 = tile row (pr / 32) * 64 + pr % 32; A1 / W1 = the same rows + 64 / + 32.  Source chunk ^= row & 7 (rule 21).
    unsigned oA0[4], oA1[4], oW0[2], oW1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = (i * 4 + wave) * 64 + lane;
        const int pr = id >> 3, c = (id & 7) ^ (pr & 7);
        const int ra = (pr >> 6) * 128 + (pr & 63);
        oA0[i] = (unsigned)(((int64_t)min(m0 + ra, p.M - 1) * p.lda + c * 8) * 2);
        oA1[i] = (unsigned)(((int64_t)min(m0 + ra + 64, p.M - 1) * p.lda + c * 8) * 2);
        if (i < 2) {
            const int rw = (pr >> 5) * 64 + (pr & 31);
            oW0[i] = (unsigned)(((int64_t)min(n0 + rw, p.N - 1) * p.ldw + c * 8) * 2);
            oW1[i] = (unsigned)(((int64_t)min(n0 + rw + 32, p.N - 1) * p.ldw + c * 8) * 2);
        }
    }
    char *const ldsw = smem + wave * 1024;
    // halves of an A piece: instructions [2h, 2h + 2)
    auto issueA = [&](const unsigned (&off)[4], int slot_base, int t, int h) {
        const char *base = p.A + (int64_t)t * 128;
        char *dst = ldsw + slot_base + (t & 1) * 16384;
#pragma unroll
        for (int i = 2 * h; i < 2 * h + 2; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void *)(base + off[i]), (AS3 void *)(dst + i * 4096), 16, 0, 0);
    };
    auto issueW = [&](const unsigned (&off)[2], int slot_base, int t) {
        const char *base = p.W + (int64_t)t * 128;
        char *dst = ldsw + slot_base;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void *)(base + off[i]), (AS3 void *)(dst + i * 4096), 16, 0, 0);
    };

    f32x4 acc[2][2][4][2];      // [m-half][n-half][m tile][n tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LN consumer: thread t finalises the statistics of tile row t from the producer's per-slot partial sums.  Called in the
    // epilogue, where its loads travel together with the bias / ln_s / ln_c loads (one exposed round trip per tile, covered
    // by the partner workgroup); at kernel entry it would delay the first LDS-DMA by a memory round trip.
    auto ln_row_stats = [&]() {
        const f32x4 *sp = (const f32x4 *)(p.ln_stats + (int64_t)min(m0 + tid, p.M - 1) * p.ln_slots * 2);
        float s1 = 0.f, s2 = 0.f;
        // the usual row widths get all their loads issued back to back (a load inside a run-time loop is waited for on the spot:
        // six dependent L2 round trips for width 768)
        auto fixed = [&](auto nq_tag) {
            constexpr int NQ = decltype(nq_tag)::value;
            f32x4 t[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) t[i] = sp[i];
#pragma unroll
            for (int i = 0; i < NQ; ++i) { s1 += t[i][0] + t[i][2]; s2 += t[i][1] + t[i][3]; }
        };
        switch (p.ln_slots) {
            case 4: fixed(std::integral_constant<int, 2>()); break;      // width 256
            case 8: fixed(std::integral_constant<int, 4>()); break;      // 512
            case 10: fixed(std::integral_constant<int, 5>()); break;     // 640
            case 12: fixed(std::integral_constant<int, 6>()); break;     // 768
            case 16: fixed(std::integral_constant<int, 8>()); break;     // 1024
            default:
                for (int i = 0; i < p.ln_slots / 2; ++i) { const f32x4 t = sp[i]; s1 += t[0] + t[2]; s2 += t[1] + t[3]; }
        }
        const float inv = 1.0f / (float)p.K;
        const float mean = s1 * inv;
        return make_float2(mean, rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + p.ln_eps));
    };
    const int nk = p.K / 64;    // >= 2 (host guarantees)
    // prologue in steady-state order: A0(0), A1(0), W0(0), W1(0), A0(1)
    issueA(oA0, DUO_A0, 0, 0); issueA(oA0, DUO_A0, 0, 1);
    issueA(oA1, DUO_A1, 0, 0); issueA(oA1, DUO_A1, 0, 1);
    issueW(oW0, DUO_W0, 0); issueW(oW1, DUO_W1, 0);
    issueA(oA0, DUO_A0, 1, 0); issueA(oA0, DUO_A0, 1, 1);
    HGR_RWAIT(6);               // A0(0), W0(0) landed

    const int offA = (wm * 64 + r) * 128;      // + m tile * 2048, within pieces A0 / A1
    const int offW = (wn * 32 + r) * 128;      // + n tile * 2048, within pieces W0 / W1
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
    vec8 af[4][2], wf0[2][2], wf1[2][2];

    // MODE 0: steady state (t + 2 < nk), 1: second-last K-tile, 2: last K-tile
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char *bufA0 = smem + DUO_A0 + (t & 1) * 16384, *bufA1 = smem + DUO_A1 + (t & 1) * 16384;
        // ---- ph1: Q(0,0) ----
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf0[j][0] = *(const vec8 *)(smem + DUO_W0 + offW + j * 2048 + sw0);
            wf0[j][1] = *(const vec8 *)(smem + DUO_W0 + offW + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(bufA0 + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(bufA0 + offA + i * 2048 + sw1);
        }
        if (MODE <= 1) { issueA(oA1, DUO_A1, t + 1, 0); issueA(oA1, DUO_A1, t + 1, 1); }
        if (MODE <= 1) HGR_RWAIT(8); else HGR_RWAIT(0);                 // W1(t) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[0][0][i][j]);
        __builtin_amdgcn_s_setprio(0);
        // ---- ph2: Q(0,1) ----
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf1[j][0] = *(const vec8 *)(smem + DUO_W1 + offW + j * 2048 + sw0);
            wf1[j][1] = *(const vec8 *)(smem + DUO_W1 + offW + j * 2048 + sw1);
        }
        if (MODE <= 1) issueW(oW0, DUO_W0, t + 1);
        if (MODE <= 1) HGR_RWAIT(14); else HGR_RBAR();                  // A1(t) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[0][1][i][j]);
        __builtin_amdgcn_s_setprio(0);
        // ---- ph3: Q(1,1) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(bufA1 + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(bufA1 + offA + i * 2048 + sw1);
        }
        if (MODE <= 1) issueW(oW1, DUO_W1, t + 1);
        if (MODE == 0) issueA(oA0, DUO_A0, t + 2, 0);
        __builtin_amdgcn_sched_barrier(0);                              // no barrier here: nothing is refilled before ph4's barrier that ph3 reads
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[1][1][i][j]);
        __builtin_amdgcn_s_setprio(0);
        // ---- ph4: Q(1,0) ----
        if (MODE == 0) issueA(oA0, DUO_A0, t + 2, 1);
        if (MODE == 0) HGR_RWAIT(6); else if (MODE == 1) HGR_RWAIT(2);  // W0(t+1), A0(t+1) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[1][0][i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
    ktile(nk - 2, std::integral_constant<int, 1>());
    ktile(nk - 1, std::integral_constant<int, 2>());
    HGR_MBAR();                 // every wave's LDS reads are done, no DMA in flight: the staging area is free
    // the epilogue is VALU / LDS work next to the partner workgroup's MFMA clusters (priority 1): run it above them, or its
    // instructions only get the issue slots the matrix stream leaves over (HGR_GEMM_DBG bit 16 = off, for A/B runs)
    if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(3);

    // tile (a, b, i, j) of this lane holds C[m][n .. n+3],  m = m0 + wm*128 + a*64 + i*16 + r,  n = n0 + wn*64 + b*32 + j*16 + g*4
    if (LN == 3) {
        // Evaluation consumers in place of the C store (main.py:136-176 consuming model/clip_tree.py:331).  This wave's 128 rows x
        // 64 columns are slice s = n0 / 64 + wn of every row; a lane holds, per row, 16 of the 64 columns (4 quads), the other
        // 48 sit in the lanes r + 16, r + 32, r + 48.  key = (orderable(value) << 32) | (0x7fffffff - train position): unsigned
        // max = "larger value, then smaller position" (the tie rule of logits[:, train_index].topk); 0 = no train column.
        const int sl = (n0 >> 6) + wn;
        int tp[2][2][4], ep[2][2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
            const int4 t4 = *(const int4 *)(p.ev_tpos + n), e4 = *(const int4 *)(p.ev_epos + n);
            tp[b][j][0] = t4.x; tp[b][j][1] = t4.y; tp[b][j][2] = t4.z; tp[b][j][3] = t4.w;
            ep[b][j][0] = e4.x; ep[b][j][1] = e4.y; ep[b][j][2] = e4.z; ep[b][j][3] = e4.w;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned long long key = 0ull;
            // per 16-column group (b, j) of the slice: largest test value, the test position of one element attaining it, and
            // the second largest value (multiplicity counted: m2 == m1 when the maximum is attained twice)
            f32x4 m1, m2;
            int p1[4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[a][b][i][j][e] + 0.0f;             // -0 -> +0, as the row sweep of hgr_eval_rows does
                    const unsigned u = __float_as_uint(v);
                    const unsigned long long k2 = ((unsigned long long)(u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u)) << 32) | (unsigned)(0x7fffffff - tp[b][j][e]);
                    if (tp[b][j][e] >= 0 && k2 > key) key = k2;
                    x[e] = ep[b][j][e] >= 0 ? v : -INFINITY;
                }
                const float hi01 = fmaxf(x[0], x[1]), lo01 = fminf(x[0], x[1]), hi23 = fmaxf(x[2], x[3]), lo23 = fminf(x[2], x[3]);
                const float top = fmaxf(hi01, hi23);
                m1[b * 2 + j] = top;
                m2[b * 2 + j] = fmaxf(fminf(hi01, hi23), fmaxf(lo01, lo23));
                p1[b * 2 + j] = x[0] == top ? ep[b][j][0] : x[1] == top ? ep[b][j][1] : x[2] == top ? ep[b][j][2] : ep[b][j][3];
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const unsigned hi = __shfl_xor((unsigned)(key >> 32), o), lo = __shfl_xor((unsigned)key, o);
                const unsigned long long x = ((unsigned long long)hi << 32) | lo;
                key = x > key ? x : key;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float o1 = __shfl_xor(m1[q], o), o2 = __shfl_xor(m2[q], o);
                    const int op = __shfl_xor(p1[q], o);
                    m2[q] = fmaxf(fminf(m1[q], o1), fmaxf(m2[q], o2));
                    p1[q] = o1 > m1[q] ? op : p1[q];
                    m1[q] = fmaxf(m1[q], o1);
                }
            }
            const int m = m0 + wm * 128 + a * 64 + i * 16 + r;
            if (g == 0 && m < p.M) {
                const int64_t at = (int64_t)m * p.ev_slices + sl;
                p.ev_key[at] = key;
                *(f32x4 *)(p.ev_tmax + at * 4) = m1;
                *(int4 *)(p.ev_p1 + at * 4) = make_int4(p1[0], p1[1], p1[2], p1[3]);
                *(f32x4 *)(p.ev_m2 + at * 4) = m2;
            }
        }
        return;
    }
    constexpr bool HAS_BIAS = epi_has_bias(EPI);
    constexpr bool HAS_ADD = EPI == HGR_EPI_BIAS_RESIDUAL || EPI == HGR_EPI_ACCUM;
    const bool full = p.vec_ok && m0 + 256 <= p.M && n0 + 128 <= p.N;
    if (full && !OUT32 && (p.ldc & 7) == 0 && (EPI <= HGR_EPI_BIAS_QUICKGELU || EPI == HGR_EPI_BIAS_RELU)) {
        // 16-bit output: the wave's 128 x 64 tile through its private LDS slice (rows of 128 B + 16 B pad), then full
        // 128-byte lines with 16-byte stores (8 lanes per row, 8 rows per instruction)
        constexpr int RS = 144;
        char *my = smem + wave * (128 * RS);
        f32x4 bq[2][2], lsq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                if (LN == 2) { bq[b][j] = *(const f32x4 *)(p.ln_c + n); lsq[b][j] = *(const f32x4 *)(p.ln_s + n); }
                else bq[b][j] = HAS_BIAS ? *(const f32x4 *)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        float2 *lnrow = (float2 *)(smem + 4 * 128 * RS);      // 256 x (mean, rstd) behind the four staging slices
        if (LN == 2) {
            lnrow[tid] = ln_row_stats();
            __syncthreads();
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
        float2 mr = make_float2(0.f, 1.f);
        if (LN == 2) mr = lnrow[wm * 128 + a * 64 + i * 16 + r];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 v;
            if (LN == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(mr.y, acc[a][b][i][j][e] - mr.x * lsq[b][j][e], bq[b][j][e]);
            } else v = acc[a][b][i][j] + bq[b][j];
            if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            if (EPI == HGR_EPI_BIAS_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *(typename T16<DT>::vec4 *)(my + (a * 64 + i * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 2) = cvt4<DT>(v[0], v[1], v[2], v[3]);
        }
        }
        // addresses = wave-uniform 64-bit base + 32-bit per-lane byte offset (one VALU add per store; a 64-bit row * ldc product
        // per access costs ~6 VALU instructions, and this epilogue competes with the partner workgroup for issue slots)
        const int ch = lane & 7, rr = lane >> 3;
        char *cw = (char *)p.C + ((int64_t)(m0 + wm * 128) * p.ldc + n0 + wn * 64) * 2;
        const unsigned ldcB = (unsigned)p.ldc * 2u;
        const unsigned cl = (unsigned)rr * ldcB + ch * 16;
        if (p.dbg & 8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) store16_sc1(cw + (cl + q * 8 * ldcB), *(const u32x4 *)(my + (q * 8 + rr) * RS + ch * 16));
            return;
        }
        if (LN == 4) {
            // training forward of the MLP: C keeps the pre-activation (backward needs it), ln_xh gets QuickGELU of the ROUNDED
            // pre-activation - the bits hgr_quickgelu16 would produce from C in a second pass, without that pass
            char *gw = (char *)p.ln_xh + ((int64_t)(m0 + wm * 128) * p.ln_ldx + n0 + wn * 64) * 2;
            const unsigned ldgB = (unsigned)p.ln_ldx * 2u;
            const unsigned gl = (unsigned)rr * ldgB + ch * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u32x4 v = *(const u32x4 *)(my + (q * 8 + rr) * RS + ch * 16);
                *(u32x4 *)(cw + (cl + q * 8 * ldcB)) = v;
                const vec8 h = __builtin_bit_cast(vec8, v);
                vec8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (E)quick_gelu_train((float)h[e]);
                *(u32x4 *)(gw + (gl + q * 8 * ldgB)) = __builtin_bit_cast(u32x4, o);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            *(u32x4 *)(cw + (cl + q * 8 * ldcB)) = *(const u32x4 *)(my + (q * 8 + rr) * RS + ch * 16);
        return;
    }
    if (full && !OUT32 && epi_has_idn16(EPI) && (p.ldc & 7) == 0 && (p.ldr & 7) == 0) {
        // relu(acc + bias + 16-bit identity) -> 16 bit (bn3(conv3) ; out += identity ; relu, clip/model.py:46-52): the sum is
        // formed in fp32 and rounded once.  4 passes of 32 rows through the wave's LDS slice (fp32 rows of 256 B + 16 B pad);
        // on the way out a lane owns 8 consecutive columns of a row, so the identity is LOADED and the result STORED as
        // 16 bytes per lane over whole 128-byte lines (8 lanes per row, 8 rows per instruction).
        constexpr int RS = 272;
        char *my = smem + wave * (32 * RS);
        f32x4 bq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[b][j] = HAS_BIAS ? *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int r8 = lane >> 3, c8 = lane & 7;
        const int64_t wrow = m0 + wm * 128, wcol = n0 + wn * 64;
        char *cw = (char *)p.C + (wrow * p.ldc + wcol) * 2;
        const char *iw = (const char *)p.res + (wrow * p.ldr + wcol) * 2;
        const unsigned ldcB = (unsigned)p.ldc * 2u, ldiB = (unsigned)p.ldr * 2u;
        const unsigned cl = r8 * ldcB + c8 * 16, il = r8 * ldiB + c8 * 16;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int rl = a * 64 + ih * 32;
            u32x4 idn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) idn[q] = *(const u32x4 *)(iw + (il + (rl + q * 8) * ldiB));
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) = acc[a][b][ih * 2 + i2][j] + bq[b][j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 lo = *(const f32x4 *)(my + (q * 8 + r8) * RS + c8 * 32);
                const f32x4 hi = *(const f32x4 *)(my + (q * 8 + r8) * RS + c8 * 32 + 16);
                const typename T16<DT>::vec8 iv = __builtin_bit_cast(typename T16<DT>::vec8, idn[q]);
                typename T16<DT>::vec8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (EPI == HGR_EPI_QGELU_GRAD16) {          // dL/dpre = dL/dpost * g'(pre): the hgr_quickgelu16 backward, one rounding
                        o[e] = (E)(lo[e] * quick_gelu_grad((float)iv[e]));
                        o[e + 4] = (E)(hi[e] * quick_gelu_grad((float)iv[e + 4]));
                    } else {
                        o[e] = (E)fmaxf(lo[e] + (float)iv[e], 0.f);
                        o[e + 4] = (E)fmaxf(hi[e] + (float)iv[e + 4], 0.f);
                    }
                }
                *(u32x4 *)(cw + (cl + (rl + q * 8) * ldcB)) = __builtin_bit_cast(u32x4, o);
            }
        }
        return;
    }
    if (full && OUT32 && (EPI == HGR_EPI_NONE || EPI == HGR_EPI_BIAS || HAS_ADD) &&
        (LN == 1 || ((p.ldc & 3) == 0 && (!HAS_ADD || EPI == HGR_EPI_ACCUM || (p.ldr & 3) == 0)))) {
        // fp32 output (+ fp32 residual / old C): 4 passes of 32 rows through the wave's private LDS slice (rows of 256 B +
        // 16 B pad); every global access is then 16 bytes per lane over whole 256-byte row segments (2 full lines per row,
        // 4 rows per instruction) instead of 64-byte fragments of 16 rows.  The pass's 8 addend loads are issued before its
        // LDS round trip; residual and C may alias: a pass loads before it stores, and passes touch disjoint rows.
        constexpr int RS = 272;
        char *my = smem + wave * (32 * RS);
        f32x4 bq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bq[b][j] = HAS_BIAS ? *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int rq = lane >> 4, cq = lane & 15;           // row-in-group and 16-byte column chunk of this lane on the way out
        // every global address below = wave-uniform 64-bit base + 32-bit per-lane byte offset (see the 16-bit epilogue)
        const int64_t wrow = m0 + wm * 128, wcol = n0 + wn * 64;
        if (LN == 1) {
            // Producer of a folded LayerNorm.  The residual stream lives in memory as a 16-bit pair: x = hi + lo, hi = x rounded to
            // the MFMA type (= the A operand of the next GEMM, no second copy of the stream), lo = f16(x - hi): |x - hi - lo| <=
            // 2^-11 |x - hi| (2^-22 |x| with f16 hi, 2^-19 |x| with bf16 hi), far below the 16-bit rounding of every GEMM input.
            // Same bytes as an fp32 read-modify-write.  Plus this wave's 64-column share of the rows' LayerNorm statistics; the
            // 16 reduction chains of a pass (8 row groups x {sum, sum of squares}) advance stage by stage (DPP latencies overlap).
            typedef typename T16<DT>::vec4 hvec4;
            char *hw = (char *)p.ln_xh + (wrow * p.ln_ldx + wcol) * 2;
            char *lw = (char *)p.ln_xl + (wrow * p.ln_ldx + wcol) * 2;
            char *sw = (char *)(p.ln_stats + (wrow * p.ln_slots + (n0 >> 6) + wn) * 2);
            const unsigned ldxB = (unsigned)p.ln_ldx * 2u, ldsB = (unsigned)p.ln_slots * 8u;
            const unsigned xl = rq * ldxB + cq * 8, sl = rq * ldsB;
            // the old pair of pass P + 1 is requested before pass P stores (passes touch disjoint rows; the compiler cannot hoist
            // the loads itself, the pointers alias): one exposed memory round trip per tile instead of four
            hvec4 ohb[2][8];
            f16x4 olb[2][8];
            auto pair_load = [&](int buf, int rl) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    ohb[buf][q] = *(const hvec4 *)(hw + (xl + (rl + q * 4) * ldxB));
                    olb[buf][q] = *(const f16x4 *)(lw + (xl + (rl + q * 4) * ldxB));
                }
            };
            pair_load(0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                const int rl = a * 64 + ih * 32;
                const int pb = ih;                               // pass a * 2 + ih uses buffer ih
                hvec4 (&oh)[8] = ohb[pb];
                f16x4 (&ol)[8] = olb[pb];
                if (a * 2 + ih > 0 && (p.dbg & 64)) pair_load(pb, rl);        // HGR_GEMM_DBG bit 64: every pass loads for itself (A/B runs)
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) = acc[a][b][ih * 2 + i2][j] + bq[b][j];
                if (a * 2 + ih < 3 && !(p.dbg & 64)) pair_load(pb ^ 1, rl + 32);
                f32x4 vq[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    vq[q] = *(const f32x4 *)(my + (q * 4 + rq) * RS + cq * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) vq[q][e] += (float)oh[q][e] + (float)ol[q][e];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const hvec4 nh = cvt4<DT>(vq[q][0], vq[q][1], vq[q][2], vq[q][3]);
                    f16x4 nl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) nl[e] = (_Float16)(vq[q][e] - (float)nh[e]);
                    *(hvec4 *)(hw + (xl + (rl + q * 4) * ldxB)) = nh;
                    *(f16x4 *)(lw + (xl + (rl + q * 4) * ldxB)) = nl;
                }
                if (p.dbg & 32) continue;
                float s1[8], s2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    s1[q] = (vq[q][0] + vq[q][1]) + (vq[q][2] + vq[q][3]);
                    s2[q] = (vq[q][0] * vq[q][0] + vq[q][1] * vq[q][1]) + (vq[q][2] * vq[q][2] + vq[q][3] * vq[q][3]);
                }
#define HGR_DPP_STAGE(CTRL) _Pragma("unroll") for (int q = 0; q < 8; ++q) { \
                    s1[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[q]), CTRL, 0xF, 0xF, true)); \
                    s2[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2[q]), CTRL, 0xF, 0xF, true)); }
                HGR_DPP_STAGE(0xB1) HGR_DPP_STAGE(0x4E) HGR_DPP_STAGE(0x141) HGR_DPP_STAGE(0x140)
#undef HGR_DPP_STAGE
                if (cq == 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) *(float2 *)(sw + (sl + (rl + q * 4) * ldsB)) = make_float2(s1[q], s2[q]);
                }
            }
            return;
        }
        const float *addp = EPI == HGR_EPI_ACCUM ? (const float *)p.C : p.res;
        const int64_t ldadd = EPI == HGR_EPI_ACCUM ? p.ldc : p.ldr;
        char *cw = (char *)p.C + (wrow * p.ldc + wcol) * 4;
        const char *aw = (const char *)addp + (wrow * ldadd + wcol) * 4;
        const unsigned ldcB = (unsigned)p.ldc * 4u, ldaB = (unsigned)ldadd * 4u;
        const unsigned cl = rq * ldcB + cq * 16, al = rq * ldaB + cq * 16;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int rl = a * 64 + ih * 32;                 // first row of the pass inside the wave's 128 rows
            f32x4 ad[8];
            if (HAS_ADD) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ad[q] = *(const f32x4 *)(aw + (al + (rl + q * 4) * ldaB));
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) = acc[a][b][ih * 2 + i2][j] + bq[b][j];
            f32x4 vq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                vq[q] = *(const f32x4 *)(my + (q * 4 + rq) * RS + cq * 16);
                if (HAS_ADD) vq[q] += ad[q];
            }
            if (p.dbg & 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) store16_sc1(cw + (cl + (rl + q * 4) * ldcB), __builtin_bit_cast(u32x4, vq[q]));
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) *(f32x4 *)(cw + (cl + (rl + q * 4) * ldcB)) = vq[q];
            }
        }
        return;
    }
    if (LN == 2) {
        // edge tile of an LN consumer (rows beyond M; N is a multiple of 128 by the host's contract): same arithmetic, guarded rows
        float2 *lnrow = (float2 *)smem;
        lnrow[tid] = ln_row_stats();
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + a * 64 + i * 16 + r;
            const float2 mr = lnrow[wm * 128 + a * 64 + i * 16 + r];
            if (m >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                const f32x4 sq = *(const f32x4 *)(p.ln_s + n), cq4 = *(const f32x4 *)(p.ln_c + n);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(mr.y, acc[a][b][i][j][e] - mr.x * sq[e], cq4[e]);
                    if (EPI == HGR_EPI_BIAS_QUICKGELU) v[e] = quick_gelu(v[e]);
                }
                store_quad<DT, HGR_EPI_NONE, OUT32>(p, v, m, n);
            }
        }
        return;
    }
    if (LN == 1) {
        // edge tile of an LN producer: the same pass structure with guarded rows (a partial last row panel)
        typedef typename T16<DT>::vec4 hvec4;
        constexpr int RS = 272;
        char *my = smem + wave * (32 * RS);
        const int rq = lane >> 4, cq = lane & 15;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int row0 = m0 + wm * 128 + a * 64 + ih * 32;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) =
                    acc[a][b][ih * 2 + i2][j] + *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t row = row0 + q * 4 + rq;
                f32x4 v = *(const f32x4 *)(my + (q * 4 + rq) * RS + cq * 16);
                const bool ok = row < p.M;
                const int64_t at = row * p.ln_ldx + n0 + wn * 64 + cq * 4;
                if (ok) {
                    const hvec4 oh = *(const hvec4 *)((const E *)p.ln_xh + at);
                    const f16x4 ol = *(const f16x4 *)((const _Float16 *)p.ln_xl + at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)oh[e] + (float)ol[e];
                }
                const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
                const float s2 = row16_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                if (ok) {
                    const hvec4 nh = cvt4<DT>(v[0], v[1], v[2], v[3]);
                    f16x4 nl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) nl[e] = (_Float16)(v[e] - (float)nh[e]);
                    *(hvec4 *)((E *)p.ln_xh + at) = nh;
                    *(f16x4 *)((_Float16 *)p.ln_xl + at) = nl;
                    if (cq == 0) *(float2 *)(p.ln_stats + (row * p.ln_slots + (n0 >> 6) + wn) * 2) = make_float2(s1, s2);
                }
            }
        }
        return;
    }
    if (LN == 4) {
        // edge tile of the dual-output forward (rows beyond M; N is a multiple of 128 by the host's contract)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + a * 64 + i * 16 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                const f32x4 v = acc[a][b][i][j] + *(const f32x4 *)(p.bias + n);
                const typename T16<DT>::vec4 pre = cvt4<DT>(v[0], v[1], v[2], v[3]);
                *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = pre;
                *(typename T16<DT>::vec4 *)((E *)p.ln_xh + (int64_t)m * p.ln_ldx + n) =
                    cvt4<DT>(quick_gelu_train((float)pre[0]), quick_gelu_train((float)pre[1]), quick_gelu_train((float)pre[2]), quick_gelu_train((float)pre[3]));
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + a * 64 + i * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
            if (n < p.N) store_quad<DT, EPI, OUT32>(p, acc[a][b][i][j], m, n);
        }
    }
}

template <int DT, int EPI>
void launch_epi(const GemmArgs &a, bool out32, dim3 grid, hipStream_t s, bool big) {
    if constexpr (EPI == HGR_EPI_BIAS || EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) {
        if (!big && !out32 && a.K == BK) {
            hipLaunchKernelGGL((gemm_nt_128<DT, EPI, false, false, false, true>), grid, dim3(NT), 0, s, a);
            return;
        }
    }
    if (big) {
        if (out32) hipLaunchKernelGGL((gemm_nt_256<DT, EPI, true>), grid, dim3(NT256), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_256<DT, EPI, false>), grid, dim3(NT256), 0, s, a);
    } else {
        if (out32) hipLaunchKernelGGL((gemm_nt_128<DT, EPI, true>), grid, dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_128<DT, EPI, false>), grid, dim3(NT), 0, s, a);
    }
}

template <int DT>
void launch_dt(const GemmArgs &a, int epi, bool out32, dim3 grid, hipStream_t s, bool big) {
    switch (epi) {
        case HGR_EPI_NONE: launch_epi<DT, HGR_EPI_NONE>(a, out32, grid, s, big); break;
        case HGR_EPI_BIAS: launch_epi<DT, HGR_EPI_BIAS>(a, out32, grid, s, big); break;
        case HGR_EPI_BIAS_QUICKGELU: launch_epi<DT, HGR_EPI_BIAS_QUICKGELU>(a, out32, grid, s, big); break;
        case HGR_EPI_BIAS_RELU: launch_epi<DT, HGR_EPI_BIAS_RELU>(a, out32, grid, s, big); break;
        case HGR_EPI_BIAS_ADD16_RELU: launch_epi<DT, HGR_EPI_BIAS_ADD16_RELU>(a, out32, grid, s, big); break;
        case HGR_EPI_ACCUM: launch_epi<DT, HGR_EPI_ACCUM>(a, out32, grid, s, big); break;
        case HGR_EPI_QGELU_GRAD16:            // 16-bit output only (checked by the host entry)
            if (big) hipLaunchKernelGGL((gemm_nt_256<DT, HGR_EPI_QGELU_GRAD16, false>), grid, dim3(NT256), 0, s, a);
            else hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_QGELU_GRAD16, false>), grid, dim3(NT), 0, s, a);
            break;
        default: launch_epi<DT, HGR_EPI_BIAS_RESIDUAL>(a, out32, grid, s, big); break;
    }
}

template <int DT>
void launch_duo(const GemmArgs &a, int epi, bool out32, dim3 grid, hipStream_t s) {
#define HGR_DUO(E) do { if (out32) hipLaunchKernelGGL((gemm_nt_duo<DT, E, true>), grid, dim3(NTD), 0, s, a); \
                        else hipLaunchKernelGGL((gemm_nt_duo<DT, E, false>), grid, dim3(NTD), 0, s, a); } while (0)
    switch (epi) {
        case HGR_EPI_NONE: HGR_DUO(HGR_EPI_NONE); break;
        case HGR_EPI_BIAS: HGR_DUO(HGR_EPI_BIAS); break;
        case HGR_EPI_BIAS_QUICKGELU: HGR_DUO(HGR_EPI_BIAS_QUICKGELU); break;
        case HGR_EPI_BIAS_RELU: HGR_DUO(HGR_EPI_BIAS_RELU); break;
        case HGR_EPI_BIAS_ADD16_RELU: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_ADD16_RELU, false>), grid, dim3(NTD), 0, s, a); break;
        case HGR_EPI_ACCUM: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_ACCUM, true>), grid, dim3(NTD), 0, s, a); break;
        case HGR_EPI_QGELU_GRAD16: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_QGELU_GRAD16, false>), grid, dim3(NTD), 0, s, a); break;
        default: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_RESIDUAL, true>), grid, dim3(NTD), 0, s, a); break;
    }
#undef HGR_DUO
}

// development knobs of gemm_nt_duo, read once: HGR_GEMM_GROUP (raster group, default 4), HGR_GEMM_DBG (bit 8: sc1 output stores)
int duo_group() {
    static int g = -1;
    if (g < 0) { const char *e = getenv("HGR_GEMM_GROUP"); g = e ? atoi(e) : 4; if (g < 1) g = 4; }
    return g;
}
int duo_dbg() {
    static int d = -1;
    if (d < 0) { const char *e = getenv("HGR_GEMM_DBG"); d = e ? atoi(e) : 0; }
    return d;
}

// tile plan override (hgr_gemm_set_tile); HGR_GEMM_TILE=128|256|2 sets the initial value
int g_force_tile = -1;
int hgr_gemm_force_tile() {
    if (g_force_tile < 0) { const char *e = getenv("HGR_GEMM_TILE"); g_force_tile = e ? atoi(e) : 0; }
    return g_force_tile;
}

}  // namespace

extern "C" int hgr_gemm_set_tile(int tile) {
    HGR_REQUIRE(tile == 0 || tile == 128 || tile == 256 || tile == 2, "hgr_gemm_set_tile: tile must be 0, 128, 256 or 2 (256 x 128 tiles, two workgroups per CU), got %d", tile);
    const int prev = hgr_gemm_force_tile();
    g_force_tile = tile;
    return prev;
}

extern "C" int hgr_gemm_nt(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                           const float *bias, const void *residual, int64_t ldr,
                           int M, int N, int K, int dtype, int epilogue, int out_f32, void *stream) {
    HGR_REQUIRE(A && W && C, "hgr_gemm_nt: null operand");
    HGR_REQUIRE(M >= 1 && N >= 1 && K >= BK, "hgr_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    HGR_REQUIRE(K % BK == 0, "hgr_gemm_nt: K=%d must be a multiple of %d (pad the operands)", K, BK);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "hgr_gemm_nt: lda=%lld ldw=%lld must be >= K and multiples of 8", (long long)lda, (long long)ldw);
    HGR_REQUIRE(hgr_aligned(A, 16) && hgr_aligned(W, 16), "hgr_gemm_nt: A and W must be 16-byte aligned");
    HGR_REQUIRE(ldc >= N, "hgr_gemm_nt: ldc=%lld < N=%d", (long long)ldc, N);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt: bad dtype %d", dtype);
    HGR_REQUIRE(epilogue >= HGR_EPI_NONE && epilogue <= HGR_EPI_QGELU_GRAD16, "hgr_gemm_nt: bad epilogue %d", epilogue);
    HGR_REQUIRE(epilogue != HGR_EPI_ACCUM || out_f32, "hgr_gemm_nt: ACCUM accumulates into an fp32 C");
    HGR_REQUIRE(!epi_has_bias(epilogue) || bias, "hgr_gemm_nt: epilogue %d needs bias", epilogue);
    HGR_REQUIRE((epilogue != HGR_EPI_BIAS_RESIDUAL && !epi_has_idn16(epilogue)) || (residual && ldr >= N), "hgr_gemm_nt: epilogue %d needs its second operand (residual / identity / pre-activation) with ldr >= N", epilogue);
    HGR_REQUIRE(!epi_has_idn16(epilogue) || !out_f32, "hgr_gemm_nt: ADD16_RELU / QGELU_GRAD16 write 16-bit output");
    HGR_REQUIRE(hgr_aligned(C, out_f32 ? 4 : 2), "hgr_gemm_nt: C misaligned");

    bool vec = (ldc % 4 == 0) && hgr_aligned(C, out_f32 ? 16 : 8);
    if (epi_has_bias(epilogue)) vec = vec && hgr_aligned(bias, 16);
    if (epilogue == HGR_EPI_BIAS_RESIDUAL) vec = vec && (ldr % 4 == 0) && hgr_aligned(residual, 16);
    if (epi_has_idn16(epilogue)) vec = vec && (ldr % 4 == 0) && hgr_aligned(residual, 8);
    static int dbg = -1, split_env = -1;
    if (dbg < 0) { const char *e = getenv("HGR_GEMM_DBG"); dbg = e ? atoi(e) : 0; }
    if (split_env < 0) { const char *e = getenv("HGR_GEMM_SPLIT"); split_env = e ? atoi(e) : 1; }
    hipStream_t s = (hipStream_t)stream;
    const size_t csz = out_f32 ? 4 : 2, rsz = epi_has_idn16(epilogue) ? 2 : 4;

    // launch rows [m_lo, m_lo + m_cnt) with one tile size
    auto launch = [&](int m_lo, int m_cnt, bool big) {
        GemmArgs a;
        a.A = (const char *)A + (size_t)m_lo * lda * 2; a.lda = lda; a.W = (const char *)W; a.ldw = ldw;
        a.C = (char *)C + (size_t)m_lo * ldc * csz; a.ldc = ldc; a.bias = bias;
        a.res = residual ? (const float *)((const char *)residual + (size_t)m_lo * ldr * rsz) : nullptr; a.ldr = ldr;
        a.M = m_cnt; a.N = N; a.K = K;
        const int T = big ? 256 : 128;
        a.tiles_m = (m_cnt + T - 1) / T;
        a.tiles_n = (N + T - 1) / T;
        // each XCD owns a contiguous range of tile ids; the operand indexed by the slow tile index is fetched ~once,
        // the other one once per XCD.  Make the bigger operand the once-fetched one.
        a.m_fastest = ((int64_t)N * K > (int64_t)m_cnt * K) ? 1 : 0;
        a.vec_ok = vec ? 1 : 0;
        a.dbg = dbg; a.kc = 0; a.csplit = 0;
        // outputs at most 64 wide (1x1 convolutions into the 64-channel ResNet stages): the 256 x 64 arrangement of the small kernel
        if (!big && N <= 64 && m_cnt >= 1024 && epilogue == HGR_EPI_BIAS_RELU && !out_f32) {
            a.tiles_m = (m_cnt + 255) / 256; a.tiles_n = 1;
            dim3 gt((unsigned)a.tiles_m);
            if (K == BK) {
                if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_BIAS_RELU, false, false, true, true>), gt, dim3(NT), 0, s, a);
                else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_BIAS_RELU, false, false, true, true>), gt, dim3(NT), 0, s, a);
                return;
            }
            if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_BIAS_RELU, false, false, true>), gt, dim3(NT), 0, s, a);
            else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_BIAS_RELU, false, false, true>), gt, dim3(NT), 0, s, a);
            return;
        }
        dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
        if (dtype == HGR_BF16) launch_dt<HGR_BF16>(a, epilogue, out_f32 != 0, grid, s, big);
        else launch_dt<HGR_F16>(a, epilogue, out_f32 != 0, grid, s, big);
    };

    // Tile choice.  The 256^2 deep-pipelined kernel owns a CU (one 512-thread workgroup), so a launch runs in rounds of
    // 256 workgroups; 128^2 tiles run 2 workgroups per CU.  Three plans are priced with measured tile times: all big,
    // all small, or full rounds of big tiles on the first row panels + the remaining panels on the small-tile kernel in a
    // second launch (pays off at long K, where a mostly empty last big round is expensive: c_proj / patch GEMM).
    const int force = hgr_gemm_force_tile();
    const int tn256 = (N + 255) / 256, tm256 = (M + 255) / 256;
    const int64_t t256 = (int64_t)tm256 * tn256;
    const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    // measured tile times on MI355X (us): 256^2 tile ~ 1.75 per K-tile + 14 (prologue + epilogue, nothing overlaps them
    // at one workgroup per CU); 128^2 tile at 2 per CU ~ 1.08 per K-tile + 9
    const double Tb = 1.75 * (K / 64) + 14.0, Ts = 1.08 * (K / 64) + 9.0;
    const double cost_small = (double)((t128 + 511) / 512) * Ts;
    const double cost_big = (double)((t256 + 255) / 256) * Tb;
    const int64_t rounds = t256 / 256;
    const int big_panels = (int)((rounds * 256) / tn256);
    const int m1 = big_panels * 256;
    double cost_split = 1e30;
    if (split_env && rounds >= 1 && m1 > 0 && m1 < M) {
        const int64_t ts = (int64_t)((M - m1 + 127) / 128) * ((N + 127) / 128);
        cost_split = (double)rounds * Tb + (double)((ts + 511) / 512) * Ts + 2.0;          // + one kernel boundary
    }
    // 256 x 128 tiles, two workgroups per CU (gemm_nt_duo): fp32 residual / 16-bit epilogues of the transformer towers
    const bool duo_ok = K >= 128 && (out_f32 || (epilogue != HGR_EPI_BIAS_RESIDUAL && epilogue != HGR_EPI_ACCUM)) &&
                        (int64_t)M * lda * 2 < (1ll << 32) && (int64_t)N * ldw * 2 < (1ll << 32) && ldc < (1 << 20) && ldr < (1 << 20);
    auto launch_d = [&]() {
        GemmArgs a;
        a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias;
        a.res = (const float *)residual; a.ldr = ldr; a.M = M; a.N = N; a.K = K;
        a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 127) / 128;
        a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0;
        a.vec_ok = vec ? 1 : 0; a.dbg = dbg; a.kc = 0; a.csplit = 0; a.group = duo_group();
        dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
        if (dtype == HGR_BF16) launch_duo<HGR_BF16>(a, epilogue, out_f32 != 0, grid, s);
        else launch_duo<HGR_F16>(a, epilogue, out_f32 != 0, grid, s);
    };
    // Plan choice for the shapes gemm_nt_duo covers: measured (tools/gemm_plan_ab.py, same-process A/B, f16, one MI355X; bit-identical
    // outputs): qkv 97 -> 91 us, out-proj 59 -> 49, c_fc 136 -> 124, c_proj 145 -> 126, patch 124 -> 118, class logits 24.7 -> 20.7,
    // text qkv 168 -> 154, ViT-L/14 out / proj 411 -> 361 / 1029 -> 999, ViT-L/14 qkv / fc a tie; a launch with fewer tiles than
    // half the chip's 512 slots (small text batches, ragged test shapes) stays on the 128^2 / cost-model plans.
    const int64_t tduo = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
    // ... and an output much narrower than its 128-column tiles (the 64-channel ResNet stage: half of every tile would be padding;
    // measured 252 vs 194 us on the 256 x 64 arrangement of the small kernel) stays where it was
    const bool duo_fits = (N + 127) / 128 * 128 - N <= N / 8;
    if (duo_ok && (force == 2 || (force == 0 && tduo >= 256 && duo_fits))) launch_d();
    else if (force == 128 || K < 128) launch(0, M, false);
    else if (epi_has_idn16(epilogue) && force != 256) launch(0, M, false);   // only the 128 kernel loads the identity / stores by full lines
    else if (force == 256) launch(0, M, true);
    else if (cost_split < cost_big && cost_split < cost_small) { launch(0, m1, true); launch(m1, M - m1, false); }
    else launch(0, M, cost_big <= cost_small && t256 >= 128);
    HGR_CHECK_LAUNCH("hgr_gemm_nt");
    return HGR_OK;
}

static int conv3x3_launch(const void *x, const void *w, const float *bias, void *out,
                          int B, int H, int W, int C, int Cout, int stride, int Kp, int dtype, bool relu, void *stream) {
    HGR_REQUIRE(x && w && out && (bias || !relu), "hgr_conv3x3_nhwc: null operand");
    HGR_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && (stride == 1 || stride == 2), "hgr_conv3x3_nhwc: bad geometry B=%d H=%d W=%d Cout=%d stride=%d", B, H, W, Cout, stride);
    HGR_REQUIRE(C >= 8 && C % 8 == 0 && C <= 16384, "hgr_conv3x3_nhwc: C=%d must be a multiple of 8 in [8, 16384]", C);
    HGR_REQUIRE(Kp >= 9 * C && Kp % BK == 0, "hgr_conv3x3_nhwc: Kp=%d must be >= 9*C and a multiple of %d", Kp, BK);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(w, 16) && hgr_aligned(out, 8) && (!bias || hgr_aligned(bias, 16)) && Cout % 4 == 0, "hgr_conv3x3_nhwc: misaligned operand / Cout %% 4 != 0");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_conv3x3_nhwc: bad dtype %d", dtype);
    // 32 input channels (the stem at 112 x 112): the direct kernel of hgr_conv_direct.hip; HGR_CONV_DIRECT=0 keeps the implicit GEMM
    static int direct_env = -1;
    if (direct_env < 0) { const char *e = getenv("HGR_CONV_DIRECT"); direct_env = e ? atoi(e) : 1; }
    if (direct_env && C == 32 && stride == 1 && (Cout == 32 || Cout == 64) && Kp >= 288 && hgr_aligned(out, 16))
        return hgr_conv3x3_c32_launch(x, w, bias, out, B, H, W, Cout, Kp, dtype, relu ? 1 : 0, stream, 0);
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const int64_t M64 = (int64_t)B * Ho * Wo;
    HGR_REQUIRE(M64 < (1ll << 31), "hgr_conv3x3_nhwc: too many output pixels");
    GemmArgs a;
    a.A = (const char *)x; a.lda = 0; a.W = (const char *)w; a.ldw = Kp;
    a.C = out; a.ldc = Cout; a.bias = bias; a.res = nullptr; a.ldr = 0;
    a.M = (int)M64; a.N = Cout; a.K = Kp;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (Cout + BN - 1) / BN;
    a.m_fastest = 0; a.vec_ok = 1; a.dbg = 0; a.kc = 0; a.csplit = 0;
    a.cH = H; a.cW = W; a.cC = C; a.cStride = stride; a.cHo = Ho; a.cWo = Wo;
    a.cMagic = (unsigned)(((1ull << 32) + (unsigned)C - 1) / (unsigned)C);
    // big tiles when the output is at least 256 wide-ish and the launch has >= 160 of them (measured at batch 512: 14x14x256
    // 160 -> 142 us with 392 tiles, 7x7x512 160 -> 135 us with 196 tiles; K is 2304 / 4608 there, so one round is long)
    const int64_t t256 = (int64_t)((a.M + 255) / 256) * ((Cout + 255) / 256);
    // ... and 256-wide tiles do not waste more columns than 128-wide ones would (Cout = 128: half of every 256^2 tile would be
    // padding - measured 1066 vs 646 us at 56x56x128 and 285 vs 160 us at 28x28x128, batch 512)
    const int waste256 = (Cout + 255) / 256 * 256 - Cout, waste128 = (Cout + 127) / 128 * 128 - Cout;
    const bool big = relu && hgr_gemm_force_tile() != 128 && Kp >= 128 && Cout >= 128 && (t256 >= 160 || hgr_gemm_force_tile() == 256) && waste256 <= waste128;
    if (big) {
        a.tiles_m = (a.M + 255) / 256; a.tiles_n = (Cout + 255) / 256;
        dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
        if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_256<HGR_BF16, HGR_EPI_BIAS_RELU, false, true>), grid, dim3(NT256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemm_nt_256<HGR_F16, HGR_EPI_BIAS_RELU, false, true>), grid, dim3(NT256), 0, (hipStream_t)stream, a);
    } else {
        if (Cout <= 64 && a.M >= 1024) {              // tall 256 x 64 tiles: no MFMAs spent on columns that do not exist
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = 1;
            dim3 gt((unsigned)a.tiles_m);
            if (relu) {
                if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_BIAS_RELU, false, true, true>), gt, dim3(NT), 0, (hipStream_t)stream, a);
                else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_BIAS_RELU, false, true, true>), gt, dim3(NT), 0, (hipStream_t)stream, a);
            } else {
                if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_NONE, false, true, true>), gt, dim3(NT), 0, (hipStream_t)stream, a);
                else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_NONE, false, true, true>), gt, dim3(NT), 0, (hipStream_t)stream, a);
            }
            HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
            return HGR_OK;
        }
        dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
        if (relu) {
            if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_BIAS_RELU, false, true>), grid, dim3(NT), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_BIAS_RELU, false, true>), grid, dim3(NT), 0, (hipStream_t)stream, a);
        } else {
            if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_NONE, false, true>), grid, dim3(NT), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_NONE, false, true>), grid, dim3(NT), 0, (hipStream_t)stream, a);
        }
    }
    HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
    return HGR_OK;
}

extern "C" int hgr_conv3x3_nhwc(const void *x, const void *w, const float *bias, void *out,
                                int B, int H, int W, int C, int Cout, int stride, int Kp, int dtype, void *stream) {
    return conv3x3_launch(x, w, bias, out, B, H, W, C, Cout, stride, Kp, dtype, true, stream);
}

extern "C" int hgr_conv3x3_nhwc_plain(const void *x, const void *w, void *out, int B, int H, int W, int C, int Cout, int Kp,
                                      int dtype, void *stream) {
    return conv3x3_launch(x, w, nullptr, out, B, H, W, C, Cout, 1, Kp, dtype, false, stream);
}

extern "C" int hgr_gemm_nt_splitk(const void *A, int64_t lda, const void *W, int64_t ldw, float *partial, int64_t ldc,
                                  int M, int N, int K, int kc, int dtype, void *stream) {
    HGR_REQUIRE(A && W && partial, "hgr_gemm_nt_splitk: null operand");
    HGR_REQUIRE(M >= 1 && N >= 1 && K >= BK && K % BK == 0 && kc >= BK && kc % BK == 0, "hgr_gemm_nt_splitk: bad shape M=%d N=%d K=%d kc=%d", M, N, K, kc);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && hgr_aligned(A, 16) && hgr_aligned(W, 16), "hgr_gemm_nt_splitk: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
    HGR_REQUIRE(ldc >= N && ldc % 4 == 0 && hgr_aligned(partial, 16), "hgr_gemm_nt_splitk: partial must be 16-byte aligned, ldc %% 4 == 0");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt_splitk: bad dtype %d", dtype);
    const int S = (K + kc - 1) / kc;
    HGR_REQUIRE(S <= 65535, "hgr_gemm_nt_splitk: %d splits exceed the grid limit", S);
    GemmArgs a;
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw;
    a.C = partial; a.ldc = ldc; a.bias = nullptr; a.res = nullptr; a.ldr = 0;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.m_fastest = 0; a.vec_ok = 1; a.dbg = 0; a.kc = kc; a.csplit = (int64_t)M * ldc;
    a.cH = a.cW = a.cC = a.cStride = a.cHo = a.cWo = 0; a.cMagic = 0;
    // every slice at least 2 K-tiles deep and an output of at least one 256^2 tile: the deep-pipelined kernel
    const bool big = hgr_gemm_force_tile() != 128 && M >= 256 && N >= 256 && kc >= 128 && (K - (S - 1) * kc) >= 128;
    if (big) {
        a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 255) / 256;
        dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)S);
        if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_256<HGR_BF16, HGR_EPI_NONE, true>), grid, dim3(NT256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemm_nt_256<HGR_F16, HGR_EPI_NONE, true>), grid, dim3(NT256), 0, (hipStream_t)stream, a);
        HGR_CHECK_LAUNCH("hgr_gemm_nt_splitk");
        return HGR_OK;
    }
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)S);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_128<HGR_BF16, HGR_EPI_NONE, true, false>), grid, dim3(NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_nt_128<HGR_F16, HGR_EPI_NONE, true, false>), grid, dim3(NT), 0, (hipStream_t)stream, a);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_splitk");
    return HGR_OK;
}

// ---- LayerNorm folded into the GEMMs around it --------------------------------------------------------------------------
namespace {
int ln_common_checks(const char *who, const void *A, int64_t lda, const void *W, int64_t ldw, int M, int N, int K, int dtype) {
    HGR_REQUIRE(A && W, "%s: null operand", who);
    HGR_REQUIRE(M >= 1 && N >= 128 && N % 128 == 0 && K >= 128 && K % 64 == 0, "%s: bad shape M=%d N=%d K=%d (N %% 128 == 0, K %% 64 == 0, K >= 128)", who, M, N, K);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && hgr_aligned(A, 16) && hgr_aligned(W, 16), "%s: operands must be 16-byte aligned with leading dimensions %% 8 == 0", who);
    HGR_REQUIRE((int64_t)M * lda * 2 < (1ll << 32) && (int64_t)N * ldw * 2 < (1ll << 32) && N < (1 << 20), "%s: operands beyond 4 GB / N >= 2^20", who);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "%s: bad dtype %d", who, dtype);
    return HGR_OK;
}
void ln_args(GemmArgs &a, const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc, int M, int N, int K) {
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = nullptr; a.res = nullptr; a.ldr = 0;
    a.M = M; a.N = N; a.K = K; a.tiles_m = (M + 255) / 256; a.tiles_n = N / 128;
    a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0; a.vec_ok = 1; a.dbg = duo_dbg(); a.kc = 0; a.csplit = 0; a.group = duo_group();
    a.cH = a.cW = a.cC = a.cStride = a.cHo = a.cWo = 0; a.cMagic = 0;
    a.ln_stats = nullptr; a.ln_slots = 0; a.ln_eps = 0.f; a.ln_xh = a.ln_xl = nullptr; a.ln_ldx = 0; a.ln_s = a.ln_c = nullptr;
}
}  // namespace

extern "C" int hgr_gemm_nt_res_stats(const void *A, int64_t lda, const void *W, int64_t ldw, void *xh, void *xl, int64_t ldx,
                                     const float *bias, float *stats, int M, int N, int K, int dtype, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_res_stats", A, lda, W, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(xh && xl && bias && stats, "hgr_gemm_nt_res_stats: null xh / xl / bias / stats");
    HGR_REQUIRE(ldx >= N && ldx % 4 == 0 && ldx < (1 << 20) && hgr_aligned(xh, 8) && hgr_aligned(xl, 8), "hgr_gemm_nt_res_stats: xh / xl must be 8-byte aligned, ldx %% 4 == 0, ldx < 2^20");
    HGR_REQUIRE(hgr_aligned(bias, 16) && hgr_aligned(stats, 8), "hgr_gemm_nt_res_stats: bias must be 16-byte, stats 8-byte aligned");
    GemmArgs a;
    ln_args(a, A, lda, W, ldw, nullptr, 0, M, N, K);
    a.bias = bias;
    a.ln_stats = stats; a.ln_slots = N / 64; a.ln_xh = xh; a.ln_xl = xl; a.ln_ldx = ldx;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_duo<HGR_BF16, HGR_EPI_BIAS_RESIDUAL, true, 1>), grid, dim3(NTD), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_nt_duo<HGR_F16, HGR_EPI_BIAS_RESIDUAL, true, 1>), grid, dim3(NTD), 0, (hipStream_t)stream, a);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_res_stats");
    return HGR_OK;
}

extern "C" int hgr_gemm_nt_ln(const void *X16, int64_t ldx, const void *Wfold, int64_t ldw, void *C, int64_t ldc,
                              const float *ln_s, const float *ln_c, const float *stats, float eps,
                              int M, int N, int K, int dtype, int act, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_ln", X16, ldx, Wfold, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(C && ln_s && ln_c && stats, "hgr_gemm_nt_ln: null C / ln_s / ln_c / stats");
    HGR_REQUIRE(K % 128 == 0, "hgr_gemm_nt_ln: the row width K=%d must be a multiple of 128 (two 64-column statistic slots per 16-byte load)", K);
    HGR_REQUIRE(ldc >= N && ldc % 8 == 0 && ldc < (1 << 20) && hgr_aligned(C, 16), "hgr_gemm_nt_ln: C must be 16-byte aligned with ldc %% 8 == 0, ldc < 2^20");
    HGR_REQUIRE(hgr_aligned(ln_s, 16) && hgr_aligned(ln_c, 16) && hgr_aligned(stats, 16), "hgr_gemm_nt_ln: ln_s / ln_c / stats must be 16-byte aligned");
    HGR_REQUIRE(act == 0 || act == 1, "hgr_gemm_nt_ln: act must be 0 (none) or 1 (QuickGELU), got %d", act);
    GemmArgs a;
    ln_args(a, X16, ldx, Wfold, ldw, C, ldc, M, N, K);
    a.ln_stats = const_cast<float *>(stats); a.ln_slots = K / 64; a.ln_eps = eps; a.ln_s = ln_s; a.ln_c = ln_c;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HGR_BF16) {
        if (act) hipLaunchKernelGGL((gemm_nt_duo<HGR_BF16, HGR_EPI_BIAS_QUICKGELU, false, 2>), grid, dim3(NTD), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_duo<HGR_BF16, HGR_EPI_BIAS, false, 2>), grid, dim3(NTD), 0, s, a);
    } else {
        if (act) hipLaunchKernelGGL((gemm_nt_duo<HGR_F16, HGR_EPI_BIAS_QUICKGELU, false, 2>), grid, dim3(NTD), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_duo<HGR_F16, HGR_EPI_BIAS, false, 2>), grid, dim3(NTD), 0, s, a);
    }
    HGR_CHECK_LAUNCH("hgr_gemm_nt_ln");
    return HGR_OK;
}

extern "C" int hgr_gemm_nt_bias_gelu_dual(const void *A, int64_t lda, const void *W, int64_t ldw, void *pre, int64_t ldpre, void *post, int64_t ldpost,
                                          const float *bias, int M, int N, int K, int dtype, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_bias_gelu_dual", A, lda, W, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(pre && post && bias, "hgr_gemm_nt_bias_gelu_dual: null pre / post / bias");
    HGR_REQUIRE(ldpre >= N && ldpost >= N && ldpre % 8 == 0 && ldpost % 8 == 0 && ldpre < (1 << 20) && ldpost < (1 << 20) && hgr_aligned(pre, 16) && hgr_aligned(post, 16) && hgr_aligned(bias, 16),
                "hgr_gemm_nt_bias_gelu_dual: pre / post / bias must be 16-byte aligned, leading dimensions >= N, %% 8 == 0, < 2^20");
    GemmArgs a;
    ln_args(a, A, lda, W, ldw, pre, ldpre, M, N, K);
    a.bias = bias; a.ln_xh = post; a.ln_ldx = ldpost;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_duo<HGR_BF16, HGR_EPI_BIAS, false, 4>), grid, dim3(NTD), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_nt_duo<HGR_F16, HGR_EPI_BIAS, false, 4>), grid, dim3(NTD), 0, (hipStream_t)stream, a);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_bias_gelu_dual");
    return HGR_OK;
}

// ---- logits GEMM with the evaluation consumers in its epilogue ------------------------------------------------------------
int hgr_logits_eval_rows_launch(const void *feat, const void *zslp, int D, int S, const unsigned long long *keys, const float *tmax,
                                const int *gp1, const float *gm2, const int32_t *level_first, int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                                const int32_t *epos, const int32_t *test_cols, int n_test, int k, int32_t *out_level, int32_t *out_top1,
                                int32_t *out_topk, int rows, int dtype, void *stream);      // hgr_select.hip

extern "C" int64_t hgr_logits_eval_workspace_bytes(int rows, int n_perm) {
    if (rows < 1 || n_perm < 128 || n_perm % 128) return -1;
    return (int64_t)rows * (n_perm / 64) * 56;           // per (row, slice): 8-byte train key + per 16-column group (max, position, second) = 4 x 12 bytes
}

extern "C" int hgr_logits_eval(const void *feat16, const void *zsl_perm16, int rows, int D, int n_perm,
                               const int32_t *tpos_perm, const int32_t *epos_perm, const int32_t *level_first,
                               int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                               const int32_t *test_cols, int n_test, int k,
                               int32_t *out_level, int32_t *out_top1, int32_t *out_topk, void *workspace, int dtype, void *stream) {
    HGR_REQUIRE(feat16 && zsl_perm16 && tpos_perm && epos_perm && level_first && filler_pos && train_cols && out_level && workspace, "hgr_logits_eval: null operand");
    HGR_REQUIRE(rows >= 1 && D >= 128 && D % 128 == 0 && D <= 1024, "hgr_logits_eval: rows=%d D=%d unsupported (D %% 128 == 0, D <= 1024)", rows, D);
    HGR_REQUIRE(n_perm >= 128 && n_perm % 128 == 0 && n_perm / 64 <= 1024, "hgr_logits_eval: n_perm=%d must be a multiple of 128 and <= 65536 (level-aligned, padded columns)", n_perm);
    HGR_REQUIRE(n_levels >= 1 && n_levels <= 32 && n_train >= 1, "hgr_logits_eval: bad sizes (n_levels <= 32)");
    HGR_REQUIRE(k == 0 || (out_topk && test_cols && k >= 1 && k <= 32 && n_test >= k), "hgr_logits_eval: bad top-k arguments");
    HGR_REQUIRE(hgr_aligned(feat16, 16) && hgr_aligned(zsl_perm16, 16) && hgr_aligned(tpos_perm, 16) && hgr_aligned(epos_perm, 16) && hgr_aligned(workspace, 16), "hgr_logits_eval: operands must be 16-byte aligned");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_logits_eval: bad dtype %d", dtype);
    const int S = n_perm / 64;
    GemmArgs a;
    ln_args(a, feat16, D, zsl_perm16, D, nullptr, 0, rows, n_perm, D);
    a.ev_key = (unsigned long long *)workspace;
    a.ev_tmax = (float *)((char *)workspace + (size_t)rows * S * 8);
    a.ev_p1 = (int *)((char *)workspace + (size_t)rows * S * 24);
    a.ev_m2 = (float *)((char *)workspace + (size_t)rows * S * 40);
    a.ev_tpos = tpos_perm; a.ev_epos = epos_perm; a.ev_slices = S;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_duo<HGR_BF16, HGR_EPI_NONE, true, 3>), grid, dim3(NTD), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_nt_duo<HGR_F16, HGR_EPI_NONE, true, 3>), grid, dim3(NTD), 0, (hipStream_t)stream, a);
    HGR_CHECK_LAUNCH("hgr_logits_eval (tile stage)");
    return hgr_logits_eval_rows_launch(feat16, zsl_perm16, D, S, a.ev_key, a.ev_tmax, a.ev_p1, a.ev_m2, level_first, n_levels, filler_pos, train_cols, n_train,
                                       epos_perm, test_cols, n_test, k, out_level, out_top1, out_topk, rows, dtype, stream);
}
