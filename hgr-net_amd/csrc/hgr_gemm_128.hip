// gemm_nt_128: 128(M) x 128(N) x 64(K) tiles, 256 threads = 4 waves in 2(M) x 2(N), each wave 64 x 64 = 4 x 4 tiles of
// v_mfma_f32_16x16x32; two LDS stages of 32 KB, 2 workgroups per CU (one's epilogue overlaps the other's main loop).
// Loop: issue stage t+1, compute stage t, vmcnt(0) + one barrier per K-tile ("minimum 2-phase").  Also the implicit-GEMM 3x3
// convolution (CONV), the 256 x 64 arrangement for narrow outputs (TALL) and the one-stage K = 64 variant (ONEK).
#include "hgr_gemm_common.h"

namespace hgr_gemm {

constexpr int NT = 256;

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
                    if (EPI == HGR_EPI_BIAS_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    const int row = j * 16 + r;
                    typename T16<DT>::vec4 o4;
                    if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = quick_gelu16<DT>(v[e]);            // product + rounding as one operation (hgr_gemm_common.h)
                    } else o4 = cvt4<DT>(v[0], v[1], v[2], v[3]);
                    *(typename T16<DT>::vec4 *)(my + row * 128 + (((i * 2 + (g >> 1)) ^ (row & 7)) * 16) + (g & 1) * 8) = o4;
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
                if (EPI == HGR_EPI_QGELU_GRAD16) {
                    const typename T16<DT>::vec8 iv = __builtin_bit_cast(typename T16<DT>::vec8, idn[h][q]);
                    typename T16<DT>::vec8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = mul16<DT>(v[e], quick_gelu_grad((float)iv[e]));      // one rounding, as in every form of this epilogue
                    *(u32x4 *)(dst + (int64_t)(h * 32 + q * 8) * p.ldc) = __builtin_bit_cast(u32x4, o);
                    continue;
                }
                if (HAS_IDN) {
                    const typename T16<DT>::vec8 iv = __builtin_bit_cast(typename T16<DT>::vec8, idn[h][q]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] + (float)iv[e];
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


namespace {
template <int DT, int EPI>
void plain_epi(const GemmArgs &a, bool out32, dim3 grid, hipStream_t s) {
    if constexpr (EPI == HGR_EPI_BIAS || EPI == HGR_EPI_BIAS_RELU || EPI == HGR_EPI_BIAS_ADD16_RELU) {
        if (!out32 && a.K == BK && !a.kc) {
            hipLaunchKernelGGL((gemm_nt_128<DT, EPI, false, false, false, true>), grid, dim3(NT), 0, s, a);
            return;
        }
    }
    if constexpr (EPI == HGR_EPI_QGELU_GRAD16 || EPI == HGR_EPI_BIAS_ADD16_RELU) {      // 16-bit output only (checked by the host entry)
        hipLaunchKernelGGL((gemm_nt_128<DT, EPI, false>), grid, dim3(NT), 0, s, a);
    } else {
        if (out32) hipLaunchKernelGGL((gemm_nt_128<DT, EPI, true>), grid, dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_128<DT, EPI, false>), grid, dim3(NT), 0, s, a);
    }
}

template <int DT>
void launch_128_dt(const GemmArgs &a, int epi, bool out32, int variant, dim3 grid, hipStream_t s) {
    if (variant == V128_TALL) {
        if (a.K == BK) hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_BIAS_RELU, false, false, true, true>), grid, dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_BIAS_RELU, false, false, true>), grid, dim3(NT), 0, s, a);
        return;
    }
    if (variant == V128_CONV) {
        if (epi == HGR_EPI_BIAS_RELU) hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_BIAS_RELU, false, true>), grid, dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_NONE, false, true>), grid, dim3(NT), 0, s, a);
        return;
    }
    if (variant == V128_CONV_TALL) {
        if (epi == HGR_EPI_BIAS_RELU) hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_BIAS_RELU, false, true, true>), grid, dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_128<DT, HGR_EPI_NONE, false, true, true>), grid, dim3(NT), 0, s, a);
        return;
    }
    switch (epi) {
        case HGR_EPI_NONE: plain_epi<DT, HGR_EPI_NONE>(a, out32, grid, s); break;
        case HGR_EPI_BIAS: plain_epi<DT, HGR_EPI_BIAS>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_QUICKGELU: plain_epi<DT, HGR_EPI_BIAS_QUICKGELU>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_RELU: plain_epi<DT, HGR_EPI_BIAS_RELU>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_ADD16_RELU: plain_epi<DT, HGR_EPI_BIAS_ADD16_RELU>(a, out32, grid, s); break;
        case HGR_EPI_ACCUM: plain_epi<DT, HGR_EPI_ACCUM>(a, out32, grid, s); break;
        case HGR_EPI_QGELU_GRAD16: plain_epi<DT, HGR_EPI_QGELU_GRAD16>(a, out32, grid, s); break;
        default: plain_epi<DT, HGR_EPI_BIAS_RESIDUAL>(a, out32, grid, s); break;
    }
}
}  // namespace

void launch_128(const GemmArgs &a, int dtype, int epi, bool out32, int variant, dim3 grid, hipStream_t s) {
    if (dtype == HGR_BF16) launch_128_dt<HGR_BF16>(a, epi, out32, variant, grid, s);
    else launch_128_dt<HGR_F16>(a, epi, out32, variant, grid, s);
}

}  // namespace hgr_gemm
