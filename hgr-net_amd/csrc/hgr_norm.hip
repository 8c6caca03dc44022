// Row-wise fp32 normalisation kernels: LayerNorm (clip/model.py:153-159), ViT token assembly +
// ln_pre (clip/model.py:223-225), row L2 normalisation (model/clip_tree.py:323,330).
//
// All three are HBM-bound streaming ops: one wave owns one row, the row lives in registers
// (float4 per lane, up to 16 per lane = 4096 columns), loads and stores are 16-byte (8-byte for the
// 16-bit outputs) and coalesced, statistics are two-pass (mean, then centred second moment) like
// torch's LayerNorm, reduced with wave shuffles.  4 rows per 256-thread block.
#include "hgr_common.h"

namespace {

constexpr int MAXV = 16;   // float4 per lane -> W <= 4096 (kernels are instantiated for 1, 2, 4, 8, 16 so narrow rows keep full occupancy)

template <int DT, bool OUT32, int NV>
__global__ __launch_bounds__(256) void layernorm_rows(const float *__restrict__ x, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, void *__restrict__ y,
                                                      int rows, int W, int64_t row_mul, const int32_t *__restrict__ row_idx, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t src = (int64_t)row * row_mul + (row_idx ? row_idx[row] : 0);
    const f32x4 *xr = (const f32x4 *)(x + src * W);
    const int nv = W >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv) {
            v[i] = (c < nv) ? xr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv && c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
    const f32x4 *gv = (const f32x4 *)gamma, *bv = (const f32x4 *)beta;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv && c < nv) {
            const f32x4 ga = gv[c], be = bv[c];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * ga[e] + be[e];
            if (OUT32) ((f32x4 *)((float *)y + (int64_t)row * W))[c] = o;
            else ((typename T16<DT>::vec4 *)((typename T16<DT>::elem *)y + (int64_t)row * W))[c] = cvt4<DT>(o[0], o[1], o[2], o[3]);
        }
    }
}

// x[b*L + t] = LN((t == 0 ? cls : patches[b*G + t-1]) + pos[t])
template <int NV, int DT = HGR_F16, bool STATS = false>
__global__ __launch_bounds__(256) void vit_embed_ln(const float *__restrict__ patches, const float *__restrict__ cls,
                                                    const float *__restrict__ pos, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float *__restrict__ x,
                                                    int B, int G, int W, float eps, void *__restrict__ xh = nullptr, void *__restrict__ xl = nullptr,
                                                    float *__restrict__ stats = nullptr) {
    const int lane = threadIdx.x & 63;
    const int L = G + 1;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, t = row - b * L;
    const f32x4 *src = (const f32x4 *)(t == 0 ? cls : patches + ((int64_t)b * G + (t - 1)) * W);
    const f32x4 *pr = (const f32x4 *)(pos + (int64_t)t * W);
    const int nv = W >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv) {
            v[i] = (c < nv) ? (src[c] + pr[c]) : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv && c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
    const f32x4 *gv = (const f32x4 *)gamma, *bv = (const f32x4 *)beta;
    f32x4 *out = STATS ? nullptr : (f32x4 *)(x + (int64_t)row * W);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv && c < nv) {
            const f32x4 ga = gv[c], be = bv[c];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * ga[e] + be[e];
            if (!STATS) out[c] = o;
            if (STATS) {
                // the first block's LayerNorm is folded into its QKV GEMM (hgr_gemm_nt_ln): the row leaves as the 16-bit pair
                // (hi, lo), x = hi + lo, the form the residual stream keeps between the GEMMs, + the (sum, sum of squares)
                // of every 64-column slot = one DPP row of 16 lanes per slot (W % 64 == 0)
                typename T16<DT>::vec4 nh;
                unsigned nl = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) { unsigned q; typename T16<DT>::elem h; pair_split<DT>(o[e], h, q); nh[e] = h; nl |= q << (8 * e); }
                ((typename T16<DT>::vec4 *)((typename T16<DT>::elem *)xh + (int64_t)row * W))[c] = nh;
                ((unsigned *)((unsigned char *)xl + (int64_t)row * W))[c] = nl;
                const float s1 = row16_sum((o[0] + o[1]) + (o[2] + o[3]));
                const float s2 = row16_sum((o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]));
                if ((lane & 15) == 0) *(float2 *)(stats + ((int64_t)row * (W >> 6) + (c >> 4)) * 2) = make_float2(s1, s2);
            }
        }
    }
}

// x fp32 [rows, W] -> the 16-bit pair (hi, lo) + per-slot LayerNorm statistics (the form hgr_gemm_nt_res_stats keeps), for the
// input of the first residual block of a tower
template <int DT>
__global__ __launch_bounds__(256) void row_stats16(const float *__restrict__ x, void *__restrict__ xh, void *__restrict__ xl, float *__restrict__ stats, int rows, int W) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4 *xr = (const f32x4 *)(x + (int64_t)row * W);
    const int nv = W >> 2;
    for (int c0 = 0; c0 < nv; c0 += 64) {
        const int c = c0 + lane;
        const bool ok = c < nv;                      // whole DPP rows are in or out: W % 64 == 0
        const f32x4 v = ok ? xr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
        const float s2 = row16_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
        if (ok) {
            typename T16<DT>::vec4 nh;
            unsigned nl = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) { unsigned q; typename T16<DT>::elem h; pair_split<DT>(v[e], h, q); nh[e] = h; nl |= q << (8 * e); }
            ((typename T16<DT>::vec4 *)((typename T16<DT>::elem *)xh + (int64_t)row * W))[c] = nh;
            ((unsigned *)((unsigned char *)xl + (int64_t)row * W))[c] = nl;
            if ((lane & 15) == 0) *(float2 *)(stats + ((int64_t)row * (W >> 6) + (c >> 4)) * 2) = make_float2(s1, s2);
        }
    }
}

// out[i] (fp32, compact rows) = hi[src] + lo[src], src = i * row_mul + (row_idx ? row_idx[i] : 0): selected rows of the pair back
// in fp32 (ln_post on the class tokens, ln_final on the EOT rows, taps)
template <int DT>
__global__ __launch_bounds__(256) void pair_rows_f32(const void *__restrict__ xh, const void *__restrict__ xl, float *__restrict__ out, int rows, int W,
                                                     int64_t row_mul, const int32_t *__restrict__ row_idx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t src = (int64_t)row * row_mul + (row_idx ? row_idx[row] : 0);
    for (int c = lane; c < (W >> 2); c += 64) {
        const typename T16<DT>::vec4 h = ((const typename T16<DT>::vec4 *)((const typename T16<DT>::elem *)xh + src * W))[c];
        const unsigned l = ((const unsigned *)((const unsigned char *)xl + src * W))[c];
        ((f32x4 *)(out + (int64_t)row * W))[c] = (f32x4){pair_dec<DT>(h[0], l & 255u), pair_dec<DT>(h[1], (l >> 8) & 255u), pair_dec<DT>(h[2], (l >> 16) & 255u), pair_dec<DT>(h[3], l >> 24)};
    }
}

template <int DT, int NV>
__global__ __launch_bounds__(256) void l2norm_rows(const float *__restrict__ x, void *__restrict__ y16, float *__restrict__ y32, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4 *xr = (const f32x4 *)(x + (int64_t)row * D);
    const int nv = D >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv) {
            v[i] = (c < nv) ? xr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
    }
    const float inv = 1.0f / sqrtf(wave_sum(s));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nv && c < nv) {
            const f32x4 o = v[i] * inv;
            if (y32) ((f32x4 *)(y32 + (int64_t)row * D))[c] = o;
            if (y16) ((typename T16<DT>::vec4 *)((typename T16<DT>::elem *)y16 + (int64_t)row * D))[c] = cvt4<DT>(o[0], o[1], o[2], o[3]);
        }
    }
}


// The visual head of a ViT (clip/model.py:231-233: x = ln_post(x[:, 0, :]); x = x @ proj) in ONE launch: the class-token rows of the
// residual pair -> fp32 -> LayerNorm (the arithmetic of layernorm_rows: two-pass statistics, one wave per row) -> 16-bit -> product
// with proj^T on the matrix cores -> fp32 features.  Replaces pair_rows_f32 + layernorm_rows + a 16-tile GEMM whose 12-deep serial
// K loop took 18 us for 0.4 GFLOP.  Workgroup (i, j): rows 16 i .. 16 i + 15, output columns 128 j .. 128 j + 127; every workgroup
// normalises its 16 rows itself (48 KB of reads, cheaper than a round trip through memory), wave w owns 32 columns and streams its
// slice of proj straight from L2 into MFMA fragments (8 k-steps in flight per trip).
template <int DT, int NV>
__global__ __launch_bounds__(256) void vit_head(const void *__restrict__ xh, const void *__restrict__ xl, int64_t ldx, int64_t row_mul,
                                                const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                const void *__restrict__ projt, float *__restrict__ out, int B, int W, int D) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    extern __shared__ __attribute__((aligned(16))) char hs[];              // [16][W * 2 + 16] bytes: the normalised rows, 16-bit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 16, col0 = blockIdx.y * 128;
    const int RS = W * 2 + 16;
    const int nv = W >> 2;
    const f32x4 *gv = (const f32x4 *)gamma, *bv = (const f32x4 *)beta;
    // wave w normalises rows 4 w .. 4 w + 3; the four rows' loads are requested together (one memory round trip per wave, not four)
    f32x4 v[4][NV];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int row = min(row0 + wave * 4 + q4, B - 1);
        const E *ph = (const E *)xh + (int64_t)row * row_mul * ldx;
        const unsigned char *pl = (const unsigned char *)xl + (int64_t)row * row_mul * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (i * 64 < nv) {
                const int cc = min(c, nv - 1);
                const vec4 h4 = ((const vec4 *)ph)[cc];
                const unsigned l4 = ((const unsigned *)pl)[cc];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q4][i][e] = c < nv ? pair_dec<DT>(h4[e], (l4 >> (8 * e)) & 255u) : 0.f;
            }
        }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int rr = wave * 4 + q4;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * 64 < nv) s += v[q4][i][0] + v[q4][i][1] + v[q4][i][2] + v[q4][i][3];
        const float mean = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (i * 64 < nv && c < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[q4][i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (i * 64 < nv && c < nv) {
                const f32x4 ga = gv[c], be = bv[c];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[q4][i][e] - mean) * rstd * ga[e] + be[e];
                *(vec4 *)(hs + rr * RS + c * 8) = cvt4<DT>(o[0], o[1], o[2], o[3]);
            }
        }
    }
    __syncthreads();
    // out^T tile [n][m] = proj^T rows (A operand, from global memory) x normalised rows (B operand, from LDS): lane (r, g) holds
    // out[row0 + r][n .. n + 3], n = col0 + wave * 32 + t * 16 + 4 g
    const int r = lane & 15, g = lane >> 4;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const E *w0 = (const E *)projt + (int64_t)min(col0 + wave * 32 + r, D - 1) * W + g * 8;
    const E *w1 = (const E *)projt + (int64_t)min(col0 + wave * 32 + 16 + r, D - 1) * W + g * 8;
    const char *hb = hs + r * RS + g * 16;
    const int ks = W >> 5;
    for (int kk = 0; kk < ks; kk += 8) {
        vec8 f0[8], f1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kq = min(kk + u, ks - 1);
            f0[u] = *(const vec8 *)(w0 + kq * 32); f1[u] = *(const vec8 *)(w1 + kq * 32);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kk + u < ks) {
                const vec8 xf = *(const vec8 *)(hb + (kk + u) * 64);
                acc[0] = T16<DT>::mfma16(f0[u], xf, acc[0]);
                acc[1] = T16<DT>::mfma16(f1[u], xf, acc[1]);
            }
        }
    }
    if (row0 + r < B) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = col0 + wave * 32 + t * 16 + g * 4;
            if (n + 3 < D) *(f32x4 *)(out + (int64_t)(row0 + r) * D + n) = acc[t];
            else
                for (int e = 0; e < 4 && n + e < D; ++e) out[(int64_t)(row0 + r) * D + n + e] = acc[t][e];
        }
    }
}

}  // namespace

extern "C" int hgr_layernorm(const float *x, const float *gamma, const float *beta, void *y, int rows, int W,
                             int64_t row_mul, const int32_t *row_idx, float eps, int dtype, int out_f32, void *stream) {
    HGR_REQUIRE(x && gamma && beta && y, "hgr_layernorm: null operand");
    HGR_REQUIRE(rows >= 1 && W >= 4 && W % 4 == 0 && W <= 4 * 64 * MAXV, "hgr_layernorm: rows=%d W=%d unsupported (W %% 4 == 0, W <= %d)", rows, W, 4 * 64 * MAXV);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(gamma, 16) && hgr_aligned(beta, 16) && hgr_aligned(y, 16), "hgr_layernorm: operands must be 16-byte aligned");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_layernorm: bad dtype %d", dtype);
    HGR_REQUIRE(row_mul >= 1, "hgr_layernorm: row_mul must be >= 1");
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define HGR_LN(NVV)                                                                                                          \
    do {                                                                                                                     \
        if (out_f32) hipLaunchKernelGGL((layernorm_rows<HGR_BF16, true, NVV>), grid, block, 0, s, x, gamma, beta, y, rows, W, row_mul, row_idx, eps); \
        else if (dtype == HGR_BF16) hipLaunchKernelGGL((layernorm_rows<HGR_BF16, false, NVV>), grid, block, 0, s, x, gamma, beta, y, rows, W, row_mul, row_idx, eps); \
        else hipLaunchKernelGGL((layernorm_rows<HGR_F16, false, NVV>), grid, block, 0, s, x, gamma, beta, y, rows, W, row_mul, row_idx, eps); \
    } while (0)
    const int nvl = (W / 4 + 63) / 64;
    if (nvl <= 1) HGR_LN(1); else if (nvl <= 2) HGR_LN(2); else if (nvl <= 4) HGR_LN(4); else if (nvl <= 8) HGR_LN(8); else HGR_LN(16);
#undef HGR_LN
    HGR_CHECK_LAUNCH("hgr_layernorm");
    return HGR_OK;
}

extern "C" int hgr_vit_embed_ln(const float *patches, const float *class_embedding, const float *positional_embedding,
                                const float *gamma, const float *beta, float *x, int B, int G, int W, float eps, void *stream) {
    HGR_REQUIRE(patches && class_embedding && positional_embedding && gamma && beta && x, "hgr_vit_embed_ln: null operand");
    HGR_REQUIRE(B >= 1 && G >= 1 && W >= 4 && W % 4 == 0 && W <= 4 * 64 * MAXV, "hgr_vit_embed_ln: B=%d G=%d W=%d unsupported", B, G, W);
    HGR_REQUIRE(hgr_aligned(patches, 16) && hgr_aligned(class_embedding, 16) && hgr_aligned(positional_embedding, 16) &&
                hgr_aligned(gamma, 16) && hgr_aligned(beta, 16) && hgr_aligned(x, 16), "hgr_vit_embed_ln: operands must be 16-byte aligned");
    const int rows = B * (G + 1);
#define HGR_VE(NVV) hipLaunchKernelGGL((vit_embed_ln<NVV>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, patches, class_embedding, positional_embedding, gamma, beta, x, B, G, W, eps)
    const int nvl = (W / 4 + 63) / 64;
    if (nvl <= 1) HGR_VE(1); else if (nvl <= 2) HGR_VE(2); else if (nvl <= 4) HGR_VE(4); else if (nvl <= 8) HGR_VE(8); else HGR_VE(16);
#undef HGR_VE
    HGR_CHECK_LAUNCH("hgr_vit_embed_ln");
    return HGR_OK;
}

extern "C" int hgr_l2norm_rows(const float *x, void *y16, float *y32, int rows, int D, int dtype, void *stream) {
    HGR_REQUIRE(x && (y16 || y32), "hgr_l2norm_rows: null operand");
    HGR_REQUIRE(rows >= 1 && D >= 4 && D % 4 == 0 && D <= 4 * 64 * MAXV, "hgr_l2norm_rows: rows=%d D=%d unsupported", rows, D);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(y16, 8) && hgr_aligned(y32, 16), "hgr_l2norm_rows: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_l2norm_rows: bad dtype %d", dtype);
    dim3 grid((rows + 3) / 4), block(256);
#define HGR_L2(NVV)                                                                                                    \
    do {                                                                                                               \
        if (dtype == HGR_BF16) hipLaunchKernelGGL((l2norm_rows<HGR_BF16, NVV>), grid, block, 0, (hipStream_t)stream, x, y16, y32, rows, D); \
        else hipLaunchKernelGGL((l2norm_rows<HGR_F16, NVV>), grid, block, 0, (hipStream_t)stream, x, y16, y32, rows, D); \
    } while (0)
    const int nvl = (D / 4 + 63) / 64;
    if (nvl <= 1) HGR_L2(1); else if (nvl <= 2) HGR_L2(2); else if (nvl <= 4) HGR_L2(4); else if (nvl <= 8) HGR_L2(8); else HGR_L2(16);
#undef HGR_L2
    HGR_CHECK_LAUNCH("hgr_l2norm_rows");
    return HGR_OK;
}


extern "C" int hgr_vit_embed_ln_stats(const float *patches, const float *class_embedding, const float *positional_embedding,
                                      const float *gamma, const float *beta, void *xh, void *xl, float *stats,
                                      int B, int G, int W, float eps, int dtype, void *stream) {
    HGR_REQUIRE(patches && class_embedding && positional_embedding && gamma && beta && xh && xl && stats, "hgr_vit_embed_ln_stats: null operand");
    HGR_REQUIRE(B >= 1 && G >= 1 && W >= 64 && W % 64 == 0 && W <= 4 * 64 * MAXV, "hgr_vit_embed_ln_stats: B=%d G=%d W=%d unsupported (W %% 64 == 0)", B, G, W);
    HGR_REQUIRE(hgr_aligned(patches, 16) && hgr_aligned(class_embedding, 16) && hgr_aligned(positional_embedding, 16) &&
                hgr_aligned(gamma, 16) && hgr_aligned(beta, 16) && hgr_aligned(xh, 8) && hgr_aligned(xl, 8) && hgr_aligned(stats, 8),
                "hgr_vit_embed_ln_stats: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_vit_embed_ln_stats: bad dtype %d", dtype);
    const int rows = B * (G + 1);
    float *nox = nullptr;
#define HGR_VE(NVV) do { \
        if (dtype == HGR_BF16) hipLaunchKernelGGL((vit_embed_ln<NVV, HGR_BF16, true>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, patches, class_embedding, positional_embedding, gamma, beta, nox, B, G, W, eps, xh, xl, stats); \
        else hipLaunchKernelGGL((vit_embed_ln<NVV, HGR_F16, true>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, patches, class_embedding, positional_embedding, gamma, beta, nox, B, G, W, eps, xh, xl, stats); } while (0)
    const int nvl = (W / 4 + 63) / 64;
    if (nvl <= 1) HGR_VE(1); else if (nvl <= 2) HGR_VE(2); else if (nvl <= 4) HGR_VE(4); else if (nvl <= 8) HGR_VE(8); else HGR_VE(16);
#undef HGR_VE
    HGR_CHECK_LAUNCH("hgr_vit_embed_ln_stats");
    return HGR_OK;
}

extern "C" int hgr_row_stats16(const float *x, void *xh, void *xl, float *stats, int rows, int W, int dtype, void *stream) {
    HGR_REQUIRE(x && xh && xl && stats, "hgr_row_stats16: null operand");
    HGR_REQUIRE(rows >= 1 && W >= 64 && W % 64 == 0, "hgr_row_stats16: rows=%d W=%d unsupported (W %% 64 == 0)", rows, W);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(xh, 8) && hgr_aligned(xl, 8) && hgr_aligned(stats, 8), "hgr_row_stats16: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_row_stats16: bad dtype %d", dtype);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((row_stats16<HGR_BF16>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, xh, xl, stats, rows, W);
    else hipLaunchKernelGGL((row_stats16<HGR_F16>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, xh, xl, stats, rows, W);
    HGR_CHECK_LAUNCH("hgr_row_stats16");
    return HGR_OK;
}

extern "C" int hgr_pair_rows_f32(const void *xh, const void *xl, float *out, int rows, int W, int64_t row_mul, const int32_t *row_idx, int dtype, void *stream) {
    HGR_REQUIRE(xh && xl && out, "hgr_pair_rows_f32: null operand");
    HGR_REQUIRE(rows >= 1 && W >= 4 && W % 4 == 0 && row_mul >= 1, "hgr_pair_rows_f32: rows=%d W=%d row_mul=%lld unsupported", rows, W, (long long)row_mul);
    HGR_REQUIRE(hgr_aligned(xh, 8) && hgr_aligned(xl, 8) && hgr_aligned(out, 16), "hgr_pair_rows_f32: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_pair_rows_f32: bad dtype %d", dtype);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((pair_rows_f32<HGR_BF16>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, xh, xl, out, rows, W, row_mul, row_idx);
    else hipLaunchKernelGGL((pair_rows_f32<HGR_F16>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, xh, xl, out, rows, W, row_mul, row_idx);
    HGR_CHECK_LAUNCH("hgr_pair_rows_f32");
    return HGR_OK;
}

extern "C" int hgr_vit_head(const void *xh, const void *xl, int64_t ldx, int64_t row_mul, const float *gamma, const float *beta, float eps,
                            const void *proj_t, float *out, int B, int W, int D, int dtype, void *stream) {
    HGR_REQUIRE(xh && xl && gamma && beta && proj_t && out, "hgr_vit_head: null operand");
    HGR_REQUIRE(B >= 1 && W >= 64 && W % 32 == 0 && W <= 1920 && D >= 1 && D % 4 == 0 && ldx >= W && ldx % 4 == 0 && row_mul >= 1,
                "hgr_vit_head: bad shape B=%d W=%d D=%d ldx=%lld (W %% 32 == 0, W <= 1920: 16 rows in 64 KB of LDS, D %% 4 == 0)", B, W, D, (long long)ldx);
    HGR_REQUIRE(hgr_aligned(xh, 8) && hgr_aligned(xl, 8) && hgr_aligned(gamma, 16) && hgr_aligned(beta, 16) && hgr_aligned(proj_t, 16) && hgr_aligned(out, 16),
                "hgr_vit_head: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_vit_head: bad dtype %d", dtype);
    const size_t lds = (size_t)16 * (W * 2 + 16);
    dim3 grid((unsigned)((B + 15) / 16), (unsigned)((D + 127) / 128));
    const int nv = (W / 4 + 63) / 64;
#define HGR_HEAD(NV) do { if (dtype == HGR_BF16) hipLaunchKernelGGL((vit_head<HGR_BF16, NV>), grid, dim3(256), lds, (hipStream_t)stream, xh, xl, ldx, row_mul, gamma, beta, eps, proj_t, out, B, W, D); \
                          else hipLaunchKernelGGL((vit_head<HGR_F16, NV>), grid, dim3(256), lds, (hipStream_t)stream, xh, xl, ldx, row_mul, gamma, beta, eps, proj_t, out, B, W, D); } while (0)
    if (nv <= 1) HGR_HEAD(1); else if (nv <= 2) HGR_HEAD(2); else if (nv <= 4) HGR_HEAD(4); else if (nv <= 8) HGR_HEAD(8); else HGR_HEAD(16);
#undef HGR_HEAD
    HGR_CHECK_LAUNCH("hgr_vit_head");
    return HGR_OK;
}
