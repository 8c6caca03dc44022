// Small streaming kernels of the ModifiedResNet tower (clip/model.py:93-150): stem im2col, 2x2 average
// pooling, attention-pool token assembly and the single-query attention.  All HBM-bound; 16-byte
// accesses along the channel dimension (NHWC keeps channels contiguous).
#include "hgr_common.h"

namespace {

// one thread per output pixel: its 27 taps (ky, kx, c order) + zero padding are assembled in registers and written as
// eight 16-byte stores (a full 128-byte row per thread); the image reads are served from L1/L2 (each pixel is touched
// by up to 4 neighbouring outputs)
template <int DT>
__global__ __launch_bounds__(256) void stem_im2col(const float *__restrict__ img, typename T16<DT>::elem *__restrict__ out, int B, int R) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    const int Ho = (R - 1) / 2 + 1;
    const int64_t total = (int64_t)B * Ho * Ho;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < total; row += (int64_t)gridDim.x * 256) {
        const int wo = (int)(row % Ho), ho = (int)((row / Ho) % Ho), b = (int)(row / ((int64_t)Ho * Ho));
        vec8 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] = (E)0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int hi = ho * 2 - 1 + ky, wi = wo * 2 - 1 + kx;
            const bool ok = hi >= 0 && hi < R && wi >= 0 && wi < R;
            const int hc = min(max(hi, 0), R - 1), wc = min(max(wi, 0), R - 1);      // unconditional loads (clamped), then select
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int k = t * 3 + c;
                const float x = img[(((int64_t)b * 3 + c) * R + hc) * R + wc];
                v[k >> 3][k & 7] = (E)(ok ? x : 0.f);
            }
        }
        vec8 *o = (vec8 *)(out + row * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = v[q];
    }
}

template <int DT>
__global__ __launch_bounds__(256) void avgpool2(const typename T16<DT>::elem *__restrict__ x, typename T16<DT>::elem *__restrict__ out,
                                                int B, int H, int W, int C) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    const int Ho = H / 2, Wo = W / 2, cv = C / 8;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv);
        int64_t t = i / cv;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const E *p00 = x + ((((int64_t)b * H + 2 * ho) * W + 2 * wo) * C) + c * 8;
        const vec8 a = *(const vec8 *)p00, bq = *(const vec8 *)(p00 + C);
        const vec8 cq = *(const vec8 *)(p00 + (int64_t)W * C), d = *(const vec8 *)(p00 + (int64_t)W * C + C);
        vec8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (E)(((float)a[e] + (float)bq[e] + (float)cq[e] + (float)d[e]) * 0.25f);
        *(vec8 *)(out + i * 8) = o;
    }
}

// one thread per (b, 8-channel chunk): mean over the S*S cells, then all S*S+1 tokens (+ positional)
template <int DT>
__global__ __launch_bounds__(256) void attnpool_tokens(const typename T16<DT>::elem *__restrict__ x, const float *__restrict__ pos,
                                                       typename T16<DT>::elem *__restrict__ out, int B, int S, int C) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    const int cv = C / 8, G = S * S, L = G + 1;
    const int64_t total = (int64_t)B * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv), b = (int)(i / cv);
        float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < G; ++g) {
            const vec8 v = *(const vec8 *)(x + ((int64_t)b * G + g) * C + c * 8);
            vec8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                sum[e] += f;
                o[e] = (E)(f + pos[(int64_t)(g + 1) * C + c * 8 + e]);
            }
            *(vec8 *)(out + ((int64_t)b * L + g + 1) * C + c * 8) = o;
        }
        vec8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (E)(sum[e] / (float)G + pos[c * 8 + e]);
        *(vec8 *)(out + (int64_t)b * L * C + c * 8) = o;
    }
}

// one wave per (b, head): lane j scores keys j, j + 64, ... (up to 4 per lane, L <= 256), wave softmax, then lane d
// accumulates head dim d
template <int DT>
__global__ __launch_bounds__(256) void attnpool_attend(const float *__restrict__ q, const typename T16<DT>::elem *__restrict__ k,
                                                       const typename T16<DT>::elem *__restrict__ v, typename T16<DT>::elem *__restrict__ out,
                                                       int B, int L, int H) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= B * H) return;
    const int b = wid / H, h = wid - b * H;
    const int64_t Ew = (int64_t)H * 64;
    const float *qh = q + (int64_t)b * Ew + h * 64;
    const int nq = (L + 63) >> 6;                 // wave-uniform
    float s[4], pe[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = -INFINITY;
        const int j = t * 64 + lane;
        if (t < nq && j < L) {
            const E *kr = k + ((int64_t)b * L + j) * Ew + h * 64;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const vec8 kv = *(const vec8 *)(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += qh[c * 8 + e] * (float)kv[e];
            }
            s[t] = acc * 0.125f;
        }
        mx = fmaxf(mx, s[t]);
    }
    mx = wave_max(mx);
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        pe[t] = (t * 64 + lane < L) ? __expf(s[t] - mx) : 0.f;
        ps += pe[t];
    }
    const float inv = 1.0f / wave_sum(ps);
    float o = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= nq) break;
        const int n = min(64, L - t * 64);
        for (int j = 0; j < n; ++j) {
            const float pj = __shfl(pe[t], j);
            o += pj * (float)v[((int64_t)b * L + t * 64 + j) * Ew + h * 64 + lane];
        }
    }
    out[(int64_t)b * Ew + h * 64 + lane] = (E)(o * inv);
}

unsigned grid_for(int64_t total) { return (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384); }

}  // namespace


extern "C" int hgr_stem_im2col(const float *image, void *out, int B, int R, int dtype, void *stream) {
    HGR_REQUIRE(image && out && B >= 1 && R >= 2, "hgr_stem_im2col: bad arguments");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_stem_im2col: bad dtype %d", dtype);
    const int Ho = (R - 1) / 2 + 1;
    const int64_t total = (int64_t)B * Ho * Ho;
    if (dtype == HGR_BF16) hipLaunchKernelGGL((stem_im2col<HGR_BF16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, image, (__bf16 *)out, B, R);
    else hipLaunchKernelGGL((stem_im2col<HGR_F16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, image, (_Float16 *)out, B, R);
    HGR_CHECK_LAUNCH("hgr_stem_im2col");
    return HGR_OK;
}

extern "C" int hgr_avgpool2_nhwc(const void *x, void *out, int B, int H, int W, int C, int dtype, void *stream) {
    HGR_REQUIRE(x && out && B >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C >= 8 && C % 8 == 0, "hgr_avgpool2_nhwc: bad geometry B=%d H=%d W=%d C=%d", B, H, W, C);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(out, 16), "hgr_avgpool2_nhwc: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_avgpool2_nhwc: bad dtype %d", dtype);
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((avgpool2<HGR_BF16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, (__bf16 *)out, B, H, W, C);
    else hipLaunchKernelGGL((avgpool2<HGR_F16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x, (_Float16 *)out, B, H, W, C);
    HGR_CHECK_LAUNCH("hgr_avgpool2_nhwc");
    return HGR_OK;
}

extern "C" int hgr_attnpool_tokens(const void *x, const float *pos, void *out, int B, int S, int C, int dtype, void *stream) {
    HGR_REQUIRE(x && pos && out && B >= 1 && S >= 1 && C >= 8 && C % 8 == 0, "hgr_attnpool_tokens: bad arguments");
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(out, 16), "hgr_attnpool_tokens: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_attnpool_tokens: bad dtype %d", dtype);
    const int64_t total = (int64_t)B * (C / 8);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((attnpool_tokens<HGR_BF16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, pos, (__bf16 *)out, B, S, C);
    else hipLaunchKernelGGL((attnpool_tokens<HGR_F16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x, pos, (_Float16 *)out, B, S, C);
    HGR_CHECK_LAUNCH("hgr_attnpool_tokens");
    return HGR_OK;
}

extern "C" int hgr_attnpool_attend(const float *q, const void *k, const void *v, void *out, int B, int L, int heads, int dtype, void *stream) {
    HGR_REQUIRE(q && k && v && out && B >= 1 && heads >= 1 && L >= 1 && L <= 256, "hgr_attnpool_attend: bad arguments (L <= 256)");
    HGR_REQUIRE(hgr_aligned(k, 16) && hgr_aligned(v, 2) && hgr_aligned(q, 4), "hgr_attnpool_attend: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_attnpool_attend: bad dtype %d", dtype);
    const unsigned blocks = (unsigned)(((int64_t)B * heads + 3) / 4);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((attnpool_attend<HGR_BF16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, q, (const __bf16 *)k, (const __bf16 *)v, (__bf16 *)out, B, L, heads);
    else hipLaunchKernelGGL((attnpool_attend<HGR_F16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, q, (const _Float16 *)k, (const _Float16 *)v, (_Float16 *)out, B, L, heads);
    HGR_CHECK_LAUNCH("hgr_attnpool_attend");
    return HGR_OK;
}
