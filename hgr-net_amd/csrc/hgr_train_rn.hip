// Backward-pass helpers of the ModifiedResNet image tower (clip/model.py:10-150) for the OM training step
// (model/clip_tree.py:222-281, the reference's README trains --arch RN50).  The heavy lifting - every convolution's
// data and weight gradient - is GEMM work done by hgr_gemm.hip (1x1: NT products on transposed operands; 3x3 dX: the
// implicit-GEMM convolution with flipped weights; dW: split-K products over the pixel dimension).  This file holds the
// streaming kernels around them: all HBM-bound, 16-byte vectors, 16-bit NHWC activations.
//
// BatchNorm runs with running statistics even in training (clip_tree.py:46 keeps the model in eval()), so the forward
// uses conv weights with BN folded in (w' = w * gamma / sigma, b' = beta - mean * gamma / sigma: hgr_bn_fold) and the
// backward produces gradients of w' and b', which hgr_bn_unfold_grad maps back to conv.weight, bn.weight and bn.bias.
#include "hgr_common.h"

namespace {

unsigned grid_for(int64_t total) { return (unsigned)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768); }

// dy *= (y > 0)   (ReLU backward from the saved OUTPUT), in place or to `out`
template <int DT>
__global__ __launch_bounds__(256) void relu_bwd16(const typename T16<DT>::elem *__restrict__ dy, const typename T16<DT>::elem *__restrict__ y,
                                                  typename T16<DT>::elem *__restrict__ out, int64_t n8) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        vec8 d = ((const vec8 *)dy)[i];
        const vec8 v = ((const vec8 *)y)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (float)v[e] > 0.f ? d[e] : (E)0.f;
        ((vec8 *)out)[i] = d;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void add16(const typename T16<DT>::elem *__restrict__ a, const typename T16<DT>::elem *__restrict__ b,
                                             typename T16<DT>::elem *__restrict__ out, int64_t n8) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const vec8 x = ((const vec8 *)a)[i], y = ((const vec8 *)b)[i];
        vec8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (E)((float)x[e] + (float)y[e]);
        ((vec8 *)out)[i] = r;
    }
}

// AvgPool2d(2) backward: dx[b, h, w, :] = dy[b, h/2, w/2, :] / 4
template <int DT>
__global__ __launch_bounds__(256) void avgpool2_bwd(const typename T16<DT>::elem *__restrict__ dy, typename T16<DT>::elem *__restrict__ dx,
                                                    int B, int H, int W, int C) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    const int c8 = C / 8, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * H * W * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c8);
        const int64_t pix = i / c8;
        const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        vec8 v = ((const vec8 *)dy)[(((int64_t)b * Ho + h / 2) * Wo + w / 2) * c8 + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (E)((float)v[e] * 0.25f);
        ((vec8 *)dx)[i] = v;
    }
}

// attention-pool token assembly backward (clip/model.py:67-69): tokens = cat(mean(x), x) + pos
//   dx[b, p, :] = dtok[b, 1 + p, :] + dtok[b, 0, :] / S      (dpos is a column sum of dtok, done by hgr_colsum)
template <int DT>
__global__ __launch_bounds__(256) void attnpool_tokens_bwd(const typename T16<DT>::elem *__restrict__ dtok, typename T16<DT>::elem *__restrict__ dx,
                                                           int B, int S, int C) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    const int c8 = C / 8;
    const int64_t total = (int64_t)B * S * c8;
    const float inv = 1.f / (float)S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c8);
        const int p = (int)((i / c8) % S), b = (int)(i / ((int64_t)c8 * S));
        const vec8 m = ((const vec8 *)dtok)[((int64_t)b * (S + 1)) * c8 + c];
        const vec8 t = ((const vec8 *)dtok)[((int64_t)b * (S + 1) + 1 + p) * c8 + c];
        vec8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (E)((float)t[e] + (float)m[e] * inv);
        ((vec8 *)dx)[i] = r;
    }
}

// Transposed im2col of a 3x3 / pad 1 / stride 1 convolution from the TRANSPOSED activation xt [C, ld] (column m = pixel
// (b, h, w)):  out[(t * C + c), m] = xt[c, m + (ky - 1) * W + (kx - 1)] if that pixel is inside the image, else 0.
// Rows are contiguous in m, so reads and writes are coalesced; the 9 shifted reads of a row hit L2.  Columns m >= M
// (the padding up to ld) are written as zeros.
template <int DT>
__global__ __launch_bounds__(256) void im2col3x3_t(const typename T16<DT>::elem *__restrict__ xt, typename T16<DT>::elem *__restrict__ out,
                                                   int H, int W, int C, int64_t M, int64_t ld) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    const int row = blockIdx.y;                  // t * C + c
    const int t = row / C, c = row - t * C;
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const E *src = xt + (int64_t)c * ld;
    E *dst = out + (int64_t)row * ld;
    const int64_t shift = (int64_t)dy * W + dx;
    // 8 consecutive pixels per thread: the shifted source is 2-byte misaligned (scalar loads, served by L1), the store is 16 B
    for (int64_t m0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; m0 < ld; m0 += (int64_t)gridDim.x * 2048) {
        vec8 v;
        int w = (int)(m0 % W), h = (int)((m0 / W) % H);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t m = m0 + e;
            const int hh = h + dy, ww = w + dx;
            v[e] = (m < M && hh >= 0 && hh < H && ww >= 0 && ww < W) ? src[m + shift] : (E)0.f;
            if (++w == W) { w = 0; if (++h == H) h = 0; }
        }
        *(vec8 *)(dst + m0) = v;
    }
}

// w16[co, (ky, kx, ci)] = w[co, ci, ky, kx] * s[co],  s = gamma / sqrt(var + eps);  b[co] = beta - mean * s
template <int DT>
__global__ __launch_bounds__(256) void bn_fold(const float *__restrict__ w, const float *__restrict__ gamma, const float *__restrict__ beta,
                                               const float *__restrict__ mean, const float *__restrict__ var, float eps,
                                               typename T16<DT>::elem *__restrict__ w16, float *__restrict__ bias, int Cin, int khw, int Kp) {
    typedef typename T16<DT>::elem E;
    const int co = blockIdx.x, K = Cin * khw;
    const float s = gamma[co] / sqrtf(var[co] + eps);
    if (threadIdx.x == 0) bias[co] = beta[co] - mean[co] * s;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        float v = 0.f;
        if (k < K) {
            const int t = k / Cin, ci = k - t * Cin;
            v = w[((int64_t)co * Cin + ci) * khw + t] * s;
        }
        w16[(int64_t)co * Kp + k] = (E)v;
    }
}

// from the gradients of the folded weight / bias back to the parameters:
//   g_w[co, ci, t] += gwf[co, (t, ci)] * s        g_gamma[co] += (sum_k gwf[co, k] w[co, k] - gbf[co] mean[co]) / sigma
//   g_beta[co] += gbf[co]
__global__ __launch_bounds__(256) void bn_unfold_grad(const float *__restrict__ gwf, int64_t ldg, const float *__restrict__ gbf,
                                                      const float *__restrict__ w, const float *__restrict__ gamma,
                                                      const float *__restrict__ mean, const float *__restrict__ var, float eps,
                                                      float *__restrict__ g_w, float *__restrict__ g_gamma, float *__restrict__ g_beta,
                                                      int Cin, int khw) {
    __shared__ float red[4];
    const int co = blockIdx.x, K = Cin * khw;
    const float inv_sigma = 1.f / sqrtf(var[co] + eps);
    const float s = gamma[co] * inv_sigma;
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const int t = k / Cin, ci = k - t * Cin;
        const int64_t wi = ((int64_t)co * Cin + ci) * khw + t;
        const float g = gwf[(int64_t)co * ldg + k];
        dot += g * w[wi];
        g_w[wi] += g * s;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = red[0] + red[1] + red[2] + red[3];
        g_gamma[co] += (tot - gbf[co] * mean[co]) * inv_sigma;
        g_beta[co] += gbf[co];
    }
}

}  // namespace

extern "C" int hgr_relu_bwd16(const void *dy, const void *y, void *out, int64_t n, int dtype, void *stream) {
    HGR_REQUIRE(dy && y && out && n >= 8 && n % 8 == 0, "hgr_relu_bwd16: n=%lld must be a positive multiple of 8", (long long)n);
    HGR_REQUIRE(hgr_aligned(dy, 16) && hgr_aligned(y, 16) && hgr_aligned(out, 16) && (dtype == HGR_BF16 || dtype == HGR_F16), "hgr_relu_bwd16: bad operand");
    if (dtype == HGR_BF16) hipLaunchKernelGGL((relu_bwd16<HGR_BF16>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)dy, (const __bf16 *)y, (__bf16 *)out, n / 8);
    else hipLaunchKernelGGL((relu_bwd16<HGR_F16>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)dy, (const _Float16 *)y, (_Float16 *)out, n / 8);
    HGR_CHECK_LAUNCH("hgr_relu_bwd16");
    return HGR_OK;
}

extern "C" int hgr_add16(const void *a, const void *b, void *out, int64_t n, int dtype, void *stream) {
    HGR_REQUIRE(a && b && out && n >= 8 && n % 8 == 0, "hgr_add16: n=%lld must be a positive multiple of 8", (long long)n);
    HGR_REQUIRE(hgr_aligned(a, 16) && hgr_aligned(b, 16) && hgr_aligned(out, 16) && (dtype == HGR_BF16 || dtype == HGR_F16), "hgr_add16: bad operand");
    if (dtype == HGR_BF16) hipLaunchKernelGGL((add16<HGR_BF16>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)a, (const __bf16 *)b, (__bf16 *)out, n / 8);
    else hipLaunchKernelGGL((add16<HGR_F16>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)a, (const _Float16 *)b, (_Float16 *)out, n / 8);
    HGR_CHECK_LAUNCH("hgr_add16");
    return HGR_OK;
}

extern "C" int hgr_avgpool2_bwd_nhwc(const void *dy, void *dx, int B, int H, int W, int C, int dtype, void *stream) {
    HGR_REQUIRE(dy && dx && B >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C >= 8 && C % 8 == 0, "hgr_avgpool2_bwd_nhwc: bad geometry B=%d H=%d W=%d C=%d", B, H, W, C);
    HGR_REQUIRE(hgr_aligned(dy, 16) && hgr_aligned(dx, 16) && (dtype == HGR_BF16 || dtype == HGR_F16), "hgr_avgpool2_bwd_nhwc: bad operand");
    const int64_t total = (int64_t)B * H * W * (C / 8);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((avgpool2_bwd<HGR_BF16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)dy, (__bf16 *)dx, B, H, W, C);
    else hipLaunchKernelGGL((avgpool2_bwd<HGR_F16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)dy, (_Float16 *)dx, B, H, W, C);
    HGR_CHECK_LAUNCH("hgr_avgpool2_bwd_nhwc");
    return HGR_OK;
}

extern "C" int hgr_attnpool_tokens_bwd(const void *dtok, void *dx, int B, int S, int C, int dtype, void *stream) {
    HGR_REQUIRE(dtok && dx && B >= 1 && S >= 1 && C >= 8 && C % 8 == 0, "hgr_attnpool_tokens_bwd: bad arguments");
    HGR_REQUIRE(hgr_aligned(dtok, 16) && hgr_aligned(dx, 16) && (dtype == HGR_BF16 || dtype == HGR_F16), "hgr_attnpool_tokens_bwd: bad operand");
    const int64_t total = (int64_t)B * S * (C / 8);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((attnpool_tokens_bwd<HGR_BF16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)dtok, (__bf16 *)dx, B, S, C);
    else hipLaunchKernelGGL((attnpool_tokens_bwd<HGR_F16>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)dtok, (_Float16 *)dx, B, S, C);
    HGR_CHECK_LAUNCH("hgr_attnpool_tokens_bwd");
    return HGR_OK;
}

extern "C" int hgr_im2col3x3_t(const void *xt, void *out, int B, int H, int W, int C, int64_t ld, int dtype, void *stream) {
    HGR_REQUIRE(xt && out && B >= 1 && H >= 1 && W >= 1 && C >= 1 && 9 * C <= 65535, "hgr_im2col3x3_t: bad geometry B=%d H=%d W=%d C=%d", B, H, W, C);
    const int64_t M = (int64_t)B * H * W;
    HGR_REQUIRE(ld >= M && ld % 8 == 0 && hgr_aligned(out, 16) && (dtype == HGR_BF16 || dtype == HGR_F16), "hgr_im2col3x3_t: ld=%lld must be >= B*H*W and a multiple of 8, out 16-byte aligned", (long long)ld);
    const dim3 grid((unsigned)((ld + 2047) / 2048 < 4096 ? (ld + 2047) / 2048 : 4096), (unsigned)(9 * C));
    if (dtype == HGR_BF16) hipLaunchKernelGGL((im2col3x3_t<HGR_BF16>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)xt, (__bf16 *)out, H, W, C, M, ld);
    else hipLaunchKernelGGL((im2col3x3_t<HGR_F16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16 *)xt, (_Float16 *)out, H, W, C, M, ld);
    HGR_CHECK_LAUNCH("hgr_im2col3x3_t");
    return HGR_OK;
}

extern "C" int hgr_bn_fold(const float *w, const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                           void *w16, float *bias, int Cout, int Cin, int khw, int Kp, int dtype, void *stream) {
    HGR_REQUIRE(w && gamma && beta && mean && var && w16 && bias, "hgr_bn_fold: null operand");
    HGR_REQUIRE(Cout >= 1 && Cin >= 1 && (khw == 1 || khw == 9) && Kp >= Cin * khw && (dtype == HGR_BF16 || dtype == HGR_F16), "hgr_bn_fold: bad shape Cout=%d Cin=%d khw=%d Kp=%d", Cout, Cin, khw, Kp);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((bn_fold<HGR_BF16>), dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, mean, var, eps, (__bf16 *)w16, bias, Cin, khw, Kp);
    else hipLaunchKernelGGL((bn_fold<HGR_F16>), dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, mean, var, eps, (_Float16 *)w16, bias, Cin, khw, Kp);
    HGR_CHECK_LAUNCH("hgr_bn_fold");
    return HGR_OK;
}

extern "C" int hgr_bn_unfold_grad(const float *gwf, int64_t ldg, const float *gbf, const float *w, const float *gamma, const float *mean,
                                  const float *var, float eps, float *g_w, float *g_gamma, float *g_beta, int Cout, int Cin, int khw,
                                  void *stream) {
    HGR_REQUIRE(gwf && gbf && w && gamma && mean && var && g_w && g_gamma && g_beta, "hgr_bn_unfold_grad: null operand");
    HGR_REQUIRE(Cout >= 1 && Cin >= 1 && (khw == 1 || khw == 9) && ldg >= (int64_t)Cin * khw, "hgr_bn_unfold_grad: bad shape");
    hipLaunchKernelGGL(bn_unfold_grad, dim3(Cout), dim3(256), 0, (hipStream_t)stream, gwf, ldg, gbf, w, gamma, mean, var, eps, g_w, g_gamma, g_beta, Cin, khw);
    HGR_CHECK_LAUNCH("hgr_bn_unfold_grad");
    return HGR_OK;
}
