// Input-side gathers: ViT patch extraction (im2col for the kernel = stride conv), text token
// embedding and the EOT position.  All HBM-bound; reads and writes are coalesced 16-byte accesses
// where the shapes allow it.
#include "hgr_common.h"

namespace {

// One thread moves 8 consecutive pixels of one image row: 32 B read, 16 B written.
// Threads are enumerated in IMAGE order (b, c, y, x8) so reads are perfectly coalesced; the P/8
// threads of one patch row write P*2 contiguous bytes of the patch row in `out`.
template <int DT>
__global__ __launch_bounds__(256) void im2col_vec8(const float *__restrict__ img, typename T16<DT>::elem *__restrict__ out,
                                                   int B, int R, int P, int Kp, int rpi, int roff) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    const int xr = R >> 3;
    const int64_t total = (int64_t)B * 3 * R * xr;
    const int g = R / P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x8 = (int)(i % xr);
        int64_t t = i / xr;
        const int y = (int)(t % R); t /= R;
        const int c = (int)(t % 3);
        const int b = (int)(t / 3);
        const f32x4 *src = (const f32x4 *)(img + (((int64_t)b * 3 + c) * R + y) * R + x8 * 8);
        const f32x4 a = src[0], d = src[1];
        vec8 v;
        v[0] = (E)a[0]; v[1] = (E)a[1]; v[2] = (E)a[2]; v[3] = (E)a[3];
        v[4] = (E)d[0]; v[5] = (E)d[1]; v[6] = (E)d[2]; v[7] = (E)d[3];
        const int x = x8 * 8, gy = y / P, py = y - gy * P, gx = x / P, px = x - gx * P;
        const int64_t row = (int64_t)b * rpi + roff + gy * g + gx;
        *(vec8 *)(out + row * Kp + (c * P + py) * P + px) = v;
    }
}

// Generic fallback (P % 8 != 0, e.g. ViT-L/14) and zero fill of the K padding: one thread per output element.
template <int DT>
__global__ __launch_bounds__(256) void im2col_scalar(const float *__restrict__ img, typename T16<DT>::elem *__restrict__ out,
                                                     int B, int R, int P, int Kp, int kfirst, int rpi, int roff) {
    typedef typename T16<DT>::elem E;
    const int g = R / P, K = 3 * P * P, span = Kp - kfirst;
    const int64_t total = (int64_t)B * g * g * span;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = kfirst + (int)(i % span);
        const int64_t row = i / span;
        float v = 0.f;
        if (k < K) {
            const int gx = (int)(row % g), gy = (int)((row / g) % g), b = (int)(row / ((int64_t)g * g));
            const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
            v = img[(((int64_t)b * 3 + c) * R + gy * P + py) * R + gx * P + px];
        }
        out[(((int64_t)(row / ((int64_t)g * g))) * rpi + roff + row % ((int64_t)g * g)) * Kp + k] = (E)v;
    }
}

__global__ __launch_bounds__(256) void text_embed(const int64_t *__restrict__ tokens, int64_t ldt, const float *__restrict__ emb,
                                                  const float *__restrict__ pos, float *__restrict__ x, int n, int L, int W, int vocab) {
    const int nv = W >> 2;
    const int64_t total = (int64_t)n * L * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % nv);
        const int64_t row = i / nv;
        const int t = (int)(row % L);
        const int64_t s = row / L;
        int64_t id = tokens[s * ldt + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 e = ((const f32x4 *)(emb + id * W))[c];
        const f32x4 p = ((const f32x4 *)(pos + (int64_t)t * W))[c];
        ((f32x4 *)(x + row * W))[c] = e + p;
    }
}

// one wave per prompt: first position of the maximum token id
__global__ __launch_bounds__(256) void eot_index(const int64_t *__restrict__ tokens, int64_t ldt, int32_t *__restrict__ eot, int n, int ctx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    long long best = -0x7fffffffffffffffLL - 1;
    int bi = 0x7fffffff;
    for (int t = lane; t < ctx; t += 64) {
        const long long v = tokens[(int64_t)row * ldt + t];
        if (v > best) { best = v; bi = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) eot[row] = bi;
}

// uint8 NHWC crops -> normalised 16-bit patch rows in (py, px, c) order: patch row py of patch (gy, gx) is the
// contiguous 3*P bytes  image[b, gy*P + py, gx*P .. gx*P + P - 1, 0..2];  out = v * a[c] + b[c] with
// a = 1/(255 std), b = -mean/std (ToTensor + Normalize of clip/clip.py:71-78 folded).  One thread = 4 pixels
// = 12 bytes in (3 dword loads), 12 values out (24 B).
template <int DT>
__global__ __launch_bounds__(256) void im2col_u8(const unsigned char *__restrict__ img, typename T16<DT>::elem *__restrict__ out,
                                                 int B, int R, int P, int Kp, int rpi, int roff, float a0, float a1, float a2, float b0, float b1, float b2) {
    typedef typename T16<DT>::elem E;
    const int g = R / P, q = R / 4;
    const int64_t total = (int64_t)B * R * q;
    const float sa[3] = {a0, a1, a2}, sb[3] = {b0, b1, b2};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x4 = (int)(i % q);
        const int y = (int)((i / q) % R);
        const int b = (int)(i / ((int64_t)q * R));
        const unsigned int *src = (const unsigned int *)(img + (((int64_t)b * R + y) * R + x4 * 4) * 3);
        const unsigned int w0 = src[0], w1 = src[1], w2 = src[2];
        unsigned char px[12];
#pragma unroll
        for (int e = 0; e < 4; ++e) { px[e] = (w0 >> (8 * e)) & 255; px[4 + e] = (w1 >> (8 * e)) & 255; px[8 + e] = (w2 >> (8 * e)) & 255; }
        const int x = x4 * 4, gy = y / P, py = y - gy * P, gx = x / P, pxo = x - gx * P;
        E *o = out + ((int64_t)b * rpi + roff + gy * g + gx) * Kp + (py * P + pxo) * 3;
#pragma unroll
        for (int e = 0; e < 12; ++e) o[e] = (E)((float)px[e] * sa[e % 3] + sb[e % 3]);
    }
}

// t[b, l] = pe[b, l] + pos[l] + (l == 0 ? cls : 0) in place on the [B*L, W] patch-GEMM output whose class rows are zero
__global__ __launch_bounds__(256) void vit_assemble(float *__restrict__ t, const float *__restrict__ cls, const float *__restrict__ pos, int B, int L, int W) {
    const int nv = W >> 2;
    const int64_t total = (int64_t)B * L * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % nv);
        const int l = (int)((i / nv) % L);
        f32x4 v = ((f32x4 *)t)[i] + ((const f32x4 *)(pos + (int64_t)l * W))[c];
        if (l == 0) v += ((const f32x4 *)cls)[c];
        ((f32x4 *)t)[i] = v;
    }
}

}  // namespace

extern "C" int hgr_vit_assemble(float *t, const float *class_embedding, const float *positional_embedding, int B, int L, int W, void *stream) {
    HGR_REQUIRE(t && class_embedding && positional_embedding && B >= 1 && L >= 1 && W >= 4 && W % 4 == 0, "hgr_vit_assemble: bad arguments");
    const int64_t total = (int64_t)B * L * (W / 4);
    hipLaunchKernelGGL(vit_assemble, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, t, class_embedding, positional_embedding, B, L, W);
    HGR_CHECK_LAUNCH("hgr_vit_assemble");
    return HGR_OK;
}

namespace {
}

extern "C" int hgr_im2col_patches_ex(const float *image, void *out, int B, int R, int P, int Kp, int rows_per_image, int row_offset, int dtype, void *stream);

extern "C" int hgr_im2col_patches(const float *image, void *out, int B, int R, int P, int Kp, int dtype, void *stream) {
    return hgr_im2col_patches_ex(image, out, B, R, P, Kp, (R / (P > 0 ? P : 1)) * (R / (P > 0 ? P : 1)), 0, dtype, stream);
}

extern "C" int hgr_im2col_patches_ex(const float *image, void *out, int B, int R, int P, int Kp, int rows_per_image, int row_offset, int dtype, void *stream) {
    HGR_REQUIRE(image && out, "hgr_im2col_patches: null operand");
    HGR_REQUIRE(B >= 1 && P >= 1 && R >= P && R % P == 0, "hgr_im2col_patches: B=%d R=%d P=%d unsupported", B, R, P);
    HGR_REQUIRE(row_offset >= 0 && rows_per_image >= row_offset + (R / P) * (R / P), "hgr_im2col_patches_ex: bad row mapping");
    const int rpi = rows_per_image, roff = row_offset;
    const int K = 3 * P * P;
    HGR_REQUIRE(Kp >= K && Kp % 64 == 0, "hgr_im2col_patches: Kp=%d must be >= %d and a multiple of 64", Kp, K);
    HGR_REQUIRE(hgr_aligned(image, 16) && hgr_aligned(out, 16), "hgr_im2col_patches: operands must be 16-byte aligned");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_im2col_patches: bad dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    const int g = R / P;
    if (P % 8 == 0) {
        const int64_t total = (int64_t)B * 3 * R * (R / 8);
        const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        if (dtype == HGR_BF16) hipLaunchKernelGGL((im2col_vec8<HGR_BF16>), dim3(blocks), dim3(256), 0, s, image, (__bf16 *)out, B, R, P, Kp, rpi, roff);
        else hipLaunchKernelGGL((im2col_vec8<HGR_F16>), dim3(blocks), dim3(256), 0, s, image, (_Float16 *)out, B, R, P, Kp, rpi, roff);
    }
    const int kfirst = (P % 8 == 0) ? K : 0;     // scalar kernel: everything, or just the zero padding
    if (Kp > kfirst) {
        const int64_t total = (int64_t)B * g * g * (Kp - kfirst);
        const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        if (dtype == HGR_BF16) hipLaunchKernelGGL((im2col_scalar<HGR_BF16>), dim3(blocks), dim3(256), 0, s, image, (__bf16 *)out, B, R, P, Kp, kfirst, rpi, roff);
        else hipLaunchKernelGGL((im2col_scalar<HGR_F16>), dim3(blocks), dim3(256), 0, s, image, (_Float16 *)out, B, R, P, Kp, kfirst, rpi, roff);
    }
    HGR_CHECK_LAUNCH("hgr_im2col_patches");
    return HGR_OK;
}

extern "C" int hgr_text_embed(const int64_t *tokens, int64_t ld_tokens, const float *token_embedding,
                              const float *positional_embedding, float *x, int n, int L, int W, int vocab, void *stream) {
    HGR_REQUIRE(tokens && token_embedding && positional_embedding && x, "hgr_text_embed: null operand");
    HGR_REQUIRE(n >= 1 && L >= 1 && ld_tokens >= L && W >= 4 && W % 4 == 0 && vocab >= 1, "hgr_text_embed: n=%d L=%d ld=%lld W=%d vocab=%d unsupported", n, L, (long long)ld_tokens, W, vocab);
    HGR_REQUIRE(hgr_aligned(token_embedding, 16) && hgr_aligned(positional_embedding, 16) && hgr_aligned(x, 16) && hgr_aligned(tokens, 8), "hgr_text_embed: misaligned operand");
    const int64_t total = (int64_t)n * L * (W / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(text_embed, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tokens, ld_tokens, token_embedding, positional_embedding, x, n, L, W, vocab);
    HGR_CHECK_LAUNCH("hgr_text_embed");
    return HGR_OK;
}

extern "C" int hgr_eot_index(const int64_t *tokens, int64_t ld_tokens, int32_t *eot, int n, int ctx, void *stream) {
    HGR_REQUIRE(tokens && eot, "hgr_eot_index: null operand");
    HGR_REQUIRE(n >= 1 && ctx >= 1 && ld_tokens >= ctx, "hgr_eot_index: n=%d ctx=%d ld=%lld unsupported", n, ctx, (long long)ld_tokens);
    hipLaunchKernelGGL(eot_index, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, tokens, ld_tokens, eot, n, ctx);
    HGR_CHECK_LAUNCH("hgr_eot_index");
    return HGR_OK;
}

extern "C" int hgr_im2col_patches_u8(const unsigned char *image, void *out, int B, int R, int P, int Kp, int rows_per_image, int row_offset,
                                     const float *mean3, const float *std3, int dtype, void *stream) {
    HGR_REQUIRE(image && out && mean3 && std3, "hgr_im2col_patches_u8: null operand");
    HGR_REQUIRE(B >= 1 && P >= 4 && P % 4 == 0 && R >= P && R % P == 0, "hgr_im2col_patches_u8: B=%d R=%d P=%d unsupported (P %% 4 == 0)", B, R, P);
    HGR_REQUIRE(Kp == 3 * P * P, "hgr_im2col_patches_u8: Kp must equal 3*P*P (no K padding on this path)");
    HGR_REQUIRE(row_offset >= 0 && rows_per_image >= row_offset + (R / P) * (R / P), "hgr_im2col_patches_u8: bad row mapping");
    HGR_REQUIRE(hgr_aligned(image, 4) && hgr_aligned(out, 8), "hgr_im2col_patches_u8: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_im2col_patches_u8: bad dtype %d", dtype);
    float a[3], b[3];
    for (int c = 0; c < 3; ++c) { a[c] = 1.0f / (255.0f * std3[c]); b[c] = -mean3[c] / std3[c]; }      // host pointers
    const int64_t total = (int64_t)B * R * (R / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((im2col_u8<HGR_BF16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, image, (__bf16 *)out, B, R, P, Kp, rows_per_image, row_offset, a[0], a[1], a[2], b[0], b[1], b[2]);
    else hipLaunchKernelGGL((im2col_u8<HGR_F16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, image, (_Float16 *)out, B, R, P, Kp, rows_per_image, row_offset, a[0], a[1], a[2], b[0], b[1], b[2]);
    HGR_CHECK_LAUNCH("hgr_im2col_patches_u8");
    return HGR_OK;
}
