// Input pipeline: Resize(n_px, BICUBIC) + CenterCrop(n_px) (+ ToTensor + Normalize) of a batch of decoded RGB images,
// bit-exact with the reference's torchvision / Pillow transform (clip/clip.py:71-78, dataset/imagenet_group.py:27-34).
//
// Pillow's ImagingResample (libImaging/Resample.c) is a separable convolution in 22-bit fixed point: a horizontal pass
// whose result is clipped to 8 bits, then a vertical pass over those bytes, clipped again.  The taps depend only on
// (input size, output size), so the host computes them in double precision exactly as Pillow does and ships them as
// int32 tables; everything on the device is integer arithmetic and therefore reproducible to the bit.
//
// One workgroup = a 16 x 16 tile of the CROPPED output of one image.  It walks the source rows its tile reads in chunks
// of 16: phase A convolves those rows horizontally for the tile's 16 columns into LDS (one (row, column) per thread, the
// 8-bit clipped intermediate of the two-pass algorithm), phase B accumulates them into each thread's output pixel with
// its vertical taps.  Source bytes are read once per tile column range (L1/L2 serve the overlap between neighbouring
// tiles); the kernel is bound by those byte reads and is far below the HBM roof at batch sizes that matter.
#include "hgr_common.h"

namespace {

constexpr int PB = 22;      // Resample.c PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ int clip8(int v) { return min(max(v >> PB, 0), 255); }

// horizontal taps of one (source row, output column): the row's bytes are fetched as aligned dwords and realigned with
// v_alignbyte (12 bytes = 3 dwords = 4 RGB taps per step) instead of 3 byte loads per tap.  The tap table is zero padded
// to a multiple of 4 (host), so the bytes read past the last live tap - at most 15, inside the caller's slack - are
// multiplied by 0.
__device__ __forceinline__ void hconv(const unsigned char *p, const int *kx, int xn, int &h0, int &h1, int &h2) {
    const uintptr_t addr = (uintptr_t)p;
    const unsigned *q = (const unsigned *)(addr & ~(uintptr_t)3);
    const unsigned sh = (unsigned)(addr & 3);
    unsigned carry = q[0];
    for (int t = 0; t < xn; t += 4) {
        const unsigned w1 = q[1], w2 = q[2], w3 = q[3];
        const int4 k = *(const int4 *)(kx + t);
        // 12 bytes starting at p + 3 t
        const unsigned d0 = __builtin_amdgcn_alignbyte(w1, carry, sh);
        const unsigned d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
        const unsigned d2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
        h0 += (int)(d0 & 255u) * k.x;         h1 += (int)((d0 >> 8) & 255u) * k.x;  h2 += (int)((d0 >> 16) & 255u) * k.x;
        h0 += (int)(d0 >> 24) * k.y;          h1 += (int)(d1 & 255u) * k.y;         h2 += (int)((d1 >> 8) & 255u) * k.y;
        h0 += (int)((d1 >> 16) & 255u) * k.z; h1 += (int)(d1 >> 24) * k.z;          h2 += (int)(d2 & 255u) * k.z;
        h0 += (int)((d2 >> 8) & 255u) * k.w;  h1 += (int)((d2 >> 16) & 255u) * k.w; h2 += (int)(d2 >> 24) * k.w;
        carry = w3; q += 3;
    }
}

constexpr int CH = 32;      // source rows per chunk: two (row, column) items per thread in the horizontal pass

__global__ __launch_bounds__(256) void preprocess_bicubic(const unsigned char *__restrict__ src, const int64_t *__restrict__ off,
                                                          const int *__restrict__ hw, const int *__restrict__ tab, const int *__restrict__ xb,
                                                          const int *__restrict__ xk, int KX, const int *__restrict__ yb,
                                                          const int *__restrict__ yk, int KY, unsigned char *__restrict__ out_u8,
                                                          float *__restrict__ out_f32, float m0, float m1, float m2, float s0,
                                                          float s1, float s2, int R, int tiles) {
    __shared__ int hrow[CH][16][3];
    const int b = blockIdx.y;
    const int tb = tab ? tab[b] : b;                 // images of one size share one set of tap tables
    const int ty0 = (blockIdx.x / tiles) * 16, tx0 = (blockIdx.x % tiles) * 16;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int ox = min(tx0 + lx, R - 1), oy = min(ty0 + ly, R - 1);          // clamped: edge threads redo a valid pixel
    const int w = hw[b * 2 + 1];
    const unsigned char *img = src + off[b];
    const int *bx = xb + ((int64_t)tb * R + ox) * 2, *by = yb + ((int64_t)tb * R + oy) * 2;
    const int xmin = bx[0], xn = bx[1], ymin = by[0], yn = by[1];
    const int *kx = xk + ((int64_t)tb * R + ox) * KX, *ky = yk + ((int64_t)tb * R + oy) * KY;
    // source rows read by this tile: bounds are non-decreasing in the output row
    const int ylast = min(ty0 + 15, R - 1);
    const int r0 = yb[((int64_t)tb * R + ty0) * 2];
    const int r1 = yb[((int64_t)tb * R + ylast) * 2] + yb[((int64_t)tb * R + ylast) * 2 + 1];

    int a0 = 1 << (PB - 1), a1 = a0, a2 = a0;
    for (int rc = r0; rc < r1; rc += CH) {
#pragma unroll
        for (int half = 0; half < CH / 16; ++half) {
            const int row = rc + half * 16 + ly;
            if (row < r1) {
                int h0 = 1 << (PB - 1), h1 = h0, h2 = h0;
                hconv(img + ((int64_t)row * w + xmin) * 3, kx, xn, h0, h1, h2);
                hrow[half * 16 + ly][lx][0] = clip8(h0); hrow[half * 16 + ly][lx][1] = clip8(h1); hrow[half * 16 + ly][lx][2] = clip8(h2);
            }
        }
        __syncthreads();
        const int lo = max(rc, ymin), hi = min(min(rc + CH, r1), ymin + yn);
        for (int rr = lo; rr < hi; ++rr) {
            const int k = ky[rr - ymin];
            a0 += hrow[rr - rc][lx][0] * k; a1 += hrow[rr - rc][lx][1] * k; a2 += hrow[rr - rc][lx][2] * k;
        }
        __syncthreads();
    }
    if (tx0 + lx >= R || ty0 + ly >= R) return;
    const int c0 = clip8(a0), c1 = clip8(a1), c2 = clip8(a2);
    if (out_u8) {
        unsigned char *o = out_u8 + (((int64_t)b * R + oy) * R + ox) * 3;
        o[0] = (unsigned char)c0; o[1] = (unsigned char)c1; o[2] = (unsigned char)c2;
    }
    if (out_f32) {
        // ToTensor (/255) then Normalize ((x - mean) / std): two correctly rounded fp32 operations each, like the tensors
        float *o = out_f32 + ((int64_t)b * 3 * R + oy) * R + ox;
        const int64_t plane = (int64_t)R * R;
        o[0] = ((float)c0 / 255.0f - m0) / s0;
        o[plane] = ((float)c1 / 255.0f - m1) / s1;
        o[2 * plane] = ((float)c2 / 255.0f - m2) / s2;
    }
}

}  // namespace

extern "C" int hgr_preprocess_bicubic(const unsigned char *src, const int64_t *off, const int *hw, const int *tab, const int *xb, const int *xk,
                                      int KX, const int *yb, const int *yk, int KY, unsigned char *out_u8, float *out_f32,
                                      const float *mean3, const float *std3, int B, int R, void *stream) {
    HGR_REQUIRE(src && off && hw && xb && xk && yb && yk, "hgr_preprocess_bicubic: null operand");
    HGR_REQUIRE(out_u8 || out_f32, "hgr_preprocess_bicubic: no output requested");
    HGR_REQUIRE(B >= 1 && R >= 1 && KX >= 1 && KY >= 1, "hgr_preprocess_bicubic: bad shape B=%d R=%d KX=%d KY=%d", B, R, KX, KY);
    HGR_REQUIRE(B <= 65535, "hgr_preprocess_bicubic: B=%d exceeds the grid limit", B);
    HGR_REQUIRE(KX % 4 == 0 && hgr_aligned(xk, 16), "hgr_preprocess_bicubic: KX=%d must be a multiple of 4 and xk 16-byte aligned", KX);
    HGR_REQUIRE(!out_f32 || (mean3 && std3), "hgr_preprocess_bicubic: fp32 output needs mean and std");
    const int tiles = (R + 15) / 16;
    const float m0 = mean3 ? mean3[0] : 0.f, m1 = mean3 ? mean3[1] : 0.f, m2 = mean3 ? mean3[2] : 0.f;
    const float s0 = std3 ? std3[0] : 1.f, s1 = std3 ? std3[1] : 1.f, s2 = std3 ? std3[2] : 1.f;
    hipLaunchKernelGGL(preprocess_bicubic, dim3(tiles * tiles, B), dim3(256), 0, (hipStream_t)stream, src, off, hw, tab, xb, xk, KX, yb, yk, KY,
                       out_u8, out_f32, m0, m1, m2, s0, s1, s2, R, tiles);
    HGR_CHECK_LAUNCH("hgr_preprocess_bicubic");
    return HGR_OK;
}
