// Direct 3x3 / pad 1 / stride 1 convolution for 32 input channels (the ModifiedResNet stem's conv2 / conv3 at 112 x 112,
// clip/model.py:106-109, and the data gradient of conv2): out[b, y, x, co] = act(bias[co] + sum_{ky,kx,c} w * in).
//
// The implicit-GEMM loader of hgr_gemm.hip gathers every K-tile with 16-byte LDS-DMA pieces; with C = 32 a pixel-tap is
// only 64 contiguous bytes, every input byte is fetched 9 times through the texture path, and those two launches ran at
// 160 / 340 TFLOP/s although they are neither MFMA- nor HBM-bound.  Here a workgroup owns a 16 x 16 output tile of one
// image: the 18 x 18 x 32 input halo tile is brought in ONCE (LDS-DMA, zero page outside the image = the padding), the
// weights sit in LDS, and the nine taps are nine k-steps of v_mfma_f32_16x16x32 whose A fragment (16 pixels of a row x 32
// channels of one tap) is a single conflict-free ds_read_b128 per lane: pixel p's 64 bytes are 4 chunks, chunk ^= (p >> 2) & 3.
// Each wave computes 4 output rows x 16 columns x Cout; results leave through LDS as whole 16-byte chunks of contiguous
// NHWC rows.
#include "hgr_common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int dc_zero_page[4] = {0u, 0u, 0u, 0u};

constexpr int TP = 18;                         // halo tile edge
constexpr int IN_CHUNKS = TP * TP * 4;         // 1296 16-byte chunks
constexpr int IN_BYTES = 21 * 1024;            // 21 wave-instructions of LDS-DMA (1344 chunks), the tail is scratch
constexpr int WROW = 592;                      // weight row stride in LDS: 288 x 2 B + 16 B pad -> conflict-free b128 reads

struct DcArgs {
    const char *x; const char *w; const float *bias; char *out;
    int B, H, W, Kp, tiles_x, tiles_y, relu;
};

template <int DT, int NOUT>                    // NOUT = Cout / 16
__global__ __launch_bounds__(256) void conv3x3_c32(DcArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    constexpr int COUT = NOUT * 16;
    __shared__ __attribute__((aligned(1024))) char smem[IN_BYTES + COUT * WROW];
    char *sIn = smem, *sW = smem + IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int y0 = ty * 16, x0 = tx * 16;
    const char *img = p.x + (int64_t)b * p.H * p.W * 64;

    // ---- input halo tile: chunk id -> (pixel, slot); slot s of pixel q holds source chunk s ^ ((q >> 2) & 3)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pc = i * 4 + wave;               // piece (1 KB) index, 21 pieces
        if (pc < 21) {                              // wave-uniform
            const int id = pc * 64 + lane;
            const int q = id >> 2, s = id & 3;
            const int py = q / TP, px = q - py * TP;
            const int yy = y0 - 1 + py, xx = x0 - 1 + px;
            const bool ok = id < IN_CHUNKS && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const char *src = ok ? img + ((int64_t)yy * p.W + xx) * 64 + ((s ^ ((q >> 2) & 3)) * 16) : (const char *)dc_zero_page;
            __builtin_amdgcn_global_load_lds((const AS1 void *)src, (AS3 void *)(sIn + pc * 1024), 16, 0, 0);
        }
    }
    // ---- weights: COUT rows of 288 live elements = 36 chunks each
    for (int id = tid; id < COUT * 36; id += 256) {
        const int row = id / 36, c = id - row * 36;
        *(u32x4 *)(sW + row * WROW + c * 16) = *(const u32x4 *)(p.w + ((int64_t)row * p.Kp + c * 8) * 2);
    }
    f32x4 bq[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) bq[i] = p.bias ? *(const f32x4 *)(p.bias + i * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[4][NOUT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NOUT; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // output row of subtile s: wave * 4 + s (tile-local); tap (ky, kx) reads halo pixel (row + ky, r + kx)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        vec8 wf[NOUT], af[4];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) wf[i] = *(const vec8 *)(sW + (i * 16 + r) * WROW + (tap * 32 + g * 8) * 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int q = (wave * 4 + s + ky) * TP + r + kx;
            af[s] = *(const vec8 *)(sIn + q * 64 + ((g ^ ((q >> 2) & 3)) * 16));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < NOUT; ++i) acc[s][i] = T16<DT>::mfma16(wf[i], af[s], acc[s][i]);
    }
    __syncthreads();                               // every wave is done with the input tile and the weights

    // lane holds out[pixel r of row (wave*4 + s)][channels i*16 + 4g .. +3]; stage the wave's 4 rows x 16 px x COUT
    constexpr int PXB = COUT * 2;                  // bytes per pixel
    char *my = smem + wave * (4 * 16 * PXB);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            f32x4 v = acc[s][i] + bq[i];
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *(vec4 *)(my + (s * 16 + r) * PXB + (i * 16 + g * 4) * 2) = cvt4<DT>(v[0], v[1], v[2], v[3]);
        }
    // a row's 16 pixels are 16 * PXB contiguous bytes of the NHWC output: 16-byte chunks, lane-linear
    constexpr int CPR = 16 * PXB / 16;             // chunks per output row of the tile (64 or 128)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int y = y0 + wave * 4 + s;
#pragma unroll
        for (int c0 = 0; c0 < CPR; c0 += 64) {
            const int c = c0 + lane;
            const int px = c / (PXB / 16);
            if (y < p.H && x0 + px < p.W)
                *(u32x4 *)(p.out + (((int64_t)b * p.H + y) * p.W + x0) * PXB + c * 16) = *(const u32x4 *)(my + s * 16 * PXB + c * 16);
        }
    }
}

}  // namespace

// Called by hgr_conv3x3_nhwc / hgr_conv3x3_nhwc_plain (hgr_gemm.hip) for C = 32, stride 1, Cout in {32, 64}.
int hgr_conv3x3_c32_launch(const void *x, const void *w, const float *bias, void *out, int B, int H, int W, int Cout, int Kp,
                           int dtype, int relu, void *stream) {
    DcArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.bias = bias; a.out = (char *)out;
    a.B = B; a.H = H; a.W = W; a.Kp = Kp; a.relu = relu;
    a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
    const int64_t nwg = (int64_t)B * a.tiles_x * a.tiles_y;
    HGR_REQUIRE(nwg < (1ll << 31), "hgr_conv3x3_nhwc: too many tiles");
    dim3 grid((unsigned)nwg);
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 32) {
        if (dtype == HGR_BF16) hipLaunchKernelGGL((conv3x3_c32<HGR_BF16, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_c32<HGR_F16, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (dtype == HGR_BF16) hipLaunchKernelGGL((conv3x3_c32<HGR_BF16, 4>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_c32<HGR_F16, 4>), grid, dim3(256), 0, s, a);
    }
    HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
    return HGR_OK;
}
