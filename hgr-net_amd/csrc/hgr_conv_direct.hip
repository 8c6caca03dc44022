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
// where the stores of lanes outside the image go (never read): every lane of every wave issues the SAME number of store instructions per
// tile, so the end-of-tile wait can be counted ("all but my youngest N stores" = the next tile's LDS-DMA has landed) instead of
// draining this tile's stores to memory
__device__ __attribute__((aligned(16))) unsigned int dc_sink[64 * 4];

constexpr int TP = 18;                         // halo tile edge
// per input-channel count CIN (32: the stem; 64: the three 3x3 convolutions of layer1, round 3): 16-byte chunks per pixel, chunks and
// LDS-DMA pieces (1 KB) of a halo tile, bytes of one halo buffer, weight row stride in LDS (9 CIN x 2 B + 32 B pad: the four
// ds_read_b128 lane groups of a 16-row x 4-chunk fragment read hit 64 distinct banks; the first build's + 16 B was 2-way)
template <int CIN> struct DcGeom {
    static constexpr int CHP = CIN / 8;                            // 4 / 8
    static constexpr int IN_CHUNKS = TP * TP * CHP;                // 1296 / 2592
    static constexpr int PIECES = (IN_CHUNKS + 63) / 64;           // 21 / 41
    static constexpr int IN_BYTES = PIECES * 1024;
    static constexpr int WROW = 9 * CIN * 2 + 32;                  // 608 / 1184
    // workgroup shape: CIN = 32: 4 waves x 4 output rows, two workgroups per CU; CIN = 64 (halo 41 KB x 2 + weights 74 KB = 156 KB):
    // 8 waves x 2 output rows, ONE workgroup per CU (two waves per SIMD all the same)
    static constexpr int NW = CIN == 32 ? 4 : 8;
    static constexpr int RPW = 16 / NW;
};

struct DcArgs {
    const char *x; const char *w; const float *bias; char *out;
    int B, H, W, Kp, tiles_x, tiles_y, relu;
};

// POOL: the 2 x 2 average pool that follows the stem's conv3 (clip/model.py:108-109) is taken from the staged 16-bit tile
// (same values and the same fp32 average as hgr_avgpool2_nhwc on the stored tensor), so only the pooled tensor is written.
// Round 3: PERSISTENT workgroups.  The first version ran one workgroup per tile: every one of the 25 088 tiles of a batch-512 launch
// re-loaded the weights (19 / 38 KB - more than its 21 KB input tile) and waited out a memory round trip before its ~1 us of MFMAs
// (437 / 228 us per launch, 11 % of the MFMA time needed, 1.6 TB/s).  Now 2 workgroups per CU keep the weights in LDS and walk over
// tiles blockIdx.x, + gridDim.x, ...; the halo tile is double-buffered: the next tile's LDS-DMA is issued before the current tile's MFMAs
// and has the whole compute + epilogue to land.  The epilogue stages through the CURRENT halo buffer (dead after the MFMAs), two
// output rows per wave at a time (16 KB), so the weights and the buffer being filled are never touched.
template <int DT, int NOUT, bool POOL = false, int CIN = 32>  // NOUT = Cout / 16
__global__ __launch_bounds__(DcGeom<CIN>::NW * 64, CIN == 32 ? 2 : 1) void conv3x3_c32(DcArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    typedef DcGeom<CIN> G;
    constexpr int COUT = NOUT * 16, NW = G::NW, RPW = G::RPW, NT = NW * 64, CHP = G::CHP, PB = CIN * 2;
    constexpr int IN_BYTES = G::IN_BYTES, WROW = G::WROW;
    __shared__ __attribute__((aligned(1024))) char smem[2 * IN_BYTES + COUT * WROW];
    char *sW = smem + 2 * IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = p.B * p.tiles_y * p.tiles_x;
    // slot s of pixel q holds source chunk s ^ swz(q): CIN = 32 (64-byte pixels) (q >> 2) & 3, CIN = 64 (128-byte pixels) q & 7
    auto swz = [](int q) { return CIN == 32 ? (q >> 2) & 3 : q & 7; };

    // input halo tile of tile t into buffer buf
    auto issue_halo = [&](int t, int buf) {
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        const char *img = p.x + (int64_t)b * p.H * p.W * PB;
        char *sIn = smem + buf * IN_BYTES;
#pragma unroll
        for (int i = 0; i < (G::PIECES + NW - 1) / NW; ++i) {
            const int pc = i * NW + wave;              // piece (1 KB) index
            if (pc < G::PIECES) {                       // wave-uniform
                const int id = pc * 64 + lane;
                const int q = id / CHP, sl = id - q * CHP;
                const int py = q / TP, px = q - py * TP;
                const int yy = y0 - 1 + py, xx = x0 - 1 + px;
                const bool ok = id < G::IN_CHUNKS && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const char *src = ok ? img + ((int64_t)yy * p.W + xx) * PB + ((sl ^ swz(q)) * 16) : (const char *)dc_zero_page;
                __builtin_amdgcn_global_load_lds((const AS1 void *)src, (AS3 void *)(sIn + pc * 1024), 16, 0, 0);
            }
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue_halo(tile, 0);
    // ---- weights: COUT rows of 9 CIN live elements, once per workgroup
    constexpr int WCH = 9 * CIN / 8;                   // chunks per weight row (36 / 72)
    for (int id = tid; id < COUT * WCH; id += NT) {
        const int row = id / WCH, c = id - row * WCH;
        *(u32x4 *)(sW + row * WROW + c * 16) = *(const u32x4 *)(p.w + ((int64_t)row * p.Kp + c * 8) * 2);
    }
    f32x4 bq[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) bq[i] = p.bias ? *(const f32x4 *)(p.bias + i * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int PXB = COUT * 2;                  // bytes per output pixel
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        const int cur = it & 1;
        const char *sIn = smem + cur * IN_BYTES;
        const int nxt = tile + (int)gridDim.x;
        if (nxt < ntiles) issue_halo(nxt, cur ^ 1);
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;

        f32x4 acc[RPW][NOUT];
#pragma unroll
        for (int s = 0; s < RPW; ++s)
#pragma unroll
            for (int i = 0; i < NOUT; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // output row of subtile s: wave * RPW + s (tile-local); tap (ky, kx), channel block kc reads halo pixel (row + ky, r + kx)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int kc = 0; kc < CIN / 32; ++kc) {
                vec8 wf[NOUT], af[RPW];
#pragma unroll
                for (int i = 0; i < NOUT; ++i) wf[i] = *(const vec8 *)(sW + (i * 16 + r) * WROW + (tap * CIN + kc * 32 + g * 8) * 2);
#pragma unroll
                for (int s = 0; s < RPW; ++s) {
                    const int q = (wave * RPW + s + ky) * TP + r + kx;
                    af[s] = *(const vec8 *)(sIn + q * PB + (((kc * 4 + g) ^ swz(q)) * 16));
                }
#pragma unroll
                for (int s = 0; s < RPW; ++s)
#pragma unroll
                    for (int i = 0; i < NOUT; ++i) acc[s][i] = T16<DT>::mfma16(wf[i], af[s], acc[s][i]);
            }
        }
        // every wave is done with this tile's input (a raw barrier: __syncthreads() would drain the next tile's LDS-DMA here)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // lane holds out[pixel r of row (wave*RPW + s)][channels i*16 + 4g .. +3]; the wave stages 2 rows x 16 px x COUT at a time in
        // its share of the (dead) current halo buffer: NW * 2 * 16 * PXB <= IN_BYTES
        char *my = smem + cur * IN_BYTES + wave * (2 * 16 * PXB);
#pragma unroll
        for (int h = 0; h < RPW / 2; ++h) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < NOUT; ++i) {
                    f32x4 v = acc[2 * h + s2][i] + bq[i];
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *(vec4 *)(my + (s2 * 16 + r) * PXB + (i * 16 + g * 4) * 2) = cvt4<DT>(v[0], v[1], v[2], v[3]);
                }
            if (POOL) {
                // the wave's 2 rows x 16 px -> 1 pooled row x 8 px; chunk c of the pooled row = (pixel c / CPP, channels 8 (c % CPP) ..)
                constexpr int CPP = PXB / 16;
                const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
                for (int c0 = 0; c0 < 8 * CPP; c0 += 64) {
                    const int c = min(c0 + lane, 8 * CPP - 1);
                    const int px = c / CPP, cc = c - px * CPP;
                    const char *s00 = my + (2 * px) * PXB + cc * 16;
                    const vec8 a = *(const vec8 *)s00, b2 = *(const vec8 *)(s00 + PXB);
                    const vec8 c2 = *(const vec8 *)(s00 + 16 * PXB), d = *(const vec8 *)(s00 + 17 * PXB);
                    vec8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (typename T16<DT>::elem)(((float)a[e] + (float)b2[e] + (float)c2[e] + (float)d[e]) * 0.25f);
                    const int yp = (y0 >> 1) + wave * (RPW / 2) + h, xp = (x0 >> 1) + px;
                    const bool live = c0 + lane < 8 * CPP && yp < Hp && xp < Wp;
                    char *dst = live ? p.out + (((int64_t)b * Hp + yp) * Wp + xp) * PXB + cc * 16 : (char *)dc_sink + lane * 16;
                    *(vec8 *)dst = o;
                }
            } else {
                // a row's 16 pixels are 16 * PXB contiguous bytes of the NHWC output: 16-byte chunks, lane-linear
                constexpr int CPR = 16 * PXB / 16;     // chunks per output row of the tile (64 or 128)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int y = y0 + wave * RPW + 2 * h + s2;
#pragma unroll
                    for (int c0 = 0; c0 < CPR; c0 += 64) {
                        const int c = c0 + lane;
                        const int px = c / (PXB / 16);
                        const bool live = y < p.H && x0 + px < p.W;
                        char *dst = live ? p.out + (((int64_t)b * p.H + y) * p.W + x0) * PXB + c * 16 : (char *)dc_sink + lane * 16;
                        *(u32x4 *)dst = *(const u32x4 *)(my + s2 * 16 * PXB + c * 16);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the staging slice is private to the wave; its reads are done before the next half overwrites it
        }
        // the next tile's halo has landed: everything but this tile's stores (the wave's youngest NST vector-memory instructions)
        constexpr int NST = (RPW / 2) * (POOL ? 1 : 2 * (16 * PXB / 16 / 64));
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NST) : "memory");   // ... for every wave; the current buffer is free for the tile after next
    }
}


// ---- stem conv1: 3 x 3, stride 2, pad 1 over the fp32 NCHW image, 3 -> Cout channels (+ folded BN + ReLU) ------------------
// clip/model.py:106 (`relu(bn1(conv1(x)))`).  As a GEMM this is K = 27: the im2col route wrote a [pixels, 64] 16-bit matrix
// (822 MB at batch 512) and read it back for one k-step of work.  Here a workgroup owns 4 output rows of one image: the 9 x 3
// input rows they touch are converted to 16 bit into LDS once (coalesced float4 loads), every 16-pixel group builds its
// MFMA fragment with eight 2-byte LDS reads per lane (K order (ky, kx, c), the 5 columns beyond 27 meet zero weights), one
// v_mfma_f32_16x16x32 per 16 output channels, and the NHWC rows leave as whole 16-byte chunks through LDS.
struct StemArgs {
    const float *img; const char *w; const float *bias; char *out;
    int B, R, Ho, Cout, Kp, tiles_y;
};

template <int DT, int NOUT>
__global__ __launch_bounds__(256) void stem_conv1(StemArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    typedef typename T16<DT>::elem E;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int R = p.R, RS = R + 8;                 // LDS row: 4 pad + R + 4 pad elements (column x at index x + 4)
    E *tile = (E *)dsm;                            // [3][9][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int ty = blockIdx.x % p.tiles_y, b = blockIdx.x / p.tiles_y;
    const int ho0 = ty * 4, yi0 = 2 * ho0 - 1;     // first input row of the tile

    // ---- stage: 27 rows of R floats -> 16 bit
    const int r4 = R >> 2;
    for (int id = tid; id < 27 * r4; id += 256) {
        const int row = id / r4, q = id - row * r4;          // row = c * 9 + local input row
        const int c = row / 9, ly = row - c * 9, yi = yi0 + ly;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (yi >= 0 && yi < R) v = *(const f32x4 *)(p.img + (((int64_t)b * 3 + c) * R + yi) * R + q * 4);
        *(vec4 *)(tile + row * RS + 4 + q * 4) = cvt4<DT>(v[0], v[1], v[2], v[3]);
    }
    for (int row = tid; row < 27; row += 256) {              // the column left of x = 0 and right of x = R - 1
        tile[row * RS + 3] = (E)0.f;
        tile[row * RS + 4 + R] = (E)0.f;
    }
    // weights: fragment i = output channels i*16 + r, k = 8g .. 8g+7; bias quads
    vec8 wf[NOUT];
    f32x4 bq[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        const int co = i * 16 + r;
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[i][e] = (E)0.f;
        if (co < p.Cout) wf[i] = *(const vec8 *)(p.w + ((int64_t)co * p.Kp + g * 8) * 2);
        const int cq = i * 16 + g * 4;
        bq[i] = cq < p.Cout ? *(const f32x4 *)(p.bias + cq) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // per-lane offsets of the 8 taps k = 8g + e relative to (output row 0 of the tile, output column 0)
    int koff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = g * 8 + e, kk = k < 27 ? k : 0;        // k >= 27 multiplies a zero weight
        const int tap = kk / 3, c = kk - tap * 3, ky = tap / 3, kx = tap - ky * 3;
        koff[e] = (c * 9 + ky) * RS + kx + 3;
    }
    __syncthreads();

    const int nsub = (p.Ho + 15) >> 4;                       // 16-pixel groups per output row
    const int PXB = p.Cout * 2;
    char *stg = dsm + ((27 * RS * 2 + 15) & ~15) + wave * (16 * NOUT * 32);   // this wave's 16 px x (NOUT*16 channels) staging
    for (int sidx = wave; sidx < 4 * nsub; sidx += 4) {
        const int ly = sidx / nsub, sub = sidx - ly * nsub;
        const int ho = ho0 + ly, wo = sub * 16 + r;
        const int base = (2 * ly) * RS + 2 * min(wo, p.Ho - 1);
        vec8 af;
#pragma unroll
        for (int e = 0; e < 8; ++e) af[e] = tile[base + koff[e]];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            f32x4 v = T16<DT>::mfma16(wf[i], af, (f32x4){0.f, 0.f, 0.f, 0.f}) + bq[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            *(vec4 *)(stg + r * (NOUT * 32) + (i * 16 + g * 4) * 2) = cvt4<DT>(v[0], v[1], v[2], v[3]);
        }
        // 16 pixels x PXB bytes, contiguous in the NHWC output (row ho, columns sub*16 ..)
        if (ho < p.Ho) {
            char *dst = p.out + (((int64_t)b * p.Ho + ho) * p.Ho + sub * 16) * PXB;
            const int cpp = PXB >> 4;                         // 16-byte chunks per pixel (Cout % 8 == 0)
            for (int c = lane; c < 16 * cpp; c += 64) {
                const int px = c / cpp, cc = c - px * cpp;
                if (sub * 16 + px < p.Ho) *(u32x4 *)(dst + c * 16) = *(const u32x4 *)(stg + px * (NOUT * 32) + cc * 16);
            }
        }
    }
}

}  // namespace

// Called by hgr_conv3x3_nhwc / hgr_conv3x3_nhwc_plain (hgr_gemm.hip) for C = 32, stride 1, Cout in {32, 64}.
int hgr_conv3x3_c32_launch(const void *x, const void *w, const float *bias, void *out, int B, int H, int W, int Cout, int Kp,
                           int dtype, int relu, void *stream, int pool, int C) {
    DcArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.bias = bias; a.out = (char *)out;
    a.B = B; a.H = H; a.W = W; a.Kp = Kp; a.relu = relu;
    a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
    const int64_t nwg = (int64_t)B * a.tiles_x * a.tiles_y;
    HGR_REQUIRE(nwg < (1ll << 31), "hgr_conv3x3_nhwc: too many tiles");
    dim3 grid((unsigned)std::min<int64_t>(nwg, C == 64 ? 256 : 512));           // persistent: 2 (C = 64: 1) workgroups per CU walk over the tiles
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) {                                  // 64 -> 64 channels (ModifiedResNet layer1): 8 waves, one workgroup per CU
        if (dtype == HGR_BF16) hipLaunchKernelGGL((conv3x3_c32<HGR_BF16, 4, false, 64>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_c32<HGR_F16, 4, false, 64>), grid, dim3(512), 0, s, a);
        HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
        return HGR_OK;
    }
    if (pool) {
        if (Cout == 32) {
            if (dtype == HGR_BF16) hipLaunchKernelGGL((conv3x3_c32<HGR_BF16, 2, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv3x3_c32<HGR_F16, 2, true>), grid, dim3(256), 0, s, a);
        } else {
            if (dtype == HGR_BF16) hipLaunchKernelGGL((conv3x3_c32<HGR_BF16, 4, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv3x3_c32<HGR_F16, 4, true>), grid, dim3(256), 0, s, a);
        }
        HGR_CHECK_LAUNCH("hgr_conv3x3_pool2_nhwc");
        return HGR_OK;
    }
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

// out NHWC [B, Ho, Ho, Cout] 16-bit = relu(conv1(image fp32 [B, 3, R, R], stride 2, pad 1) + bias); w [Cout, Kp] 16-bit in
// (ky, kx, c) order, zero beyond column 27 (the folded layout of clip/model.py `_fold`).
extern "C" int hgr_stem_conv1(const float *image, const void *w, const float *bias, void *out, int B, int R, int Cout, int Kp,
                              int dtype, void *stream) {
    HGR_REQUIRE(image && w && bias && out && B >= 1 && R >= 4 && R % 4 == 0 && R <= 1024, "hgr_stem_conv1: bad arguments (R a multiple of 4, <= 1024)");
    HGR_REQUIRE(Cout >= 8 && Cout % 8 == 0 && Cout <= 48 && Kp >= 32 && Kp % 8 == 0, "hgr_stem_conv1: Cout=%d must be a multiple of 8 up to 48, Kp=%d >= 32", Cout, Kp);
    HGR_REQUIRE(hgr_aligned(image, 16) && hgr_aligned(w, 16) && hgr_aligned(bias, 16) && hgr_aligned(out, 16), "hgr_stem_conv1: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_stem_conv1: bad dtype %d", dtype);
    StemArgs a;
    a.img = image; a.w = (const char *)w; a.bias = bias; a.out = (char *)out;
    a.B = B; a.R = R; a.Ho = (R - 1) / 2 + 1; a.Cout = Cout; a.Kp = Kp; a.tiles_y = (a.Ho + 3) / 4;
    const int nout = (Cout + 15) / 16;
    const size_t lds = (((size_t)27 * (R + 8) * 2 + 15) & ~(size_t)15) + 4 * 16 * nout * 32;
    HGR_REQUIRE(lds <= 64 * 1024, "hgr_stem_conv1: R=%d needs %zu bytes of LDS", R, lds);
    dim3 grid((unsigned)(B * a.tiles_y));
    hipStream_t s = (hipStream_t)stream;
    if (nout <= 2) {
        if (dtype == HGR_BF16) hipLaunchKernelGGL((stem_conv1<HGR_BF16, 2>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stem_conv1<HGR_F16, 2>), grid, dim3(256), lds, s, a);
    } else {
        if (dtype == HGR_BF16) hipLaunchKernelGGL((stem_conv1<HGR_BF16, 3>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stem_conv1<HGR_F16, 3>), grid, dim3(256), lds, s, a);
    }
    HGR_CHECK_LAUNCH("hgr_stem_conv1");
    return HGR_OK;
}

// relu(conv3x3(x) + bias) followed by the 2 x 2 average pool, one kernel; x NHWC [B, H, W, 32] (H, W even), out NHWC
// [B, H/2, W/2, Cout], Cout in {32, 64}: the end of the ModifiedResNet stem (clip/model.py:108-109).
extern "C" int hgr_conv3x3_pool2_nhwc(const void *x, const void *w, const float *bias, void *out, int B, int H, int W, int C, int Cout,
                                      int Kp, int dtype, void *stream) {
    HGR_REQUIRE(x && w && bias && out && B >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "hgr_conv3x3_pool2_nhwc: bad geometry B=%d H=%d W=%d", B, H, W);
    HGR_REQUIRE(C == 32 && (Cout == 32 || Cout == 64) && Kp >= 288 && Kp % 8 == 0, "hgr_conv3x3_pool2_nhwc: built for C = 32, Cout in {32, 64} (got C=%d Cout=%d Kp=%d)", C, Cout, Kp);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(w, 16) && hgr_aligned(out, 16) && hgr_aligned(bias, 16), "hgr_conv3x3_pool2_nhwc: misaligned operand");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_conv3x3_pool2_nhwc: bad dtype %d", dtype);
    return hgr_conv3x3_c32_launch(x, w, bias, out, B, H, W, Cout, Kp, dtype, 1, stream, 1, 32);
}
