// Backward / optimizer kernels of the OM training step (reference: model/clip_tree.py:222-281 `train_batch`,
// main.py:86-94 clip_grad_norm_ + AdamW).  The heavy products reuse hgr_gemm_nt (dX = dY . W with the weight
// pre-transposed, dW = dY^T . X on transposed activations with the fp32 accumulate epilogue); everything here
// is the glue around them: streaming element-wise / row-wise kernels (HBM-bound) and small fp32 products.
#include "hgr_common.h"
#include <atomic>
#include <stdlib.h>

namespace {

unsigned grid1(int64_t total, int per = 256, int cap = 16384) { int64_t g = (total + per - 1) / per; return (unsigned)(g < cap ? (g > 0 ? g : 1) : cap); }

// ---- y[c][r] = x[r][c], 16-bit, 64 x 64 tiles through LDS ------------------------------------------------
__global__ __launch_bounds__(256) void transpose16_scalar(const unsigned short *__restrict__ x, int64_t ldx, unsigned short *__restrict__ y, int64_t ldy, int rows, int cols) {
    __shared__ unsigned short t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        t[i][tx] = (r < rows && c < cols) ? x[(int64_t)r * ldx + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) y[(int64_t)c * ldy + r] = t[tx][i];
    }
}

// 64 x 64 tile, 16-byte global loads and stores (8 elements): rows in, rows out; the transposition happens in the LDS
// reads (8 two-byte reads of one column per thread).  Needs ldx, ldy % 8 == 0 and 16-byte aligned bases; ragged edges
// are handled element-wise.
// CS = 1 (bf16) / 2 (f16): additionally partial[blockIdx.y][c] = sum of the tile's 64 rows of column c (fp32, rows in
// order) - the bias gradient rides on the transposition that the weight gradient needs anyway, instead of reading dY again.
template <int CS>
__global__ __launch_bounds__(256) void transpose16(const unsigned short *__restrict__ x, int64_t ldx, unsigned short *__restrict__ y, int64_t ldy, int rows, int cols,
                                                   float *__restrict__ partial) {
    typedef __attribute__((ext_vector_type(8))) unsigned short us8;
    __shared__ __attribute__((aligned(16))) unsigned short t[64][72];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = threadIdx.x + 256 * j;
        const int r = id >> 3, c = (id & 7) * 8;
        us8 v;
        if (r0 + r < rows && c0 + c + 7 < cols) v = *(const us8 *)(x + (int64_t)(r0 + r) * ldx + c0 + c);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (r0 + r < rows && c0 + c + e < cols) ? x[(int64_t)(r0 + r) * ldx + c0 + c + e] : (unsigned short)0;
        *(us8 *)&t[r][c] = v;
    }
    __syncthreads();
    if (CS && threadIdx.x < 64 && c0 + threadIdx.x < cols) {          // rows beyond `rows` were staged as zeros
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; i += 2) {
            const unsigned short u0 = t[i][threadIdx.x], u1 = t[i + 1][threadIdx.x];
            if (CS == 1) { a0 += __uint_as_float((unsigned)u0 << 16); a1 += __uint_as_float((unsigned)u1 << 16); }
            else { a0 += (float)__builtin_bit_cast(_Float16, u0); a1 += (float)__builtin_bit_cast(_Float16, u1); }
        }
        partial[(int64_t)blockIdx.y * cols + c0 + threadIdx.x] = a0 + a1;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = threadIdx.x + 256 * j;
        const int oc = id >> 3, o8 = (id & 7) * 8;          // output row = input column oc; 8 consecutive input rows
        if (c0 + oc >= cols) continue;
        us8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t[o8 + e][oc];
        unsigned short *dst = y + (int64_t)(c0 + oc) * ldy + r0 + o8;
        if (r0 + o8 + 7 < rows) *(us8 *)dst = v;
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) if (r0 + o8 + e < rows) dst[e] = v[e];
    }
}

// ---- column sums: partial[rb][c] = sum over a 512-row band, then out[c] (+)= sum_rb partial ----------------
template <int DT, bool F32>
__global__ __launch_bounds__(256) void colsum_partial(const void *__restrict__ x, int64_t ldx, int rows, int cols, float *__restrict__ partial) {
    typedef typename T16<DT>::elem E;
    __shared__ float s[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 512, r1 = min(rows, r0 + 512);
    float acc = 0.f;
    if (c < cols)
        for (int r = r0 + q; r < r1; r += 4)
            acc += F32 ? ((const float *)x)[(int64_t)r * ldx + c] : (float)((const E *)x)[(int64_t)r * ldx + c];
    s[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}
// 16-byte loads: 8 (16-bit) or 4 (fp32) columns per thread, 64 columns and up to 512 rows per block like colsum_partial;
// the row groups are combined through LDS in a fixed order.  Needs 16-byte aligned rows (ldx, base) and cols % 64 == 0.
template <int DT, bool F32>
__global__ __launch_bounds__(256) void colsum_partial_vec(const void *__restrict__ x, int64_t ldx, int rows, int cols, float *__restrict__ partial) {
    typedef typename T16<DT>::vec8 vec8;
    constexpr int CPT = F32 ? 4 : 8, TX = 64 / CPT, TY = 256 / TX;       // columns per thread, threads across, row groups
    __shared__ float s[TY][64];
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int c = blockIdx.x * 64 + cx * CPT;
    const int r0 = blockIdx.y * 512, r1 = min(rows, r0 + 512);
    float acc[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) acc[e] = 0.f;
    for (int r = r0 + ry; r < r1; r += TY) {
        if (F32) {
            const f32x4 v = *(const f32x4 *)((const float *)x + (int64_t)r * ldx + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        } else {
            const vec8 v = *(const vec8 *)((const typename T16<DT>::elem *)x + (int64_t)r * ldx + c);
#pragma unroll
            for (int e = 0; e < CPT; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < CPT; ++e) s[ry][cx * CPT + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int g = 0; g < TY; ++g) t += s[g][threadIdx.x];
        partial[(int64_t)blockIdx.y * cols + blockIdx.x * 64 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final(const float *__restrict__ partial, int nrb, int cols, float *__restrict__ out, int accumulate, float alpha) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.f;
    for (int rb = 0; rb < nrb; ++rb) acc += partial[(int64_t)rb * cols + c];      // fixed order: bit-reproducible
    out[c] = (accumulate ? out[c] : 0.f) + alpha * acc;
}

// Many bands (bias gradients over 10^5..10^6 pixels: thousands of bands of a few dozen columns): one thread per column
// walking every band is a serial chain of thousands of loads.  Here a block owns 64 columns and the bands
// [blockIdx.y * chunk, +chunk): its 4 waves take bands q, q + 4, ... with 4 independent accumulators each and the 16
// sub-sums are combined in a fixed order.  MID: the chunk's sum replaces the chunk's first band in place (every thread
// only ever touches its own column, and the writer has finished reading); the final pass then walks those bands.
template <bool MID>
__global__ __launch_bounds__(256) void colsum_reduce_deep(float *__restrict__ partial, int nrb, int chunk, int64_t bstride, int cols,
                                                          float *__restrict__ out, int accumulate, float alpha) {
    __shared__ float s[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b0 = blockIdx.y * chunk, b1 = min(nrb, b0 + chunk);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        const float *pc = partial + c;
        int rb = b0 + q;
        for (; rb + 12 < b1; rb += 16) {
            a0 += pc[(int64_t)rb * bstride];
            a1 += pc[(int64_t)(rb + 4) * bstride];
            a2 += pc[(int64_t)(rb + 8) * bstride];
            a3 += pc[(int64_t)(rb + 12) * bstride];
        }
        for (; rb < b1; rb += 4) a0 += pc[(int64_t)rb * bstride];
    }
    s[q][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && c < cols) {
        const float t = (s[0][lane] + s[1][lane]) + (s[2][lane] + s[3][lane]);
        if (MID) partial[(int64_t)b0 * bstride + c] = t;
        else out[c] = (accumulate ? out[c] : 0.f) + alpha * t;
    }
}

// few fp32 rows (the split-K slices of a weight gradient: 2 - 16 rows of N * K columns): one pass, 16-byte loads, rows added in order -
// the band kernel + second stage above are two launches and 4-byte loads for this shape (145 reductions per ViT-L/14 training step)
__global__ __launch_bounds__(256) void colsum_rows_f32(const float *__restrict__ x, int64_t ldx, int rows, int64_t cols4, float *__restrict__ out, int accumulate, float alpha) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cols4; c += (int64_t)gridDim.x * 256) {
        f32x4 acc = ((const f32x4 *)x)[c];
        for (int r = 1; r < rows; ++r) acc += *(const f32x4 *)(x + (int64_t)r * ldx + c * 4);
        f32x4 o = accumulate ? ((const f32x4 *)out)[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += alpha * acc[e];
        ((f32x4 *)out)[c] = o;
    }
}

// second stage of every column sum: partial [nrb][cols] -> out
void colsum_finish(float *partial, int nrb, int cols, float *out, int accumulate, float alpha, hipStream_t s) {
    if (nrb >= 2048) {
        const int chunk = 256, n2 = (nrb + chunk - 1) / chunk;
        hipLaunchKernelGGL(colsum_reduce_deep<true>, dim3((cols + 63) / 64, n2), dim3(256), 0, s, partial, nrb, chunk, (int64_t)cols, cols, out, 0, 1.0f);
        hipLaunchKernelGGL(colsum_reduce_deep<false>, dim3((cols + 63) / 64, 1), dim3(256), 0, s, partial, n2, n2, (int64_t)chunk * cols, cols, out, accumulate, alpha);
    } else if (nrb >= 32) {
        hipLaunchKernelGGL(colsum_reduce_deep<false>, dim3((cols + 63) / 64, 1), dim3(256), 0, s, partial, nrb, nrb, (int64_t)cols, cols, out, accumulate, alpha);
    } else {
        hipLaunchKernelGGL(colsum_final, dim3((cols + 255) / 256), dim3(256), 0, s, partial, nrb, cols, out, accumulate, alpha);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void cast16(const float *__restrict__ x, typename T16<DT>::elem *__restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = ((const f32x4 *)x)[i];
        ((typename T16<DT>::vec4 *)y)[i] = cvt4<DT>(v[0], v[1], v[2], v[3]);
    }
}

// fp32 [rows, cols] -> its 16-bit copy y [rows, ldy] AND the transposed copy yt [cols, ldyt] in one pass (64 x 64 tiles through LDS;
// the roundings are hgr_cast16's): the per-step refresh of a Linear's training operands (forward / weight-gradient operand and the
// [K, N] operand of dX = dY W) was three launches per weight (cast, zero fill, transpose) on ~150 weights, launch-bound at step start
template <int DT>
__global__ __launch_bounds__(256) void cast16_transpose(const float *__restrict__ x, int64_t ldx, typename T16<DT>::elem *__restrict__ y, int64_t ldy,
                                                        typename T16<DT>::elem *__restrict__ yt, int64_t ldyt, int rows, int cols, int vec) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    __shared__ __attribute__((aligned(16))) E t[64][72];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = threadIdx.x + 256 * j;
        const int r = id >> 3, c = (id & 7) * 8;
        vec8 v;
        const bool in = r0 + r < rows && c0 + c + 7 < cols;
        if (in && vec) {
            const f32x4 a = *(const f32x4 *)(x + (int64_t)(r0 + r) * ldx + c0 + c), b = *(const f32x4 *)(x + (int64_t)(r0 + r) * ldx + c0 + c + 4);
            const typename T16<DT>::vec4 lo = cvt4<DT>(a[0], a[1], a[2], a[3]), hi = cvt4<DT>(b[0], b[1], b[2], b[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            *(vec8 *)(y + (int64_t)(r0 + r) * ldy + c0 + c) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = r0 + r < rows && c0 + c + e < cols;
                const typename T16<DT>::vec4 q = cvt4<DT>(ok ? x[(int64_t)(r0 + r) * ldx + c0 + c + e] : 0.f, 0.f, 0.f, 0.f);
                v[e] = q[0];
                if (ok) y[(int64_t)(r0 + r) * ldy + c0 + c + e] = v[e];
            }
        }
        *(vec8 *)&t[r][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = threadIdx.x + 256 * j;
        const int oc = id >> 3, o8 = (id & 7) * 8;          // output row = input column oc; 8 consecutive input rows
        if (c0 + oc >= cols) continue;
        vec8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t[o8 + e][oc];
        E *dst = yt + (int64_t)(c0 + oc) * ldyt + r0 + o8;
        if (vec && r0 + o8 + 7 < rows) *(vec8 *)dst = v;
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) if (r0 + o8 + e < rows) dst[e] = v[e];
    }
}

// ---- QuickGELU forward / backward on 16-bit tensors (training keeps the pre-activation) ------------------
__device__ __forceinline__ float sigm(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)); }
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void gelu16(const typename T16<DT>::elem *__restrict__ a, const typename T16<DT>::elem *__restrict__ du,
                                              typename T16<DT>::elem *__restrict__ out, int64_t n8) {
    typedef typename T16<DT>::vec8 vec8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const vec8 av = ((const vec8 *)a)[i];
        vec8 dv, o;
        if (BWD) dv = ((const vec8 *)du)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)av[e], s = sigm(1.702f * x);
            o[e] = BWD ? mul16<DT>((float)dv[e] * s, 1.0f + 1.702f * x * (1.0f - s)) : mul16<DT>(x, s);      // the last product rounded once: the GEMM epilogues' twin
        }
        ((vec8 *)out)[i] = o;
    }
}

// ---- LayerNorm backward: dx[src] += dLN ; per-wave partial dgamma/dbeta rows (reduced by colsum) -----------
template <int DT, bool DYF32, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd(const void *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ gamma,
                                                     float *__restrict__ dx, float *__restrict__ pg, float *__restrict__ pb,
                                                     int rows, int W, int64_t row_mul, const int32_t *__restrict__ row_idx, float eps,
                                                     typename T16<DT>::elem *__restrict__ dx16, float *__restrict__ pc) {
    typedef typename T16<DT>::elem E;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nv = W >> 2;
    f32x4 ag[NV], ab[NV], ac[NV];      // ac: column sums of the ROUNDED dx16 rows this wave wrote (pc != null) - the bias gradient of the Linear that consumes dx16
#pragma unroll
    for (int i = 0; i < NV; ++i) { ag[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i]; ac[i] = ag[i]; }
    for (int row = wid; row < rows; row += nw) {
        const int64_t src = (int64_t)row * row_mul + (row_idx ? row_idx[row] : 0);
        const f32x4 *xr = (const f32x4 *)(x + src * W);
        f32x4 v[NV], g[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            v[i] = (c < nv) ? xr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        const float mean = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
        // g = dy * gamma ; xhat = (x - mean) * rstd ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < nv) {
                f32x4 d;
                if (DYF32) d = ((const f32x4 *)((const float *)dy + (int64_t)row * W))[c];
                else {
                    const typename T16<DT>::vec4 d16 = ((const typename T16<DT>::vec4 *)((const E *)dy + (int64_t)row * W))[c];
                    d = (f32x4){(float)d16[0], (float)d16[1], (float)d16[2], (float)d16[3]};
                }
                const f32x4 ga = ((const f32x4 *)gamma)[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[i][e] - mean) * rstd;
                    ag[i][e] += d[e] * xh;
                    ab[i][e] += d[e];
                    g[i][e] = d[e] * ga[e];
                    sg += g[i][e];
                    sgx += g[i][e] * xh;
                    v[i][e] = xh;
                }
            }
        }
        const float mg = wave_sum(sg) / (float)W, mgx = wave_sum(sgx) / (float)W;
        f32x4 *dxr = (f32x4 *)(dx + src * W);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                f32x4 o = dxr[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rstd * (g[i][e] - mg - v[i][e] * mgx);
                dxr[c] = o;
                // the updated gradient also as the 16-bit operand of the next data-/weight-gradient GEMMs (what hgr_cast16 of dx gives)
                if (dx16) {
                    const typename T16<DT>::vec4 h = cvt4<DT>(o[0], o[1], o[2], o[3]);
                    ((typename T16<DT>::vec4 *)(dx16 + src * W))[c] = h;
                    if (pc)
#pragma unroll
                        for (int e = 0; e < 4; ++e) ac[i][e] += (float)h[e];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            ((f32x4 *)(pg + (int64_t)wid * W))[c] = ag[i];
            ((f32x4 *)(pb + (int64_t)wid * W))[c] = ab[i];
            if (pc) ((f32x4 *)(pc + (int64_t)wid * W))[c] = ac[i];
        }
    }
}

// The per-wave partial rows of layernorm_bwd -> dgamma / dbeta / (dx16 column sums), all three in ONE launch: blockIdx.y picks the
// matrix, a block owns 64 columns and walks all nw rows (16 row groups x 16-byte loads, combined through LDS in a fixed order), then
// accumulates into its target.  Three hgr_colsum calls were six launches of ~8 us on 2 - 8 MB each, 72 times per ViT-L/14 step.
__global__ __launch_bounds__(256) void ln_bwd_reduce3(const float *__restrict__ p0, const float *__restrict__ p1, const float *__restrict__ p2, int nw, int W,
                                                      float *__restrict__ o0, float *__restrict__ o1, float *__restrict__ o2) {
    __shared__ f32x4 sm[16][16];
    const float *src = blockIdx.y == 0 ? p0 : blockIdx.y == 1 ? p1 : p2;
    float *dst = blockIdx.y == 0 ? o0 : blockIdx.y == 1 ? o1 : o2;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + tx * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < W)
        for (int r = ty; r < nw; r += 16) acc += *(const f32x4 *)(src + (int64_t)r * W + c);
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < W) {
        f32x4 t = sm[0][tx];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += sm[g][tx];
        f32x4 o = *(const f32x4 *)(dst + c);
        o += t;
        *(f32x4 *)(dst + c) = o;
    }
}

// ---- attention backward for ANY length (L <= 288 here): tiled, on the exact-fp32 MFMA ---------------------------
// v_mfma_f32_32x32x2_f32 with operands read from fp32 LDS tiles [64][65]: lane l supplies A(i = l & 31, k = l >> 5)
// and B(k = l >> 5, j = l & 31) per 2-deep step, so a transposed operand is just a different index function and the
// five products (S = QK^T, dP = dO V^T, dV = P^T dO, dQ = dS K, dK = dS^T Q) need no transposed copies.  One
// workgroup (4 waves, each a 32 x 32 quadrant of every 64 x 64 block product) per (batch, head):
//   sweep 0: row max / sum of exp over all key blocks (online merge) and D = rowsum(dO * O);
//   sweep 1: per query block, dQ += dS K over the key blocks;   sweep 2: per key block, dK, dV over the query blocks.
// P is recomputed from the row statistics (flash-attention style), nothing of size L x L is ever stored.
struct Tile { float (*t)[65]; };

// C/D layout of the 32x32 MFMA shapes: accumulator register g of lane (r32, h) holds row HGR_ACC_ROW(g, h), column r32
#define HGR_ACC_ROW(reg, h) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (h))

// ---- attention backward, 32 < L <= 320 (ViT image towers: 50 / 197 / 257 tokens; untrimmed text: 77) --------------------
// One workgroup per (batch, head), 4 waves, 64 x 64 blocks of the L x L score matrix, every product on the 16-bit matrix
// cores (v_mfma_f32_32x32x16, fp32 accumulate; a wave owns a 32 x 32 quadrant).  Flash-attention style, two sweeps:
//   sweep 0   D = rowsum(dO * O) and the softmax statistics (row max, 1 / row sum) by recomputing S = Q K^T block by block;
//   sweep 1   for every key block j:  for every query block i:  S, dP = dO V^T  ->  P, dS = P (dP - D) / 8 in registers  ->
//             dV_j += P^T dO_i,  dK_j += dS^T Q_i,  dQ_i += dS K_j.   dQ of ALL query blocks lives in registers (5 x 16 per
//             lane), so nothing is recomputed a second time and there are no atomics.
// P and dS are rounded to the 16-bit type before the second products (as the forward's P is); Q, K, V, dO are 16-bit in
// memory anyway.  An MFMA operand is 8 consecutive k for one row.  P^T / dS / dS^T are written from the accumulator layout
// (4 consecutive rows per lane -> one 8-byte store in the transposed image); the transposed views of the STAGED tiles
// (dO^T for dV, Q^T for dK, K^T for dQ) are never materialised: those B operands come out of the row-major tile through
// gfx950's transposing LDS read (ds_read_b64_tr_b16, mm16_bt below).  7 tiles + the row statistics = 67 KB of LDS, so TWO
// workgroups share a CU and one's barriers / global loads overlap the other's matrix work.  LDS rows are padded to 72
// elements: the 16-byte operand reads of 32 consecutive rows are bank-conflict free.
template <int DT>
__device__ __forceinline__ void mm16(f32x16 &acc, const typename T16<DT>::elem (*A)[72], int arow, const typename T16<DT>::elem (*B)[72], int brow, int lane) {
    typedef typename T16<DT>::vec8 vec8;
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        acc = T16<DT>::mfma32(*(const vec8 *)&A[arow + r][kk * 16 + kh * 8], *(const vec8 *)&B[brow + r][kk * 16 + kh * 8], acc);
}

typedef __attribute__((ext_vector_type(4))) short tr_s16x4;
typedef __attribute__((ext_vector_type(8))) short tr_s16x8;
__device__ __forceinline__ tr_s16x4 tr_read64(const void *p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 tr_s16x4 *)p); }

// acc[m][n] += sum_k A[arow + m][k] * Bt[k][bcol + n], k = 0..63: the B operand "8 consecutive k of column n" is read from
// the row-major tile Bt with two transposing reads.  A 16-lane group (n = bcol + 16 * ((lane >> 4) & 1) + lane % 16, k half
// kh = lane >> 5) reads a 4-row x 16-column block: lane 4 q + pp of the group supplies the address of (row k0 + q, columns
// 4 pp .. 4 pp + 3) and receives the 4 rows of column lane % 16.  EXEC must be all ones (callers branch block-uniformly).
template <int DT>
__device__ __forceinline__ void mm16_bt(f32x16 &acc, const typename T16<DT>::elem (*A)[72], int arow, const typename T16<DT>::elem (*Bt)[72], int bcol, int lane) {
    typedef typename T16<DT>::vec8 vec8;
    const int r = lane & 31, kh = lane >> 5, l16 = lane & 15;
    const int q = l16 >> 2, c = bcol + ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k0 = kk * 16 + kh * 8;
        const tr_s16x4 b0 = tr_read64(&Bt[k0 + q][c]), b1 = tr_read64(&Bt[k0 + 4 + q][c]);
        const vec8 bf = __builtin_bit_cast(vec8, (tr_s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        acc = T16<DT>::mfma32(*(const vec8 *)&A[arow + r][kk * 16 + kh * 8], bf, acc);
    }
}

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void mha_bwd_tiled(const typename T16<DT>::elem *__restrict__ qkv, const typename T16<DT>::elem *__restrict__ outp,
                                                        const typename T16<DT>::elem *__restrict__ dout, typename T16<DT>::elem *__restrict__ dqkv, int L, int H,
                                                        const float2 *__restrict__ stats, float *__restrict__ colpart) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    typedef E (*TileP)[72];
    constexpr int TB = 64 * 72 * 2;                               // one 16-bit tile
    __shared__ __attribute__((aligned(16))) char smem[7 * TB];
    const TileP sQ = (TileP)smem, sK = (TileP)(smem + TB), sV = (TileP)(smem + 2 * TB), sO = (TileP)(smem + 3 * TB),
                sDS = (TileP)(smem + 4 * TB), sDSt = (TileP)(smem + 5 * TB), sPt = (TileP)(smem + 6 * TB);
    float (*const sS)[65] = (float (*)[65])(smem + 4 * TB);      // sweep 0 only: fp32 scores over the (then unused) dS / dS^T tiles
    static_assert(64 * 65 * 4 <= 2 * TB, "score block must fit the two tiles it aliases");
    __shared__ __attribute__((aligned(16))) float rM[320];
    __shared__ __attribute__((aligned(16))) float rLinv[320];
    __shared__ __attribute__((aligned(16))) float rD[320];
    // L = 64 n + 1 (ViT-L/14: 256 patches + the class token): the last token would cost a whole fifth row AND column of 64 x 64 block
    // pairs (9 of 25) for one valid row / column.  Instead the block sweep covers the first 64 n tokens and the last token T enters as
    // vectors: its score column / row against every token (fp32 dot products of the same 16-bit operands), rounded to 16 bit like the
    // block path's P and dS, applied as rank-1 updates to the dK / dV / dQ accumulators, and its own three output rows as one reduction.
    __shared__ __attribute__((aligned(16))) float tokT[4][64];        // Q[T], K[T], V[T], dO[T]
    __shared__ __attribute__((aligned(16))) float sPc[320], sDSc[320], sPr[320], sDSr[320];   // P / dS of column T (per query) and of row T (per key)
    __shared__ float sDT;                                                                      // D[T] = dO[T] . O[T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int W = H * 64;
    const int64_t ld = 3 * (int64_t)W;
    const E *base = qkv + (int64_t)b * L * ld + h * 64;
    const E *dob = dout + (int64_t)b * L * W + h * 64;
    const E *ob = outp + (int64_t)b * L * W + h * 64;
    E *dqb = dqkv + (int64_t)b * L * ld + h * 64;
    const int nb = (L + 63) / 64;
    const bool strip = !CAUSAL && stats && L > 64 && (L & 63) == 1;   // block-uniform (the row statistics must be the forward's)
    const int nbe = strip ? nb - 1 : nb;                              // blocks the pair sweep covers
    const int T = L - 1;

    // 64 x 64 block of 16-bit rows -> registers (fetch) -> LDS row-major (put); rows past L are zero.  Every load is unconditional on a
    // clamped row: a load under its row guard is waited for on the spot, which serialised the two to ten loads of a block (round 4:
    // 257 tokens 714 -> 586 us with the strip path below, 50 tokens 164 -> 124 us per launch)
    auto fetch = [&](vec8 (&v)[2], const E *src, int64_t stride, int r0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = tid + 256 * j;
            const int r = id >> 3, c = (id & 7) * 8;
            v[j] = *(const vec8 *)(src + (int64_t)min(r0 + r, L - 1) * stride + c);       // unconditional load (row clamped), zeroed below
            if (r0 + r >= L)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = (E)0.f;
        }
    };
    auto put = [&](TileP dst, const vec8 (&v)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = tid + 256 * j;
            *(vec8 *)&dst[id >> 3][(id & 7) * 8] = v[j];
        }
    };
    // 16-bit global rows from an accumulator quadrant; returns this lane's column sum of the values AS ROUNDED (rows < L)
    auto write_rows = [&](E *dstg, int64_t stride, int r0, const f32x16 &acc) {
        float cs = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = r0 + wr * 32 + HGR_ACC_ROW(g, hh);
            if (row < L) { const E v = (E)acc[g]; dstg[(int64_t)row * stride + wc * 32 + r32] = v; cs += (float)v; }
        }
        return cs;
    };
    // masked, scaled score of (query q, key k) from a raw dot product
    auto score = [&](float raw, int q, int k) { return (k >= L || (CAUSAL && k > q)) ? -INFINITY : raw * 0.125f; };

    float tq = 0.f, tk = 0.f, tv = 0.f;        // token T's own output rows as rounded (threads 0 .. 63 = head dimensions; colpart adds them)
    if (strip) {
        // One pass over the 64-row blocks replaces sweep 0: D = rowsum(dO * O); the last token's score column and row against the block
        // (four 64-deep dot products per token, split over the four waves by dimension quarter), P / dS of that column and row rounded
        // to 16 bit as the block path rounds them; and the block's share of token T's own rows
        //   dQ[T] = sum_k dS[T][k] K[k],   dK[T] = sum_q dS[q][T] Q[q],   dV[T] = sum_q P[q][T] dO[q]     (T itself included).
        const float2 *st = stats + ((int64_t)b * H + h) * L;
        for (int i = tid; i < nb * 64; i += 256) {
            const float2 v = i < L ? st[i] : make_float2(0.f, 0.f);
            rM[i] = v.x; rLinv[i] = v.y;
        }
        if (tid < 64) {
            tokT[0][tid] = (float)base[(int64_t)T * ld + tid];          tokT[1][tid] = (float)base[(int64_t)T * ld + W + tid];
            tokT[2][tid] = (float)base[(int64_t)T * ld + 2 * W + tid];  tokT[3][tid] = (float)dob[(int64_t)T * W + tid];
            const float dt_ = wave_sum(tokT[3][tid] * (float)ob[(int64_t)T * W + tid]);      // D[T], needed for row T from the first block on
            if (tid == 0) sDT = dt_;
        }
        float *red = (float *)(smem + 5 * TB);          // [4][4][64] partial dot products, over the (unused) dS^T / P^T tiles
        const TileP sOut = sDS;                         // the forward output O of the block
        const int col = tid & 63, part = tid >> 6;
        float aq = 0.f, ak = 0.f, av = 0.f;
        // all ten 16-byte loads of a block in flight together (load() waits for each of its loads under its row guard), and the next
        // block's while this one is worked on
        vec8 fq[2], fk[2], fv[2], fo[2], fz[2];
        fetch(fq, base, ld, 0); fetch(fk, base + W, ld, 0); fetch(fv, base + 2 * W, ld, 0); fetch(fo, dob, W, 0); fetch(fz, ob, W, 0);
        for (int j = 0; j < nb; ++j) {
            __syncthreads();
            put(sQ, fq); put(sK, fk); put(sV, fv); put(sO, fo); put(sOut, fz);
            if (j + 1 < nb) {
                const int r1 = (j + 1) * 64;
                fetch(fq, base, ld, r1); fetch(fk, base + W, ld, r1); fetch(fv, base + 2 * W, ld, r1); fetch(fo, dob, W, r1); fetch(fz, ob, W, r1);
            }
            __syncthreads();
            {
                const int row = tid >> 2, p4 = tid & 3;
                float acc = 0.f;
                for (int d = p4 * 16; d < p4 * 16 + 16; ++d) acc += (float)sO[row][d] * (float)sOut[row][d];
                acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
                if (p4 == 0) rD[j * 64 + row] = acc;
            }
            {
                float sc = 0.f, dpc = 0.f, sr = 0.f, dpr = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int d0 = part * 16 + c * 8;
                    const vec8 q8 = *(const vec8 *)&sQ[col][d0], k8 = *(const vec8 *)&sK[col][d0], v8 = *(const vec8 *)&sV[col][d0], o8 = *(const vec8 *)&sO[col][d0];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sc += (float)q8[e] * tokT[1][d0 + e];     // S[tok][T]  = Q[tok] . K[T]
                        dpc += (float)o8[e] * tokT[2][d0 + e];    // dP[tok][T] = dO[tok] . V[T]
                        sr += tokT[0][d0 + e] * (float)k8[e];     // S[T][tok]  = Q[T] . K[tok]
                        dpr += tokT[3][d0 + e] * (float)v8[e];    // dP[T][tok] = dO[T] . V[tok]
                    }
                }
                red[(0 * 4 + part) * 64 + col] = sc; red[(1 * 4 + part) * 64 + col] = dpc;
                red[(2 * 4 + part) * 64 + col] = sr; red[(3 * 4 + part) * 64 + col] = dpr;
            }
            __syncthreads();
            if (tid < 64) {
                const int tok = j * 64 + tid;
                float pc = 0.f, dsc = 0.f, pr = 0.f, dsr = 0.f;
                if (tok < L) {
                    const float sc = (red[(0 * 4 + 0) * 64 + tid] + red[(0 * 4 + 1) * 64 + tid]) + (red[(0 * 4 + 2) * 64 + tid] + red[(0 * 4 + 3) * 64 + tid]);
                    const float dpc = (red[(1 * 4 + 0) * 64 + tid] + red[(1 * 4 + 1) * 64 + tid]) + (red[(1 * 4 + 2) * 64 + tid] + red[(1 * 4 + 3) * 64 + tid]);
                    const float sr = (red[(2 * 4 + 0) * 64 + tid] + red[(2 * 4 + 1) * 64 + tid]) + (red[(2 * 4 + 2) * 64 + tid] + red[(2 * 4 + 3) * 64 + tid]);
                    const float dpr = (red[(3 * 4 + 0) * 64 + tid] + red[(3 * 4 + 1) * 64 + tid]) + (red[(3 * 4 + 2) * 64 + tid] + red[(3 * 4 + 3) * 64 + tid]);
                    pc = __expf(sc * 0.125f - rM[tok]) * rLinv[tok];
                    pr = __expf(sr * 0.125f - rM[T]) * rLinv[T];
                    dsc = pc * (dpc - rD[tok]) * 0.125f;
                    dsr = pr * (dpr - sDT) * 0.125f;
                }
                sPc[tok] = (float)(E)pc;  sDSc[tok] = (float)(E)dsc;
                sPr[tok] = (float)(E)pr;  sDSr[tok] = (float)(E)dsr;
            }
            __syncthreads();
            {   // the block's share of token T's own rows
                float q_ = 0.f, k_ = 0.f, v_ = 0.f;
#pragma unroll 4
                for (int r = part * 16; r < part * 16 + 16; ++r) {
                    q_ += sDSr[j * 64 + r] * (float)sK[r][col];
                    k_ += sDSc[j * 64 + r] * (float)sQ[r][col];
                    v_ += sPc[j * 64 + r] * (float)sO[r][col];
                }
                aq += q_; ak += k_; av += v_;
            }
        }
        __syncthreads();
        red[(0 * 4 + part) * 64 + col] = aq; red[(1 * 4 + part) * 64 + col] = ak; red[(2 * 4 + part) * 64 + col] = av;
        __syncthreads();
        if (tid < 64) {
            const E vq = (E)((red[(0 * 4 + 0) * 64 + tid] + red[(0 * 4 + 1) * 64 + tid]) + (red[(0 * 4 + 2) * 64 + tid] + red[(0 * 4 + 3) * 64 + tid]));
            const E vk = (E)((red[(1 * 4 + 0) * 64 + tid] + red[(1 * 4 + 1) * 64 + tid]) + (red[(1 * 4 + 2) * 64 + tid] + red[(1 * 4 + 3) * 64 + tid]));
            const E vv = (E)((red[(2 * 4 + 0) * 64 + tid] + red[(2 * 4 + 1) * 64 + tid]) + (red[(2 * 4 + 2) * 64 + tid] + red[(2 * 4 + 3) * 64 + tid]));
            E *o = dqb + (int64_t)T * ld + tid;
            o[0] = vq; o[W] = vk; o[2 * W] = vv;
            tq = (float)vq; tk = (float)vk; tv = (float)vv;
        }
    }
    // ---- sweep 0: row statistics ---------------------------------------------------------------------------------------------------------
    for (int qi = 0; qi < (strip ? 0 : nb); ++qi) {
        {
            vec8 fq[2], fo[2], fz[2];
            fetch(fq, base, ld, qi * 64); fetch(fo, dob, W, qi * 64); fetch(fz, ob, W, qi * 64);
            __syncthreads();
            put(sQ, fq); put(sO, fo);
            put(sV, fz);                                       // forward output O of this query block (in sV for the moment)
        }
        __syncthreads();
        {   // D = rowsum(dO * O): 4 threads per row
            const int row = tid >> 2, part = tid & 3;
            float acc = 0.f;
            for (int d = part * 16; d < part * 16 + 16; ++d) acc += (float)sO[row][d] * (float)sV[row][d];
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
            if (part == 0) { rD[qi * 64 + row] = acc; rM[qi * 64 + row] = -INFINITY; rLinv[qi * 64 + row] = 0.f; }
        }
        if (stats) continue;                                    // block-uniform: the forward kept the row statistics (hgr_mha_stats)
        vec8 pk[2];
        fetch(pk, base + W, ld, 0);
        for (int kj = 0; kj < nb; ++kj) {
            if (CAUSAL && kj > qi) break;
            __syncthreads();
            put(sK, pk);
            fetch(pk, base + W, ld, min(kj + 1, nb - 1) * 64);      // the next key block (a harmless re-read after the last one)
            __syncthreads();
            f32x16 acc = {0.f};
            mm16<DT>(acc, sQ, wr * 32, sK, wc * 32, lane);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int rr = wr * 32 + HGR_ACC_ROW(g, hh), cc = wc * 32 + r32;
                sS[rr][cc] = score(acc[g], qi * 64 + rr, kj * 64 + cc);
            }
            __syncthreads();
            const int row = tid >> 2, part = tid & 3;
            float mb = -INFINITY;
            for (int c = part; c < 64; c += 4) mb = fmaxf(mb, sS[row][c]);
            mb = fmaxf(mb, __shfl_xor(mb, 1)); mb = fmaxf(mb, __shfl_xor(mb, 2));
            const float mo = rM[qi * 64 + row], mn = fmaxf(mo, mb);
            float sum = 0.f;
            if (mn > -INFINITY) for (int c = part; c < 64; c += 4) sum += __expf(sS[row][c] - mn);
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
            if (part == 0) {
                rLinv[qi * 64 + row] = rLinv[qi * 64 + row] * (mo > -INFINITY ? __expf(mo - mn) : 0.f) + sum;   // running sum for now
                rM[qi * 64 + row] = mn;
            }
        }
    }
    __syncthreads();
    if (stats) {
        const float2 *st = stats + ((int64_t)b * H + h) * L;
        for (int i = tid; i < nb * 64; i += 256) {
            const float2 v = i < L ? st[i] : make_float2(0.f, 0.f);
            rM[i] = v.x; rLinv[i] = v.y;
        }
    } else {
        for (int i = tid; i < nb * 64; i += 256) rLinv[i] = (i < L && rLinv[i] > 0.f) ? 1.0f / rLinv[i] : 0.f;
    }
    __syncthreads();

    // ---- sweep 1: dK, dV per key block, dQ of every query block accumulated in registers -------------------------------------------------
    f32x16 dq[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) dq[i] = (f32x16){0.f};
    float csq = 0.f, csk = 0.f, csv = 0.f;                       // column sums of the rounded dq / dk / dv rows (colpart)
    vec8 pq[2], po[2];                                           // Q / dO rows of the next (key block, query block) pair, in flight
    fetch(pq, base, ld, 0);
    fetch(po, dob, W, 0);
    for (int kj = 0; kj < nbe; ++kj) {
        {
            vec8 fk[2], fv[2];
            fetch(fk, base + W, ld, kj * 64);
            fetch(fv, base + 2 * W, ld, kj * 64);
            __syncthreads();
            put(sK, fk); put(sV, fv);
        }
        f32x16 dk = {0.f}, dv = {0.f};
#pragma unroll
        for (int qi = 0; qi < 5; ++qi) {
            if (qi >= nbe || (CAUSAL && qi < kj)) continue;      // block-uniform
            __syncthreads();
            put(sQ, pq);
            put(sO, po);
            {
                const int nxt = qi + 1 < nbe ? qi + 1 : (CAUSAL ? min(kj + 1, nbe - 1) : 0);   // the pair after this one starts there
                fetch(pq, base, ld, nxt * 64);
                fetch(po, dob, W, nxt * 64);
            }
            __syncthreads();
            f32x16 sacc = {0.f}, dp = {0.f};
            mm16<DT>(sacc, sQ, wr * 32, sK, wc * 32, lane);       // S = Q K^T
            mm16<DT>(dp, sO, wr * 32, sV, wc * 32, lane);         // dP = dO V^T
            const int cc = wc * 32 + r32;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                      // 4 consecutive rows of this lane's column per step
                vec4 p4, d4;
                const int rr0 = wr * 32 + HGR_ACC_ROW(g4 * 4, hh);     // a multiple of 4: the 4 rows' statistics are one 16-byte LDS read each
                const f32x4 m4 = *(const f32x4 *)&rM[qi * 64 + rr0], l4 = *(const f32x4 *)&rLinv[qi * 64 + rr0], dd4 = *(const f32x4 *)&rD[qi * 64 + rr0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int g = g4 * 4 + e;
                    const int rr = rr0 + e, q = qi * 64 + rr;
                    const float sc = score(sacc[g], q, kj * 64 + cc);
                    const float pv = (sc > -INFINITY) ? __expf(sc - m4[e]) * l4[e] : 0.f;
                    const float dsv = pv * (dp[g] - dd4[e]) * 0.125f;
                    p4[e] = (E)pv; d4[e] = (E)dsv;
                    sDS[rr][cc] = (E)dsv;
                }
                *(vec4 *)&sPt[cc][rr0] = p4;                      // transposed images: [key][query]
                *(vec4 *)&sDSt[cc][rr0] = d4;
            }
            __syncthreads();
            mm16_bt<DT>(dv, sPt, wr * 32, sO, wc * 32, lane);     // dV += P^T dO      (rows = keys, columns = d; k = queries)
            mm16_bt<DT>(dk, sDSt, wr * 32, sQ, wc * 32, lane);    // dK += dS^T Q
            mm16_bt<DT>(dq[qi], sDS, wr * 32, sK, wc * 32, lane); // dQ_i += dS K      (k = keys; reading dS through the transposing read
                                                                  // too, out of its [key][query] image, measured 7 % slower than this row-major copy)
        }
        if (strip) {         // query T against this key block: dK[k] += dS[T][k] Q[T], dV[k] += P[T][k] dO[T]
            const float qT = tokT[0][wc * 32 + r32], oT = tokT[3][wc * 32 + r32];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int k = kj * 64 + wr * 32 + HGR_ACC_ROW(g, hh);
                dk[g] += sDSr[k] * qT;
                dv[g] += sPr[k] * oT;
            }
        }
        csk += write_rows(dqb + W, ld, kj * 64, dk);
        csv += write_rows(dqb + 2 * W, ld, kj * 64, dv);
    }
#pragma unroll
    for (int qi = 0; qi < 5; ++qi)
        if (qi < nbe) {
            if (strip) {     // key T against this query block: dQ[q] += dS[q][T] K[T]
                const float kT = tokT[1][wc * 32 + r32];
#pragma unroll
                for (int g = 0; g < 16; ++g) dq[qi][g] += sDSc[qi * 64 + wr * 32 + HGR_ACC_ROW(g, hh)] * kT;
            }
            csq += write_rows(dqb, ld, qi * 64, dq[qi]);
        }
    if (colpart) {
        // column sums of this (batch, head)'s dq / dk / dv rows = its share of the in_proj bias gradient: lane halves (hh) by shuffle,
        // the two wave rows through LDS (the tiles are dead), then colpart[b][which * W + h * 64 + c] - every entry written once
        float *sc = (float *)smem;                  // [3][2][64]
        csq += __shfl_xor(csq, 32); csk += __shfl_xor(csk, 32); csv += __shfl_xor(csv, 32);
        __syncthreads();
        if (hh == 0) {
            sc[(0 * 2 + wr) * 64 + wc * 32 + r32] = csq;
            sc[(1 * 2 + wr) * 64 + wc * 32 + r32] = csk;
            sc[(2 * 2 + wr) * 64 + wc * 32 + r32] = csv;
        }
        __syncthreads();
        if (tid < 64) {                         // threads 0 .. 63 hold token T's rounded values of the strip path (zero otherwise)
            sc[384 + tid] = tq; sc[384 + 64 + tid] = tk; sc[384 + 128 + tid] = tv;
        }
        __syncthreads();
        if (tid < 192) {
            const int which = tid >> 6, c = tid & 63;
            colpart[(int64_t)b * 3 * W + which * W + h * 64 + c] = (sc[(which * 2 + 0) * 64 + c] + sc[(which * 2 + 1) * 64 + c]) + sc[384 + which * 64 + c];
        }
    }
}

// Same arithmetic for short sequences (text prompts are ~8-20 tokens after EOT trimming): LP = 16 or 32 padded
// positions, 20-45 KB of LDS so several (batch, head) workgroups share a CU, plain per-output loops.
// ---- attention backward, L <= 32 (EOT-trimmed prompts of the text tower: thousands of tiny problems per inner step) ----------
// One WAVE per (batch, head): the whole L x L problem is a single 32 x 32 block, so every product is a chain of 2 or 4
// v_mfma_f32_32x32x16 with the operands staged once in LDS (rows past L zero-filled): S = Q K^T and dP = dO V^T (k = 64),
// the row softmax and D = rowsum(P * dP) by shuffles inside the 32-lane halves of the accumulator layout (a lane owns one
// key column and 16 query rows), then dV = P^T dO, dK = dS^T Q, dQ = dS K (k = 32) on 16-bit P / dS images written from
// the accumulators.  No barriers (one wave), 26 KB of LDS (the transposed operands come from ds_read_b64_tr_b16), 6 waves per CU.
template <int DT>
__device__ __forceinline__ void mm16k(f32x16 &acc, const typename T16<DT>::elem *A, int lda, const typename T16<DT>::elem *B, int ldb, int ksteps, int lane) {
    typedef typename T16<DT>::vec8 vec8;
    const int r = lane & 31, kh = lane >> 5;
    for (int kk = 0; kk < ksteps; ++kk)
        acc = T16<DT>::mfma32(*(const vec8 *)(A + r * lda + kk * 16 + kh * 8), *(const vec8 *)(B + r * ldb + kk * 16 + kh * 8), acc);
}

// mm16k with the B operand read out of a row-major [k][n] tile by the transposing LDS read (see mm16_bt): acc[m][n] += sum_k A[m][k] Bt[k][bcol + n]
template <int DT>
__device__ __forceinline__ void mm16k_bt(f32x16 &acc, const typename T16<DT>::elem *A, int lda, const typename T16<DT>::elem *Bt, int ldb, int bcol, int ksteps, int lane) {
    typedef typename T16<DT>::vec8 vec8;
    const int r = lane & 31, kh = lane >> 5, l16 = lane & 15;
    const int q = l16 >> 2, c = bcol + ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
    for (int kk = 0; kk < ksteps; ++kk) {
        const int k0 = kk * 16 + kh * 8;
        const tr_s16x4 b0 = tr_read64(Bt + (k0 + q) * ldb + c), b1 = tr_read64(Bt + (k0 + 4 + q) * ldb + c);
        const vec8 bf = __builtin_bit_cast(vec8, (tr_s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        acc = T16<DT>::mfma32(*(const vec8 *)(A + r * lda + kk * 16 + kh * 8), bf, acc);
    }
}

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(64) void mha_bwd_wave(const typename T16<DT>::elem *__restrict__ qkv, const typename T16<DT>::elem *__restrict__ dout,
                                                   typename T16<DT>::elem *__restrict__ dqkv, int L, int H, float *__restrict__ colpart) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    constexpr int LR = 72, LT = 40;                 // row strides (elements): [32][72] row images, [64][40] / [32][40] transposed images
    // Q^T, K^T, dO^T are never staged: the k = 32 products read them out of the row-major tiles with ds_read_b64_tr_b16 (mm16k_bt):
    // 26 instead of 41 KB of LDS (6 instead of 3 workgroups per CU) and no 2-byte transposing stores
    __shared__ __attribute__((aligned(16))) E sQ[32 * LR], sK[32 * LR], sV[32 * LR], sO[32 * LR], sDS[32 * LT], sDSt[32 * LT], sPt[32 * LT];
    const int lane = threadIdx.x, r32 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int W = H * 64;
    const int64_t ld = 3 * (int64_t)W;
    const E *base = qkv + (int64_t)b * L * ld + h * 64;
    const E *dob = dout + (int64_t)b * L * W + h * 64;
    // stage Q, K, V, dO (32 x 64 each: 256 chunks of 8 elements, 4 per lane) and the transposed images of Q, K, dO
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int id = lane + 64 * j, r = id >> 3, c = (id & 7) * 8;
        // unconditional loads of a clamped row (a load under a branch is waited for on the spot: four round trips instead of one), zeroed after
        const int rc = min(r, L - 1);
        vec8 q = *(const vec8 *)(base + (int64_t)rc * ld + c), k = *(const vec8 *)(base + (int64_t)rc * ld + W + c);
        vec8 v = *(const vec8 *)(base + (int64_t)rc * ld + 2 * W + c), o = *(const vec8 *)(dob + (int64_t)rc * W + c);
        if (r >= L) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { q[e] = (E)0.f; k[e] = (E)0.f; v[e] = (E)0.f; o[e] = (E)0.f; }
        }
        *(vec8 *)&sQ[r * LR + c] = q; *(vec8 *)&sK[r * LR + c] = k; *(vec8 *)&sV[r * LR + c] = v; *(vec8 *)&sO[r * LR + c] = o;
    }
    __syncthreads();
    f32x16 sacc = {0.f}, dp = {0.f};
    mm16k<DT>(sacc, sQ, LR, sK, LR, 4, lane);       // S[q][j]
    mm16k<DT>(dp, sO, LR, sV, LR, 4, lane);         // dP[q][j]
    // this lane: key column j = r32, query rows HGR_ACC_ROW(g, hh); a row's 32 columns live in the 32 lanes of one half
    float pv[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int q = HGR_ACC_ROW(g, hh);
        const float sc = (r32 >= L || (CAUSAL && r32 > q)) ? -INFINITY : sacc[g] * 0.125f;
        float mx = sc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = (sc > -INFINITY) ? __expf(sc - mx) : 0.f;
        float sum = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float p = (q < L && sum > 0.f) ? e / sum : 0.f;
        float d = p * dp[g];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) d += __shfl_xor(d, o);
        pv[g] = p;
        dp[g] = p * (dp[g] - d) * 0.125f;            // dS
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        vec4 p4, d4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = g4 * 4 + e;
            p4[e] = (E)pv[g]; d4[e] = (E)dp[g];
            sDS[HGR_ACC_ROW(g, hh) * LT + r32] = (E)dp[g];
        }
        const int q0 = HGR_ACC_ROW(g4 * 4, hh);
        *(vec4 *)&sPt[r32 * LT + q0] = p4;
        *(vec4 *)&sDSt[r32 * LT + q0] = d4;
    }
    __syncthreads();
    E *dqb = dqkv + (int64_t)b * L * ld + h * 64;
#pragma unroll
    for (int half = 0; half < 2; ++half) {          // the 64 head dimensions as two blocks of 32 columns
        f32x16 dq = {0.f}, dk = {0.f}, dv = {0.f};
        mm16k_bt<DT>(dq, sDS, LT, sK, LR, half * 32, 2, lane);          // dQ[q][d] = sum_j dS[q][j] K[j][d]
        mm16k_bt<DT>(dk, sDSt, LT, sQ, LR, half * 32, 2, lane);         // dK[j][d] = sum_q dS[q][j] Q[q][d]
        mm16k_bt<DT>(dv, sPt, LT, sO, LR, half * 32, 2, lane);          // dV[j][d] = sum_q P[q][j] dO[q][d]
        float cq = 0.f, ck = 0.f, cv = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = HGR_ACC_ROW(g, hh);
            if (row < L) {
                E *o = dqb + (int64_t)row * ld + half * 32 + r32;
                const E vq = (E)dq[g], vk = (E)dk[g], vv = (E)dv[g];
                o[0] = vq; o[W] = vk; o[2 * W] = vv;
                cq += (float)vq; ck += (float)vk; cv += (float)vv;
            }
        }
        if (colpart) {      // column sums of the rounded rows (the in_proj bias gradient's share of this (batch, head)), see mha_bwd_tiled
            cq += __shfl_xor(cq, 32); ck += __shfl_xor(ck, 32); cv += __shfl_xor(cv, 32);
            if (hh == 0) {
                float *cp = colpart + (int64_t)b * 3 * W + h * 64 + half * 32 + r32;
                cp[0] = cq; cp[W] = ck; cp[2 * W] = cv;
            }
        }
    }
}

// ---- softmax cross-entropy over rows of fp32 logits: loss_row, dlogits = (softmax - onehot) * gscale ----------
__global__ __launch_bounds__(256) void ce_rows(const float *__restrict__ logits, int64_t ld, const int32_t *__restrict__ labels, int rows, int n,
                                               float gscale, float *__restrict__ loss_rows, float *__restrict__ dlogits, int64_t ldd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *lr = logits + (int64_t)row * ld;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, lr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += __expf(lr[j] - mx);
    sum = wave_sum(sum);
    const int lab = labels[row];
    const float lse = mx + __logf(sum);
    if (lane == 0) loss_rows[row] = lse - lr[lab];
    if (dlogits)
        for (int j = lane; j < n; j += 64) dlogits[(int64_t)row * ldd + j] = (__expf(lr[j] - lse) - (j == lab ? 1.f : 0.f)) * gscale;
}

// ---- y = x/|x| backward: dx (+)= (dy - y (y . dy)) / |x| -------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_bwd(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int rows, int D, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (int64_t)row * D, *dr = dy + (int64_t)row * D;
    float ss = 0.f, sd = 0.f;
    for (int j = lane; j < D; j += 64) { ss += xr[j] * xr[j]; sd += xr[j] * dr[j]; }
    ss = wave_sum(ss); sd = wave_sum(sd);
    const float inv = 1.0f / sqrtf(ss);
    const float k = sd * inv * inv * inv;                 // (y . dy) / |x| * (1/|x|) with y = x / |x|
    float *o = dx + (int64_t)row * D;
    for (int j = lane; j < D; j += 64) o[j] = (accumulate ? o[j] : 0.f) + dr[j] * inv - xr[j] * k;
}

// fp32 product with generic strides on the fp32 matrix cores (v_mfma_f32_32x32x2f32: exact fp32 multiply-add).
// 64 x 64 output tile per workgroup, 4 waves of 32 x 32, K in steps of 32 through LDS ([k][m] and [k][n] images, so an
// MFMA operand read is one conflict-free ds_read_b32 per lane), the next K-step's global loads in flight while the
// current one computes.  An operand tile is fetched with 16-byte loads along whichever of its two dimensions has
// stride 1 (row-major and transposed views are both common here); anything else, and every edge, falls back to
// guarded scalar loads.  Used for the loss head / projections of the training step and DGP's dense layers.
constexpr int MM_BK = 32, MM_LD = 68;

// 8 floats of a [64 (outer) x 32 (k)] operand tile for thread `t` -> regs; `so` = outer stride, `sk` = k stride
__device__ __forceinline__ void mm_fetch(const float *__restrict__ P, int64_t so, int64_t sk, int o0, int O, int k0, int K, bool vec, int t, float (&v)[8]) {
    if (sk == 1) {                       // k contiguous: rows t/8 and t/8 + 32, 4 consecutive k
        const int kk = k0 + (t & 7) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = o0 + (t >> 3) + 32 * j;
            const float *q = P + (int64_t)o * so + kk;
            if (vec && o < O && kk + 3 < K) { const f32x4 x = *(const f32x4 *)q; v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3]; }
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = (o < O && kk + e < K) ? q[e] : 0.f;
        }
    } else {                             // outer contiguous (or generic): k rows t/16 and t/16 + 16, 4 consecutive outer
        const int oo = o0 + (t & 15) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + (t >> 4) + 16 * j;
            const float *q = P + (int64_t)k * sk + (int64_t)oo * so;
            if (vec && so == 1 && k < K && oo + 3 < O) { const f32x4 x = *(const f32x4 *)q; v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3]; }
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = (k < K && oo + e < O) ? q[(int64_t)e * so] : 0.f;
        }
    }
}

__device__ __forceinline__ void mm_stage(float (*S)[MM_LD], int64_t sk, int t, const float (&v)[8]) {
    if (sk == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) S[(t & 7) * 4 + e][(t >> 3) + 32 * j] = v[4 * j + e];
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) *(f32x4 *)&S[(t >> 4) + 16 * j][(t & 15) * 4] = (f32x4){v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    }
}

__global__ __launch_bounds__(256) void matmul_f32(const float *__restrict__ A, int64_t sam, int64_t sak, const float *__restrict__ B, int64_t sbk, int64_t sbn,
                                                  float *__restrict__ C, int64_t ldc, int M, int N, int K, float alpha, int accumulate, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[2][MM_BK][MM_LD], Bs[2][MM_BK][MM_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int r = lane & 31, h = lane >> 5;
    float ra[8], rb[8];
    mm_fetch(A, sam, sak, m0, M, 0, K, vecA != 0, t, ra);
    mm_fetch(B, sbn, sbk, n0, N, 0, K, vecB != 0, t, rb);
    mm_stage(As[0], sak, t, ra);
    mm_stage(Bs[0], sbk, t, rb);
    __syncthreads();
    f32x16 acc = {0.f};
    const int nk = (K + MM_BK - 1) / MM_BK;
    for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        if (s + 1 < nk) {
            mm_fetch(A, sam, sak, m0, M, (s + 1) * MM_BK, K, vecA != 0, t, ra);
            mm_fetch(B, sbn, sbk, n0, N, (s + 1) * MM_BK, K, vecB != 0, t, rb);
        }
#pragma unroll
        for (int kk = 0; kk < MM_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][kk + h][wm + r], Bs[cur][kk + h][wn + r], acc, 0, 0, 0);
        if (s + 1 < nk) {
            mm_stage(As[cur ^ 1], sak, t, ra);
            mm_stage(Bs[cur ^ 1], sbk, t, rb);
        }
        __syncthreads();
    }
    const int n = n0 + wn + r;
    if (n < N)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wm + (i & 3) + 8 * (i >> 2) + 4 * h;           // C/D layout of the 32x32 shapes
            if (m < M) {
                float *c = C + (int64_t)m * ldc + n;
                *c = (accumulate ? *c : 0.f) + alpha * acc[i];
            }
        }
}

// ---- token-embedding gradient: table[token[i,t]] += dx[i*L+t] (fp32 atomics; repeated tokens contend) -------------
__global__ __launch_bounds__(256) void embed_scatter_add(const int64_t *__restrict__ tokens, int64_t ldt, const float *__restrict__ dx, float *__restrict__ dtable,
                                                         int n, int L, int W, int vocab) {
    const int64_t total = (int64_t)n * L * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % W);
        const int64_t row = i / W;
        const int t = (int)(row % L);
        int64_t id = tokens[(row / L) * ldt + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        atomicAdd(dtable + id * W + c, dx[i]);
    }
}

// ---- sum of squares (global grad norm) and fused AdamW with the clip factor read from device memory -----------------
// Deterministic: every block leaves its partial in g_sumsq_part, the block that draws the last ticket adds the partials in index
// order.  (The first version atomicAdd'ed the block sums into *out: the order, hence the last bits of the global gradient norm,
// hence of the clip factor and of every weight, differed from run to run and from RANK to RANK - data-parallel replicas drifted
// apart by ulps per step; found by bench.py's dp_check.)  One agent-scope release per block and one acquire in the last block
// (cdna_hip_programming.md, Guideline 16).  The scratch is process-wide but SLOTTED: the host hands consecutive launches consecutive
// slots of a ring of SUMSQ_SLOTS (partials + ticket each), so launches in flight on different streams do not share a ticket unless more
// than SUMSQ_SLOTS of them overlap (include/hgr.h states that bound).
constexpr int SUMSQ_SLOTS = 16;
__device__ float g_sumsq_part[SUMSQ_SLOTS][1024];
__device__ unsigned g_sumsq_ticket[SUMSQ_SLOTS];
__global__ __launch_bounds__(256) void sumsq(const float *__restrict__ x, int64_t n, float *__restrict__ out, int slot) {
    float *const part = g_sumsq_part[slot];
    unsigned *const ticket = &g_sumsq_ticket[slot];
    __shared__ float s[4];
    __shared__ unsigned s_last;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // fixed order: thread t sums partials t, t + 256, ... ; then the wave / block tree above
    float tot = 0.f;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) tot += __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot = wave_sum(tot);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        *out += (s[0] + s[1]) + (s[2] + s[3]);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// torch.optim.AdamW semantics: p *= 1 - lr*wd ; m, v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
// `sumsq_total` (may be NULL) holds the squared global grad norm: grads are scaled by min(1, max_norm/(norm+1e-6))
// like torch.nn.utils.clip_grad_norm_ (main.py:88).
__global__ __launch_bounds__(256) void adamw(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n,
                                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, const float *__restrict__ sumsq_total, float max_norm, float gscale) {
    float clip = gscale;                                      // gscale: e.g. 1/world after a sum all-reduce
    if (sumsq_total) { const float norm = sqrtf(*sumsq_total) * gscale; clip = gscale * fminf(1.f, max_norm / (norm + 1e-6f)); }
    const float rs2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi, vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] * (1.f - lr * wd) - (lr / bc1) * mi / (sqrtf(vi) * rs2 + eps);
    }
}

// gather / scatter-add of rows (EOT rows, class-token rows): out[r] = x[r*mul + idx[r]] ; x[...] += in[r]
__global__ __launch_bounds__(256) void rows_axpy(float *__restrict__ dst, int64_t dst_mul, const int32_t *__restrict__ dst_idx, const float *__restrict__ src, int rows, int W, float alpha) {
    const int64_t total = (int64_t)rows * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / W), c = (int)(i % W);
        const int64_t d = (int64_t)r * dst_mul + (dst_idx ? dst_idx[r] : 0);
        dst[d * W + c] += alpha * src[i];
    }
}

// CoOp learnable context (reference model/CoOp.py:98-113): rows 1 .. n_ctx of every prompt are the shared context
// vectors instead of token embeddings.  x[i*L + 1 + c] = ctx[c] + positional[1 + c].
__global__ __launch_bounds__(256) void ctx_splice(float *__restrict__ x, const float *__restrict__ ctx, const float *__restrict__ pos, int n, int L, int W, int n_ctx) {
    const int64_t total = (int64_t)n * n_ctx * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % W);
        const int k = (int)((i / W) % n_ctx);
        const int64_t pr = i / ((int64_t)W * n_ctx);
        x[(pr * L + 1 + k) * W + c] = ctx[(int64_t)k * W + c] + pos[(int64_t)(1 + k) * W + c];
    }
}
// backward: dctx[k] += sum_i dx[i*L + 1 + k] in a fixed order (bit-reproducible), and those rows of dx are cleared so the
// token-embedding scatter that follows gives the placeholder tokens no gradient.  One block per (k, 64-column chunk).
__global__ __launch_bounds__(256) void ctx_splice_bwd(float *__restrict__ dx, float *__restrict__ dctx, int n, int L, int W, int n_ctx) {
    __shared__ float s[4][64];
    const int k = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < W)
        for (int i = q; i < n; i += 4) {
            float *p = dx + ((int64_t)i * L + 1 + k) * W + c;
            acc += *p;
            *p = 0.f;
        }
    s[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && c < W) dctx[(int64_t)k * W + c] += s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

// dst[r] = src[idx[r]] for W-wide fp32 rows
__global__ __launch_bounds__(256) void rows_gather(float *__restrict__ dst, const float *__restrict__ src, const int32_t *__restrict__ idx, int rows, int W) {
    const int64_t total = (int64_t)rows * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        dst[i] = src[(int64_t)idx[i / W] * W + i % W];
}

}  // namespace

#define DT_OK(name) HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, name ": bad dtype %d", dtype)

extern "C" int hgr_transpose16(const void *x, int64_t ldx, void *y, int64_t ldy, int rows, int cols, void *stream) {
    HGR_REQUIRE(x && y && rows >= 1 && cols >= 1 && ldx >= cols && ldy >= rows, "hgr_transpose16: bad arguments");
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (ldx % 8 == 0 && ldy % 8 == 0 && hgr_aligned(x, 16) && hgr_aligned(y, 16))
        hipLaunchKernelGGL(transpose16<0>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, ldx, (unsigned short *)y, ldy, rows, cols, (float *)nullptr);
    else
        hipLaunchKernelGGL(transpose16_scalar, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, ldx, (unsigned short *)y, ldy, rows, cols);
    HGR_CHECK_LAUNCH("hgr_transpose16");
    return HGR_OK;
}

extern "C" int hgr_colsum(const void *x, int64_t ldx, int rows, int cols, int x_f32, int dtype, float *out, int accumulate, float alpha, float *scratch, void *stream) {
    HGR_REQUIRE(x && out && scratch && rows >= 1 && cols >= 1 && ldx >= cols, "hgr_colsum: bad arguments");
    DT_OK("hgr_colsum");
    const int nrb = (rows + 511) / 512;
    dim3 g((cols + 63) / 64, nrb);
    hipStream_t s = (hipStream_t)stream;
    if (x_f32 && rows <= 32 && cols % 4 == 0 && ldx % 4 == 0 && hgr_aligned(x, 16) && hgr_aligned(out, 16)) {
        const int64_t c4 = cols / 4;
        hipLaunchKernelGGL(colsum_rows_f32, dim3(grid1(c4)), dim3(256), 0, s, (const float *)x, ldx, rows, c4, out, accumulate, alpha);
        HGR_CHECK_LAUNCH("hgr_colsum");
        return HGR_OK;
    }
    const bool vec = cols % 64 == 0 && hgr_aligned(x, 16) && ldx % (x_f32 ? 4 : 8) == 0 && rows >= 32;
    if (vec) {
        if (x_f32) hipLaunchKernelGGL((colsum_partial_vec<HGR_BF16, true>), g, dim3(256), 0, s, x, ldx, rows, cols, scratch);
        else if (dtype == HGR_BF16) hipLaunchKernelGGL((colsum_partial_vec<HGR_BF16, false>), g, dim3(256), 0, s, x, ldx, rows, cols, scratch);
        else hipLaunchKernelGGL((colsum_partial_vec<HGR_F16, false>), g, dim3(256), 0, s, x, ldx, rows, cols, scratch);
    } else if (x_f32) hipLaunchKernelGGL((colsum_partial<HGR_BF16, true>), g, dim3(256), 0, s, x, ldx, rows, cols, scratch);
    else if (dtype == HGR_BF16) hipLaunchKernelGGL((colsum_partial<HGR_BF16, false>), g, dim3(256), 0, s, x, ldx, rows, cols, scratch);
    else hipLaunchKernelGGL((colsum_partial<HGR_F16, false>), g, dim3(256), 0, s, x, ldx, rows, cols, scratch);
    colsum_finish(scratch, nrb, cols, out, accumulate, alpha, s);
    HGR_CHECK_LAUNCH("hgr_colsum");
    return HGR_OK;
}

extern "C" int hgr_transpose16_colsum(const void *x, int64_t ldx, void *y, int64_t ldy, int rows, int cols, int dtype,
                                      float *out, int accumulate, float alpha, float *scratch, void *stream) {
    HGR_REQUIRE(x && y && out && scratch && rows >= 1 && cols >= 1 && ldx >= cols && ldy >= rows, "hgr_transpose16_colsum: bad arguments");
    HGR_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && hgr_aligned(x, 16) && hgr_aligned(y, 16), "hgr_transpose16_colsum: 16-byte aligned rows required (ldx, ldy multiples of 8)");
    DT_OK("hgr_transpose16_colsum");
    const int nrb = (rows + 63) / 64;
    dim3 grid((cols + 63) / 64, nrb);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HGR_BF16) hipLaunchKernelGGL(transpose16<1>, grid, dim3(256), 0, s, (const unsigned short *)x, ldx, (unsigned short *)y, ldy, rows, cols, scratch);
    else hipLaunchKernelGGL(transpose16<2>, grid, dim3(256), 0, s, (const unsigned short *)x, ldx, (unsigned short *)y, ldy, rows, cols, scratch);
    colsum_finish(scratch, nrb, cols, out, accumulate, alpha, s);
    HGR_CHECK_LAUNCH("hgr_transpose16_colsum");
    return HGR_OK;
}

extern "C" int hgr_cast16(const float *x, void *y, int64_t n, int dtype, void *stream) {
    HGR_REQUIRE(x && y && n >= 4 && n % 4 == 0 && hgr_aligned(x, 16) && hgr_aligned(y, 8), "hgr_cast16: n %% 4 == 0 and aligned operands required");
    DT_OK("hgr_cast16");
    if (dtype == HGR_BF16) hipLaunchKernelGGL((cast16<HGR_BF16>), dim3(grid1(n / 4)), dim3(256), 0, (hipStream_t)stream, x, (__bf16 *)y, n / 4);
    else hipLaunchKernelGGL((cast16<HGR_F16>), dim3(grid1(n / 4)), dim3(256), 0, (hipStream_t)stream, x, (_Float16 *)y, n / 4);
    HGR_CHECK_LAUNCH("hgr_cast16");
    return HGR_OK;
}

extern "C" int hgr_cast16_transpose(const float *x, int64_t ldx, void *y, int64_t ldy, void *yt, int64_t ldyt, int rows, int cols, int dtype, void *stream) {
    HGR_REQUIRE(x && y && yt && rows >= 1 && cols >= 1 && ldx >= cols && ldy >= cols && ldyt >= rows, "hgr_cast16_transpose: bad arguments");
    DT_OK("hgr_cast16_transpose");
    const int vec = ldx % 4 == 0 && ldy % 8 == 0 && ldyt % 8 == 0 && hgr_aligned(x, 16) && hgr_aligned(y, 16) && hgr_aligned(yt, 16);
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (dtype == HGR_BF16) hipLaunchKernelGGL((cast16_transpose<HGR_BF16>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (__bf16 *)y, ldy, (__bf16 *)yt, ldyt, rows, cols, vec);
    else hipLaunchKernelGGL((cast16_transpose<HGR_F16>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (_Float16 *)y, ldy, (_Float16 *)yt, ldyt, rows, cols, vec);
    HGR_CHECK_LAUNCH("hgr_cast16_transpose");
    return HGR_OK;
}

extern "C" int hgr_quickgelu16(const void *a, const void *du, void *out, int64_t n, int backward, int dtype, void *stream) {
    HGR_REQUIRE(a && out && (!backward || du) && n >= 8 && n % 8 == 0 && hgr_aligned(a, 16) && hgr_aligned(out, 16) && hgr_aligned(du, 16), "hgr_quickgelu16: bad arguments");
    DT_OK("hgr_quickgelu16");
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(grid1(n / 8));
    if (dtype == HGR_BF16) {
        if (backward) hipLaunchKernelGGL((gelu16<HGR_BF16, true>), g, dim3(256), 0, s, (const __bf16 *)a, (const __bf16 *)du, (__bf16 *)out, n / 8);
        else hipLaunchKernelGGL((gelu16<HGR_BF16, false>), g, dim3(256), 0, s, (const __bf16 *)a, (const __bf16 *)du, (__bf16 *)out, n / 8);
    } else {
        if (backward) hipLaunchKernelGGL((gelu16<HGR_F16, true>), g, dim3(256), 0, s, (const _Float16 *)a, (const _Float16 *)du, (_Float16 *)out, n / 8);
        else hipLaunchKernelGGL((gelu16<HGR_F16, false>), g, dim3(256), 0, s, (const _Float16 *)a, (const _Float16 *)du, (_Float16 *)out, n / 8);
    }
    HGR_CHECK_LAUNCH("hgr_quickgelu16");
    return HGR_OK;
}

static int layernorm_bwd_entry(const char *name, const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, void *dx16, float *dgamma, float *dbeta,
                               float *scratch, int rows, int W, int64_t row_mul, const int32_t *row_idx, float eps, int dtype, void *stream, float *dx16_colsum = nullptr) {
    HGR_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && scratch, "%s: null operand", name);
    HGR_REQUIRE(rows >= 1 && W >= 4 && W % 4 == 0 && W <= 4096 && row_mul >= 1, "%s: rows=%d W=%d unsupported", name, rows, W);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "%s: bad dtype %d", name, dtype);
    HGR_REQUIRE(hgr_aligned(dx16, 8), "%s: dx16 must be 8-byte aligned", name);
    const int blocks = rows < 4 * 512 ? (rows + 3) / 4 : 512;          // <= 2048 waves, each leaves one partial row
    const int nw = blocks * 4;
    float *pg = scratch, *pb = scratch + (int64_t)nw * W;
    float *cs = pb + (int64_t)nw * W;                                   // colsum scratch: ceil(nw/512) * W floats
    float *pc = dx16_colsum ? cs + ((int64_t)(nw + 511) / 512) * W : nullptr;   // third partial matrix behind it (hgr_layernorm_bwd_scratch_floats counts it)
    hipStream_t s = (hipStream_t)stream;
    const int nvl = (W / 4 + 63) / 64;
#define HGR_LNB(NVV)                                                                                                             \
    do {                                                                                                                         \
        if (dy_f32 && dtype == HGR_BF16) hipLaunchKernelGGL((layernorm_bwd<HGR_BF16, true, NVV>), dim3(blocks), dim3(256), 0, s, dy, x, gamma, dx, pg, pb, rows, W, row_mul, row_idx, eps, (__bf16 *)dx16, pc); \
        else if (dy_f32) hipLaunchKernelGGL((layernorm_bwd<HGR_F16, true, NVV>), dim3(blocks), dim3(256), 0, s, dy, x, gamma, dx, pg, pb, rows, W, row_mul, row_idx, eps, (_Float16 *)dx16, pc); \
        else if (dtype == HGR_BF16) hipLaunchKernelGGL((layernorm_bwd<HGR_BF16, false, NVV>), dim3(blocks), dim3(256), 0, s, dy, x, gamma, dx, pg, pb, rows, W, row_mul, row_idx, eps, (__bf16 *)dx16, pc); \
        else hipLaunchKernelGGL((layernorm_bwd<HGR_F16, false, NVV>), dim3(blocks), dim3(256), 0, s, dy, x, gamma, dx, pg, pb, rows, W, row_mul, row_idx, eps, (_Float16 *)dx16, pc); \
    } while (0)
    if (nvl <= 1) HGR_LNB(1); else if (nvl <= 2) HGR_LNB(2); else if (nvl <= 4) HGR_LNB(4); else if (nvl <= 8) HGR_LNB(8); else HGR_LNB(16);
#undef HGR_LNB
    HGR_CHECK_LAUNCH(name);
    if (hgr_aligned(dgamma, 16) && hgr_aligned(dbeta, 16) && hgr_aligned(dx16_colsum, 16)) {      // (W % 4 == 0 is required above)
        hipLaunchKernelGGL(ln_bwd_reduce3, dim3((W + 63) / 64, pc ? 3 : 2), dim3(256), 0, s, pg, pb, pc, nw, W, dgamma, dbeta, dx16_colsum);
        HGR_CHECK_LAUNCH(name);
        return HGR_OK;
    }
    int rc = hgr_colsum(pg, W, nw, W, 1, HGR_BF16, dgamma, 1, 1.0f, cs, stream);
    if (rc) return rc;
    rc = hgr_colsum(pb, W, nw, W, 1, HGR_BF16, dbeta, 1, 1.0f, cs, stream);
    if (rc || !pc) return rc;
    return hgr_colsum(pc, W, nw, W, 1, HGR_BF16, dx16_colsum, 1, 1.0f, cs, stream);
}

extern "C" int hgr_layernorm_bwd(const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, float *dgamma, float *dbeta, float *scratch,
                                 int rows, int W, int64_t row_mul, const int32_t *row_idx, float eps, int dtype, void *stream) {
    return layernorm_bwd_entry("hgr_layernorm_bwd", dy, dy_f32, x, gamma, dx, nullptr, dgamma, dbeta, scratch, rows, W, row_mul, row_idx, eps, dtype, stream);
}

extern "C" int hgr_layernorm_bwd_cast(const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, void *dx16, float *dgamma, float *dbeta,
                                      float *scratch, int rows, int W, int64_t row_mul, const int32_t *row_idx, float eps, int dtype, void *stream) {
    HGR_REQUIRE(dx16, "hgr_layernorm_bwd_cast: null dx16");
    return layernorm_bwd_entry("hgr_layernorm_bwd_cast", dy, dy_f32, x, gamma, dx, dx16, dgamma, dbeta, scratch, rows, W, row_mul, row_idx, eps, dtype, stream);
}

// ... and dx16_colsum[c] += sum over the rows of dx16[:, c] as rounded: the bias gradient of the Linear whose output gradient dx16 is
// (out_proj / c_proj of a residual block: their dY IS the residual-stream gradient), without an hgr_colsum pass over dx16
extern "C" int hgr_layernorm_bwd_cast_colsum(const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, void *dx16, float *dgamma, float *dbeta,
                                             float *dx16_colsum, float *scratch, int rows, int W, int64_t row_mul, const int32_t *row_idx, float eps, int dtype, void *stream) {
    HGR_REQUIRE(dx16 && dx16_colsum, "hgr_layernorm_bwd_cast_colsum: null dx16 / dx16_colsum");
    return layernorm_bwd_entry("hgr_layernorm_bwd_cast_colsum", dy, dy_f32, x, gamma, dx, dx16, dgamma, dbeta, scratch, rows, W, row_mul, row_idx, eps, dtype, stream, dx16_colsum);
}

extern "C" int64_t hgr_layernorm_bwd_scratch_floats(int rows, int W) {
    const int blocks = rows < 4 * 512 ? (rows + 3) / 4 : 512;
    const int64_t nw = (int64_t)blocks * 4;
    return 3 * nw * W + ((nw + 511) / 512) * W;       // dgamma / dbeta / dx16 column-sum partial rows + the colsum scratch
}

static int mha_bwd_entry(const char *name, const void *qkv, const void *out, const void *dout, void *dqkv, const float *stats,
                         int B, int L, int heads, int causal, int dtype, void *stream, float *colpart = nullptr) {
    HGR_REQUIRE(qkv && out && dout && dqkv && B >= 1 && heads >= 1, "%s: bad arguments", name);
    HGR_REQUIRE(L >= 1 && L <= 320, "%s: L=%d unsupported (L <= 320)", name, L);
    HGR_REQUIRE(hgr_aligned(stats, 8), "%s: stats must be 8-byte aligned", name);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "%s: bad dtype %d", name, dtype);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(B * heads);
    if (L <= 32) {           // one 32 x 32 block per (batch, head): the statistics are recomputed in registers, `stats` is not needed
        if (dtype == HGR_BF16) {
            if (causal) hipLaunchKernelGGL((mha_bwd_wave<HGR_BF16, true>), g, dim3(64), 0, s, (const __bf16 *)qkv, (const __bf16 *)dout, (__bf16 *)dqkv, L, heads, colpart);
            else hipLaunchKernelGGL((mha_bwd_wave<HGR_BF16, false>), g, dim3(64), 0, s, (const __bf16 *)qkv, (const __bf16 *)dout, (__bf16 *)dqkv, L, heads, colpart);
        } else {
            if (causal) hipLaunchKernelGGL((mha_bwd_wave<HGR_F16, true>), g, dim3(64), 0, s, (const _Float16 *)qkv, (const _Float16 *)dout, (_Float16 *)dqkv, L, heads, colpart);
            else hipLaunchKernelGGL((mha_bwd_wave<HGR_F16, false>), g, dim3(64), 0, s, (const _Float16 *)qkv, (const _Float16 *)dout, (_Float16 *)dqkv, L, heads, colpart);
        }
    }
    else {
        const float2 *st = (const float2 *)stats;
#define HGR_MT(CAUS)                                                                                                              \
    do {                                                                                                                          \
        if (dtype == HGR_BF16) hipLaunchKernelGGL((mha_bwd_tiled<HGR_BF16, CAUS>), g, dim3(256), 0, s, (const __bf16 *)qkv, (const __bf16 *)out, (const __bf16 *)dout, (__bf16 *)dqkv, L, heads, st, colpart); \
        else hipLaunchKernelGGL((mha_bwd_tiled<HGR_F16, CAUS>), g, dim3(256), 0, s, (const _Float16 *)qkv, (const _Float16 *)out, (const _Float16 *)dout, (_Float16 *)dqkv, L, heads, st, colpart); \
    } while (0)
        if (causal) HGR_MT(true); else HGR_MT(false);
#undef HGR_MT
    }
    HGR_CHECK_LAUNCH(name);
    return HGR_OK;
}

extern "C" int hgr_mha_bwd(const void *qkv, const void *out, const void *dout, void *dqkv, int B, int L, int heads, int causal, int dtype, void *stream) {
    return mha_bwd_entry("hgr_mha_bwd", qkv, out, dout, dqkv, nullptr, B, L, heads, causal, dtype, stream);
}

extern "C" int hgr_mha_bwd_stats(const void *qkv, const void *out, const void *dout, void *dqkv, const float *stats,
                                 int B, int L, int heads, int causal, int dtype, void *stream) {
    HGR_REQUIRE(stats, "hgr_mha_bwd_stats: null stats");
    return mha_bwd_entry("hgr_mha_bwd_stats", qkv, out, dout, dqkv, stats, B, L, heads, causal, dtype, stream);
}

// ... and colsum_part[b][3W] (fp32, every entry written) = column sums over sequence b's L rows of dqkv AS ROUNDED: summed over b
// (hgr_colsum) that is the in_proj bias gradient, without a pass over the [B*L, 3W] gradient.  stats may be NULL (recomputed).
extern "C" int hgr_mha_bwd_colsum(const void *qkv, const void *out, const void *dout, void *dqkv, const float *stats, float *colsum_part,
                                  int B, int L, int heads, int causal, int dtype, void *stream) {
    HGR_REQUIRE(colsum_part, "hgr_mha_bwd_colsum: null colsum_part");
    return mha_bwd_entry("hgr_mha_bwd_colsum", qkv, out, dout, dqkv, stats, B, L, heads, causal, dtype, stream, colsum_part);
}

extern "C" int hgr_ce_rows(const float *logits, int64_t ld, const int32_t *labels, int rows, int n, float gscale, float *loss_rows, float *dlogits, int64_t ldd, void *stream) {
    HGR_REQUIRE(logits && labels && loss_rows && rows >= 1 && n >= 1 && ld >= n && (!dlogits || ldd >= n), "hgr_ce_rows: bad arguments");
    hipLaunchKernelGGL(ce_rows, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, rows, n, gscale, loss_rows, dlogits, ldd);
    HGR_CHECK_LAUNCH("hgr_ce_rows");
    return HGR_OK;
}

extern "C" int hgr_l2norm_bwd(const float *x, const float *dy, float *dx, int rows, int D, int accumulate, void *stream) {
    HGR_REQUIRE(x && dy && dx && rows >= 1 && D >= 1, "hgr_l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, D, accumulate);
    HGR_CHECK_LAUNCH("hgr_l2norm_bwd");
    return HGR_OK;
}

extern "C" int hgr_matmul_f32(const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn, float *C, int64_t ldc,
                              int M, int N, int K, float alpha, int accumulate, void *stream) {
    HGR_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && ldc >= N, "hgr_matmul_f32: bad arguments");
    // 16-byte loads need the contiguous dimension's rows to start 16-byte aligned
    const int vecA = hgr_aligned(A, 16) && ((sak == 1 && sam % 4 == 0) || (sam == 1 && sak % 4 == 0));
    const int vecB = hgr_aligned(B, 16) && ((sbk == 1 && sbn % 4 == 0) || (sbn == 1 && sbk % 4 == 0));
    hipLaunchKernelGGL(matmul_f32, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, accumulate, vecA, vecB);
    HGR_CHECK_LAUNCH("hgr_matmul_f32");
    return HGR_OK;
}

extern "C" int hgr_embed_scatter_add(const int64_t *tokens, int64_t ld_tokens, const float *dx, float *dtable, int n, int L, int W, int vocab, void *stream) {
    HGR_REQUIRE(tokens && dx && dtable && n >= 1 && L >= 1 && W >= 1 && vocab >= 1 && ld_tokens >= L, "hgr_embed_scatter_add: bad arguments");
    hipLaunchKernelGGL(embed_scatter_add, dim3(grid1((int64_t)n * L * W)), dim3(256), 0, (hipStream_t)stream, tokens, ld_tokens, dx, dtable, n, L, W, vocab);
    HGR_CHECK_LAUNCH("hgr_embed_scatter_add");
    return HGR_OK;
}

namespace {
// out (+)= alpha * sum_i a[i] b[i]: ONE workgroup of 1024 threads, fixed summation order (deterministic); meant for
// vectors of at most a few 100 K elements (d logit_scale = sum(dlogits * logits) of the loss head)
__global__ __launch_bounds__(1024) void dot_f32(const float *__restrict__ a, const float *__restrict__ b, int64_t n, float *__restrict__ out,
                                                float alpha, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        *out = (accumulate ? *out : 0.f) + alpha * t;
    }
}
}  // namespace

extern "C" int hgr_dot_f32(const float *a, const float *b, int64_t n, float *out, float alpha, int accumulate, void *stream) {
    HGR_REQUIRE(a && b && out && n >= 1, "hgr_dot_f32: bad arguments");
    hipLaunchKernelGGL(dot_f32, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, out, alpha, accumulate);
    HGR_CHECK_LAUNCH("hgr_dot_f32");
    return HGR_OK;
}

extern "C" int hgr_sumsq(const float *x, int64_t n, float *out, void *stream) {
    HGR_REQUIRE(x && out && n >= 1, "hgr_sumsq: bad arguments");
    static std::atomic<unsigned> next_slot{0};                  // consecutive launches take consecutive scratch slots (see the kernel)
    const int slot = (int)(next_slot.fetch_add(1u, std::memory_order_relaxed) % SUMSQ_SLOTS);
    hipLaunchKernelGGL(sumsq, dim3(grid1(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, out, slot);
    HGR_CHECK_LAUNCH("hgr_sumsq");
    return HGR_OK;
}

extern "C" int hgr_adamw(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                         int step, const float *sumsq_total, float max_norm, float grad_scale, void *stream) {
    HGR_REQUIRE(p && g && m && v && n >= 1 && step >= 1, "hgr_adamw: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw, dim3(grid1(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, sumsq_total, max_norm, grad_scale);
    HGR_CHECK_LAUNCH("hgr_adamw");
    return HGR_OK;
}

extern "C" int hgr_rows_axpy(float *dst, int64_t dst_mul, const int32_t *dst_idx, const float *src, int rows, int W, float alpha, void *stream) {
    HGR_REQUIRE(dst && src && rows >= 1 && W >= 1 && dst_mul >= 0 && (dst_mul >= 1 || dst_idx), "hgr_rows_axpy: bad arguments");
    hipLaunchKernelGGL(rows_axpy, dim3(grid1((int64_t)rows * W)), dim3(256), 0, (hipStream_t)stream, dst, dst_mul, dst_idx, src, rows, W, alpha);
    HGR_CHECK_LAUNCH("hgr_rows_axpy");
    return HGR_OK;
}

extern "C" int hgr_rows_gather(float *dst, const float *src, const int32_t *idx, int rows, int W, void *stream) {
    HGR_REQUIRE(dst && src && idx && rows >= 1 && W >= 1, "hgr_rows_gather: bad arguments");
    hipLaunchKernelGGL(rows_gather, dim3(grid1((int64_t)rows * W)), dim3(256), 0, (hipStream_t)stream, dst, src, idx, rows, W);
    HGR_CHECK_LAUNCH("hgr_rows_gather");
    return HGR_OK;
}

extern "C" int hgr_ctx_splice(float *x, const float *ctx, const float *positional_embedding, int n, int L, int W, int n_ctx, void *stream) {
    HGR_REQUIRE(x && ctx && positional_embedding && n >= 1 && n_ctx >= 1 && L >= n_ctx + 2 && W >= 1, "hgr_ctx_splice: bad arguments (L >= n_ctx + 2)");
    hipLaunchKernelGGL(ctx_splice, dim3(grid1((int64_t)n * n_ctx * W)), dim3(256), 0, (hipStream_t)stream, x, ctx, positional_embedding, n, L, W, n_ctx);
    HGR_CHECK_LAUNCH("hgr_ctx_splice");
    return HGR_OK;
}

extern "C" int hgr_ctx_splice_bwd(float *dx, float *dctx, int n, int L, int W, int n_ctx, void *stream) {
    HGR_REQUIRE(dx && dctx && n >= 1 && n_ctx >= 1 && L >= n_ctx + 2 && W >= 1, "hgr_ctx_splice_bwd: bad arguments");
    hipLaunchKernelGGL(ctx_splice_bwd, dim3((W + 63) / 64, n_ctx), dim3(256), 0, (hipStream_t)stream, dx, dctx, n, L, W, n_ctx);
    HGR_CHECK_LAUNCH("hgr_ctx_splice_bwd");
    return HGR_OK;
}
