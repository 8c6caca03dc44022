// Selection over the logits rows: top-k over a column subset (main.py:136-139,157) and the
// level-segmented arg-max of the hierarchy metrics (main.py:162-176).  Index work: results are
// defined bit-exactly (largest value first, ties to the lowest subset position).
#include "hgr_common.h"
#include <math.h>

namespace {

struct Best { float v; int p; };

__device__ __forceinline__ bool better(float v, int p, float bv, int bp) { return v > bv || (v == bv && p < bp); }

__device__ __forceinline__ Best wave_best(Best b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(b.v, o);
        const int op = __shfl_xor(b.p, o);
        if (better(ov, op, b.v, b.p)) { b.v = ov; b.p = op; }
    }
    return b;
}

// One workgroup per row.  The row's subset is staged once in LDS (<= 40000 floats); every thread
// keeps the best of its own strided slice; each of the k rounds is one block reduction, and only
// the thread that owned the winner rescans its slice.
__global__ __launch_bounds__(256) void topk_rows(const float *__restrict__ logits, int64_t ld, const int32_t *__restrict__ cols,
                                                 int n_cols, int k, int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    float *vals = (float *)dyn;
    __shared__ float s_v[4];
    __shared__ int s_p[4];
    __shared__ int s_win;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const float *lr = logits + (int64_t)row * ld;
    for (int p = tid; p < n_cols; p += 256) vals[p] = lr[cols ? cols[p] : p];
    __syncthreads();

    auto scan = [&]() {
        Best b = {-INFINITY, 0x7fffffff};
        for (int p = tid; p < n_cols; p += 256) {
            const float v = vals[p];
            if (v > b.v) { b.v = v; b.p = p; }       // ascending p: strict '>' keeps the lowest position
        }
        return b;
    };
    Best mine = scan();
    for (int j = 0; j < k; ++j) {
        const Best w = wave_best(mine);
        if (lane == 0) { s_v[wave] = w.v; s_p[wave] = w.p; }
        __syncthreads();
        if (tid == 0) {
            Best b = {s_v[0], s_p[0]};
#pragma unroll
            for (int i = 1; i < 4; ++i)
                if (better(s_v[i], s_p[i], b.v, b.p)) { b.v = s_v[i]; b.p = s_p[i]; }
            s_win = b.p;
            const bool ok = b.p < n_cols;
            out_idx[(int64_t)row * k + j] = ok ? (cols ? cols[b.p] : b.p) : -1;
            if (out_val) out_val[(int64_t)row * k + j] = b.v;
            if (ok) vals[b.p] = -INFINITY;
        }
        __syncthreads();
        if (mine.p == s_win) mine = scan();           // exactly one thread owned it
        __syncthreads();
    }
}

// One workgroup per row, one coalesced pass.  v_p = in level l ? logit : -1 ; first maximum wins.
// Positions outside level l all carry the filler -1, the first of them is position 0 unless position 0
// itself is in level l (then the first position of another level).
template <int NLV>
__global__ __launch_bounds__(256) void level_argmax(const float *__restrict__ logits, int64_t ld, const int32_t *__restrict__ cols,
                                                    int n_cols, const int32_t *__restrict__ depth, int n_levels, int32_t *__restrict__ out) {
    __shared__ float s_v[4][NLV];
    __shared__ int s_p[4][NLV];
    __shared__ int s_fd[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const float *lr = logits + (int64_t)row * ld;
    float bv[NLV];
    int bp[NLV];
#pragma unroll
    for (int l = 0; l < NLV; ++l) { bv[l] = -INFINITY; bp[l] = 0x7fffffff; }
    const int lvl0 = depth[cols ? cols[0] : 0];
    int fd = 0x7fffffff;                                // first position whose level differs from position 0's
    for (int p = tid; p < n_cols; p += 256) {
        const int c = cols ? cols[p] : p;
        const int lv = depth[c];
        const float v = lr[c];
        if (lv != lvl0 && p < fd) fd = p;
#pragma unroll
        for (int l = 0; l < NLV; ++l)
            if (lv == l && v > bv[l]) { bv[l] = v; bp[l] = p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fd = min(fd, __shfl_xor(fd, o));
#pragma unroll
    for (int l = 0; l < NLV; ++l) {
        Best b = {bv[l], bp[l]};
        b = wave_best(b);
        if (lane == 0) { s_v[wave][l] = b.v; s_p[wave][l] = b.p; }
    }
    if (lane == 0) s_fd[wave] = fd;
    __syncthreads();
    if (tid < n_levels) {
        const int l = tid;
        Best b = {s_v[0][l], s_p[0][l]};
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (better(s_v[i][l], s_p[i][l], b.v, b.p)) { b.v = s_v[i][l]; b.p = s_p[i][l]; }
        const int fdd = min(min(s_fd[0], s_fd[1]), min(s_fd[2], s_fd[3]));
        const int fo = (l != lvl0) ? 0 : fdd;           // first filler position (may not exist)
        int win;
        const bool has_c = b.p < n_cols, has_f = fo < n_cols;
        if (has_c && (!has_f || b.v > -1.0f || (b.v == -1.0f && b.p < fo))) win = b.p;
        else win = has_f ? fo : b.p;
        out[(int64_t)row * n_levels + l] = cols ? cols[win] : win;
    }
}

}  // namespace

extern "C" int hgr_topk_rows(const float *logits, int64_t ld, const int32_t *cols, int n_cols, int k,
                             int32_t *out_idx, float *out_val, int rows, void *stream) {
    HGR_REQUIRE(logits && out_idx, "hgr_topk_rows: null operand");
    HGR_REQUIRE(rows >= 1 && n_cols >= 1 && n_cols <= 40000 && k >= 1 && k <= 32 && k <= n_cols,
                "hgr_topk_rows: rows=%d n_cols=%d k=%d unsupported (n_cols <= 40000, k <= min(32, n_cols))", rows, n_cols, k);
    HGR_REQUIRE(ld >= 1, "hgr_topk_rows: bad ld");
    static size_t granted = 0;
    const size_t bytes = ((size_t)n_cols * 4 + 15) & ~(size_t)15;
    if (bytes > granted) {
        hipError_t e = hipFuncSetAttribute((const void *)topk_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return hgr_set_error(HGR_ELAUNCH, "hgr_topk_rows: cannot reserve %zu B of LDS: %s", bytes, hipGetErrorString(e));
        granted = bytes;
    }
    hipLaunchKernelGGL(topk_rows, dim3(rows), dim3(256), bytes, (hipStream_t)stream, logits, ld, cols, n_cols, k, out_idx, out_val);
    HGR_CHECK_LAUNCH("hgr_topk_rows");
    return HGR_OK;
}

extern "C" int hgr_level_argmax(const float *logits, int64_t ld, const int32_t *cols, int n_cols,
                                const int32_t *depth, int n_levels, int32_t *out, int rows, void *stream) {
    HGR_REQUIRE(logits && depth && out, "hgr_level_argmax: null operand");
    HGR_REQUIRE(rows >= 1 && n_cols >= 1 && n_levels >= 1 && n_levels <= 32, "hgr_level_argmax: rows=%d n_cols=%d n_levels=%d unsupported (n_levels <= 32)", rows, n_cols, n_levels);
    hipStream_t s = (hipStream_t)stream;
    if (n_levels <= 16) hipLaunchKernelGGL((level_argmax<16>), dim3(rows), dim3(256), 0, s, logits, ld, cols, n_cols, depth, n_levels, out);
    else hipLaunchKernelGGL((level_argmax<32>), dim3(rows), dim3(256), 0, s, logits, ld, cols, n_cols, depth, n_levels, out);
    HGR_CHECK_LAUNCH("hgr_level_argmax");
    return HGR_OK;
}
