// Selection over the logits rows: top-k over a column subset (main.py:136-139,157) and the
// level-segmented arg-max of the hierarchy metrics (main.py:162-176).  Index work: results are
// defined bit-exactly (largest value first, ties to the lowest subset position).
#include "hgr_common.h"
#include <stdlib.h>
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

// One workgroup per row; the row's subset is staged once in LDS (<= 40000 floats).
//   1. every thread finds the best element of its strided slice;
//   2. the k-th best of those 256 slice maxima is a lower bound t of the true k-th largest value
//      (k distinct elements are >= t), found by rank counting (256 broadcast LDS reads per thread);
//   3. only slices whose maximum reaches t can hold candidates: they append every element >= t
//      (by (value, position) order) to a small LDS list - typically k .. 2k entries;
//   4. candidates are ranked by brute-force comparison and ranks < k are written out.
// Exact for any input (ties broken towards the lowest position); if the candidate list overflows
// (adversarial, heavily duplicated data) the block falls back to k rounds of block-wide arg-max.
constexpr int TOPK_CAND = 1024;

__global__ __launch_bounds__(256) void topk_rows(const float *__restrict__ logits, int64_t ld, const int32_t *__restrict__ cols,
                                                 int n_cols, int k, int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    float *vals = (float *)dyn;
    __shared__ float s_mv[256];
    __shared__ int s_mp[256];
    __shared__ float s_cv[TOPK_CAND];
    __shared__ int s_cp[TOPK_CAND];
    __shared__ float s_t;
    __shared__ int s_tp, s_cnt, s_win;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const float *lr = logits + (int64_t)row * ld;
    Best mine = {-INFINITY, 0x7fffffff};
    for (int p = tid; p < n_cols; p += 256) {
        const float v = lr[cols ? cols[p] : p];
        vals[p] = v;
        if (v > mine.v) { mine.v = v; mine.p = p; }      // ascending p: strict '>' keeps the lowest position
    }
    s_mv[tid] = mine.v; s_mp[tid] = mine.p;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // rank of my slice maximum among the 256 maxima (threads with an empty slice hold (-inf, INT_MAX))
    int rank = 0;
    for (int j = 0; j < 256; ++j) rank += better(s_mv[j], s_mp[j], mine.v, mine.p) ? 1 : 0;
    const int kth = min(k, n_cols) - 1;                    // n_cols >= k is guaranteed by the host; < 256 slices may be empty
    if (rank == min(kth, min(n_cols, 256) - 1)) { s_t = mine.v; s_tp = mine.p; }
    __syncthreads();
    const float t = s_t; const int tp = s_tp;
    // candidates: everything not worse than (t, tp); only slices whose maximum qualifies can contain any
    if (!better(t, tp, mine.v, mine.p)) {
        for (int p = tid; p < n_cols; p += 256) {
            const float v = vals[p];
            if (!better(t, tp, v, p)) {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < TOPK_CAND) { s_cv[slot] = v; s_cp[slot] = p; }
            }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt <= TOPK_CAND) {
        for (int c = tid; c < cnt; c += 256) {
            const float v = s_cv[c]; const int p = s_cp[c];
            int rk = 0;
            for (int j = 0; j < cnt; ++j) rk += better(s_cv[j], s_cp[j], v, p) ? 1 : 0;
            if (rk < k) {
                out_idx[(int64_t)row * k + rk] = cols ? cols[p] : p;
                if (out_val) out_val[(int64_t)row * k + rk] = v;
            }
        }
        return;
    }
    // fallback: k rounds of block arg-max with removal
    auto scan = [&]() {
        Best b = {-INFINITY, 0x7fffffff};
        for (int p = tid; p < n_cols; p += 256) {
            const float v = vals[p];
            if (v > b.v) { b.v = v; b.p = p; }
        }
        return b;
    };
    for (int j = 0; j < k; ++j) {
        const Best w = wave_best(mine);
        if (lane == 0) { s_mv[wave] = w.v; s_mp[wave] = w.p; }
        __syncthreads();
        if (tid == 0) {
            Best b = {s_mv[0], s_mp[0]};
#pragma unroll
            for (int i = 1; i < 4; ++i)
                if (better(s_mv[i], s_mp[i], b.v, b.p)) { b.v = s_mv[i]; b.p = s_mp[i]; }
            s_win = b.p;
            const bool ok = b.p < n_cols;
            out_idx[(int64_t)row * k + j] = ok ? (cols ? cols[b.p] : b.p) : -1;
            if (out_val) out_val[(int64_t)row * k + j] = b.v;
            if (ok) vals[b.p] = -INFINITY;
        }
        __syncthreads();
        if (mine.p == s_win) mine = scan();
        __syncthreads();
    }
}

// One workgroup per row, one coalesced pass.  v_p = in level l ? logit : -1 ; first maximum wins.
// Positions outside level l all carry the filler -1, the first of them is position 0 unless position 0
// itself is in level l (then the first position of another level).
template <int NLV>
__global__ __launch_bounds__(256) void level_argmax(const float *__restrict__ logits, int64_t ld, const int32_t *__restrict__ cols,
                                                    int n_cols, const int32_t *__restrict__ depth, int n_levels, int32_t *__restrict__ out, int32_t *__restrict__ out_top1) {
    __shared__ float s_v[4][NLV];
    __shared__ float s_bv[NLV];
    __shared__ int s_bp[NLV];
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
    for (int p0 = tid; p0 < n_cols; p0 += 4 * 256) {       // 4 positions per trip: the dependent column -> (depth, logit) loads in two batches
        int c4[4], lv4[4];
        float v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int p = min(p0 + u * 256, n_cols - 1); c4[u] = cols ? cols[p] : p; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { lv4[u] = depth[c4[u]]; v4[u] = lr[c4[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * 256;
            if (p >= n_cols) break;
            const int lv = lv4[u];
            const float v = v4[u];
            if (lv != lvl0 && p < fd) fd = p;
#pragma unroll
            for (int l = 0; l < NLV; ++l)
                if (lv == l && v > bv[l]) { bv[l] = v; bp[l] = p; }
        }
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
        s_bv[l] = b.v; s_bp[l] = b.p;
    }
    if (out_top1) {
        // every subset column belongs to exactly one level, so the unmasked top-1 over the subset
        // (main.py:157) is the best of the per-level bests
        __syncthreads();
        if (tid == 0) {
            Best b = {-INFINITY, 0x7fffffff};
            for (int l = 0; l < n_levels; ++l)
                if (better(s_bv[l], s_bp[l], b.v, b.p)) { b.v = s_bv[l]; b.p = s_bp[l]; }
            out_top1[row] = b.p < n_cols ? (cols ? cols[b.p] : b.p) : -1;
        }
    }
}

// ---- fused evaluation of one logits row (T1 + T2 + T3 of main.py:136-176) ------------------------------------------
// One workgroup per row, tiny LDS footprint (8 workgroups per CU hide the load latency), the row (<= 160 KB) stays
// L2-resident between its three sweeps:
//   * level-segmented arg-max: ONE coalesced sweep over the row; every thread keeps the running best (value, position)
//     of every depth level in registers (statically unrolled over <= 16 / 32 levels), then one wave-shuffle reduction per
//     level and 4 partials through LDS.  (Measured alternatives: a row staged in LDS + CSR walk of the levels is bound by
//     one workgroup per CU and the index loads' latency, 175 us; folding into 12 LDS atomicMax slots serialises in the
//     per-CU LDS atomic unit, 160 us.)
//   * the unmasked top-1 = best of the level slots;
//   * top-k over the test columns with the threshold / candidate algorithm of topk_rows: the slice maxima come out of the
//     same first sweep, one more coalesced sweep collects the candidates.
// Both sweeps run over the COLUMNS (fully coalesced) with dense maps column -> subset position (`train_pos`, `test_pos`,
// -1 = not in the subset; the position only breaks ties); `lvl8[c]` = depth of node c; `filler_pos[l]` = first train
// position NOT at depth l (-1 if none): the reference fills those columns with -1 before its arg-max, so the filler
// competes (main.py:170-173).
constexpr int EV_NT = 1024;                 // threads per row: 16 waves, so 512 rows keep 8 waves per SIMD busy chip-wide
constexpr int EV_NW = EV_NT / 64;

__device__ __forceinline__ Best block_best16(Best b, float *s_v, int *s_p, int tid) {
    const Best w = wave_best(b);
    __syncthreads();
    if ((tid & 63) == 0) { s_v[tid >> 6] = w.v; s_p[tid >> 6] = w.p; }
    __syncthreads();
    Best r = {s_v[0], s_p[0]};
    for (int i = 1; i < EV_NW; ++i)
        if (better(s_v[i], s_p[i], r.v, r.p)) { r.v = s_v[i]; r.p = s_p[i]; }
    return r;
}

template <int NLV>
__global__ __launch_bounds__(EV_NT) void eval_rows(const float *__restrict__ logits, int64_t ld, int n_nodes, const unsigned char *__restrict__ lvl8,
                                                 const int32_t *__restrict__ train_pos, const int32_t *__restrict__ train_cols, int n_train, int n_levels,
                                                 const int32_t *__restrict__ filler_pos, const int32_t *__restrict__ test_pos, const int32_t *__restrict__ test_cols,
                                                 int n_test, int k, int32_t *__restrict__ out_level, int32_t *__restrict__ out_top1, int32_t *__restrict__ out_topk) {
    __shared__ float s_lv[EV_NW][NLV];
    __shared__ int s_lp[EV_NW][NLV];
    __shared__ float s_v[EV_NW];
    __shared__ int s_p[EV_NW];
    __shared__ float s_mv[256];
    __shared__ int s_mp[256];
    __shared__ float s_cv[TOPK_CAND];
    __shared__ int s_cp[TOPK_CAND];
    __shared__ float s_t;
    __shared__ int s_tp, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    const float *lr = logits + (int64_t)r * ld;
    if (tid == 0) s_cnt = 0;
    // sweep 1 (coalesced over the columns): per-thread running best of every level (registers, statically unrolled) and
    // this thread's best test element
    float bv[NLV];
    int bp[NLV];
#pragma unroll
    for (int l = 0; l < NLV; ++l) { bv[l] = -INFINITY; bp[l] = 0x7fffffff; }
    Best mine = {-INFINITY, 0x7fffffff};
    for (int c = tid; c < n_nodes; c += EV_NT) {
        const float v = lr[c] + 0.0f;
        const int tp = train_pos[c];
        const int lv = tp >= 0 ? (int)lvl8[c] : -1;
#pragma unroll
        for (int l = 0; l < NLV; ++l)
            if (lv == l && better(v, tp, bv[l], bp[l])) { bv[l] = v; bp[l] = tp; }
        if (k > 0) {
            const int te = test_pos[c];
            if (te >= 0 && better(v, te, mine.v, mine.p)) { mine.v = v; mine.p = te; }
        }
    }
#pragma unroll
    for (int l = 0; l < NLV; ++l) {
        Best b = {bv[l], bp[l]};
        b = wave_best(b);
        if (lane == 0) { s_lv[wave][l] = b.v; s_lp[wave][l] = b.p; }
    }
    // 256 slice maxima for the top-k threshold: best of each group of 4 consecutive threads
    {
        Best gmx = mine;
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
            const float ov = __shfl_xor(gmx.v, o);
            const int op = __shfl_xor(gmx.p, o);
            if (better(ov, op, gmx.v, gmx.p)) { gmx.v = ov; gmx.p = op; }
        }
        if ((tid & 3) == 0) { s_mv[tid >> 2] = gmx.v; s_mp[tid >> 2] = gmx.p; }
    }
    __syncthreads();
    if (wave == 0) {                                        // lane l < n_levels finishes level l; the whole wave takes part in the shuffles
        const int l = lane;
        Best b = {-INFINITY, 0x7fffffff};
        if (l < n_levels) {
            b = Best{s_lv[0][l], s_lp[0][l]};
            for (int i = 1; i < EV_NW; ++i)
                if (better(s_lv[i][l], s_lp[i][l], b.v, b.p)) { b.v = s_lv[i][l]; b.p = s_lp[i][l]; }
            const int fo = filler_pos[l];
            const bool has_c = b.p < n_train, has_f = fo >= 0;
            int win;
            if (has_c && (!has_f || b.v > -1.0f || (b.v == -1.0f && b.p < fo))) win = b.p;
            else win = has_f ? fo : b.p;
            out_level[(int64_t)r * n_levels + l] = train_cols[win];
        }
        const Best top = wave_best(b);                      // unmasked top-1 = best of the level bests
        if (lane == 0 && out_top1) out_top1[r] = top.p < n_train ? train_cols[top.p] : -1;
    }
    if (k <= 0) return;
    // top-k over the test subset: threshold = k-th best of the 256 slice maxima, then sweep 2 collects the candidates
    if (tid < 256) {
        const float mv = s_mv[tid]; const int mp = s_mp[tid];
        int rank = 0, nonempty = 0;
        for (int j = 0; j < 256; ++j) { rank += better(s_mv[j], s_mp[j], mv, mp) ? 1 : 0; nonempty += s_mp[j] != 0x7fffffff ? 1 : 0; }
        // k distinct elements >= t are guaranteed only if at least k slices are non-empty (clustered / tiny test sets can
        // put several test columns into one slice); otherwise every test element is a candidate
        if (nonempty < k) { if (tid == 0) { s_t = -INFINITY; s_tp = 0x7fffffff; } }
        else if (rank == k - 1) { s_t = mv; s_tp = mp; }
    }
    __syncthreads();
    const float t = s_t; const int tp0 = s_tp;
    for (int c = tid; c < n_nodes; c += EV_NT) {
        const int te = test_pos[c];
        if (te < 0) continue;
        const float v = lr[c] + 0.0f;
        if (!better(t, tp0, v, te)) {
            const int slot = atomicAdd(&s_cnt, 1);
            if (slot < TOPK_CAND) { s_cv[slot] = v; s_cp[slot] = te; }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt <= TOPK_CAND) {
        for (int c = tid; c < cnt; c += EV_NT) {
            const float v = s_cv[c]; const int p = s_cp[c];
            int rk = 0;
            for (int j = 0; j < cnt; ++j) rk += better(s_cv[j], s_cp[j], v, p) ? 1 : 0;
            if (rk < k) out_topk[(int64_t)r * k + rk] = test_cols[p];
        }
    } else {
        // heavily duplicated data: k rounds of block arg-max; an element is "removed" by requiring it to be worse than the
        // previous winner in (value, position) order, so nothing has to be written back
        Best last = {INFINITY, -1};
        for (int j = 0; j < k; ++j) {
            Best b = {-INFINITY, 0x7fffffff};
            for (int c = tid; c < n_nodes; c += EV_NT) {
                const int te = test_pos[c];
                if (te < 0) continue;
                const float v = lr[c] + 0.0f;
                if (better(last.v, last.p, v, te) && better(v, te, b.v, b.p)) { b.v = v; b.p = te; }
            }
            b = block_best16(b, s_v, s_p, tid);
            if (tid == 0) out_topk[(int64_t)r * k + j] = b.p < n_test ? test_cols[b.p] : -1;
            last = b;
            __syncthreads();
        }
    }
}

// ---- same contract, LDS-private accumulators (n_levels <= 16) -----------------------------------------------------------
constexpr int EL_NT = 512, EL_NW = EL_NT / 64, EL_NLV = 16;
constexpr int EL_U = 8;                 // columns per thread per trip of the two sweeps

__device__ __forceinline__ Best block_best8(Best b, float *s_v, int *s_p, int tid) {
    const Best w = wave_best(b);
    __syncthreads();
    if ((tid & 63) == 0) { s_v[tid >> 6] = w.v; s_p[tid >> 6] = w.p; }
    __syncthreads();
    Best r = {s_v[0], s_p[0]};
    for (int i = 1; i < EL_NW; ++i)
        if (better(s_v[i], s_p[i], r.v, r.p)) { r.v = s_v[i]; r.p = s_p[i]; }
    return r;
}

__global__ __launch_bounds__(EL_NT) void eval_rows_lds(const float *__restrict__ logits, int64_t ld, int n_nodes, const unsigned char *__restrict__ lvl8,
                                                 const int32_t *__restrict__ train_pos, const int32_t *__restrict__ train_cols, int n_train, int n_levels,
                                                 const int32_t *__restrict__ filler_pos, const int32_t *__restrict__ test_pos, const int32_t *__restrict__ test_cols,
                                                 int n_test, int k, int32_t *__restrict__ out_level, int32_t *__restrict__ out_top1, int32_t *__restrict__ out_topk) {
    __shared__ unsigned long long acc[EL_NLV][EL_NT];      // per-thread, per-level running best as one orderable 64-bit key
    __shared__ unsigned long long s_key[EL_NLV];
    __shared__ float s_v[EL_NW];
    __shared__ int s_p[EL_NW];
    __shared__ float s_mv[256];
    __shared__ int s_mp[256];
    __shared__ float s_cv[TOPK_CAND];
    __shared__ int s_cp[TOPK_CAND];
    __shared__ float s_t;
    __shared__ int s_tp, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    const float *lr = logits + (int64_t)r * ld;
    if (tid == 0) s_cnt = 0;
    // sweep 1 (coalesced over the columns): a column belongs to ONE level, so instead of a 16-way compare chain over register
    // accumulators each thread keeps its per-level bests in its own LDS column (dynamic index = the level): one 8-byte
    // read, one 64-bit compare and a rare write per column.  key = (orderable(value) << 32) | (0x7fffffff - position):
    // unsigned max == "larger value, then smaller position"; 0 = empty.
#pragma unroll
    for (int l = 0; l < EL_NLV; ++l) acc[l][tid] = 0ull;
    Best mine = {-INFINITY, 0x7fffffff};
    // EL_U columns per trip, every load issued before the first use (a load under a branch is waited for on the spot); the sweep is a
    // chain of memory round trips with 2 workgroups per CU to hide them, so the trip count is what it costs: 11 trips of 4 -> 6 trips of 8
    for (int c0 = tid; c0 < n_nodes; c0 += EL_U * EL_NT) {
        float v4[EL_U]; int tp4[EL_U], lv4[EL_U], te4[EL_U];
#pragma unroll
        for (int u4 = 0; u4 < EL_U; ++u4) {
            const int c = min(c0 + u4 * EL_NT, n_nodes - 1);
            v4[u4] = lr[c]; tp4[u4] = train_pos[c]; lv4[u4] = lvl8[c]; te4[u4] = k > 0 ? test_pos[c] : -1;
        }
#pragma unroll
        for (int u4 = 0; u4 < EL_U; ++u4) {
            if (c0 + u4 * EL_NT >= n_nodes) break;
            const float v = v4[u4] + 0.0f;
            const int tp = tp4[u4];
            if (tp >= 0 && lv4[u4] < EL_NLV) {                  // the level test is always true for a valid index (depth < n_levels <= 16)
                const unsigned u = __float_as_uint(v);
                const unsigned long long key = ((unsigned long long)(u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u)) << 32) | (unsigned)(0x7fffffff - tp);
                unsigned long long *slot = &acc[lv4[u4]][tid];
                if (key > *slot) *slot = key;
            }
            const int te = te4[u4];
            if (te >= 0 && better(v, te, mine.v, mine.p)) { mine.v = v; mine.p = te; }
        }
    }
    // 256 slice maxima for the top-k threshold: best of each pair of consecutive threads
    {
        Best gmx = mine;
#pragma unroll
        for (int o = 1; o <= 1; o <<= 1) {
            const float ov = __shfl_xor(gmx.v, o);
            const int op = __shfl_xor(gmx.p, o);
            if (better(ov, op, gmx.v, gmx.p)) { gmx.v = ov; gmx.p = op; }
        }
        if ((tid & 1) == 0) { s_mv[tid >> 1] = gmx.v; s_mp[tid >> 1] = gmx.p; }
    }
    __syncthreads();
    for (int l = wave; l < EL_NLV; l += EL_NW) {              // wave w reduces levels w, w + 8: 512 keys each
        unsigned long long m = 0ull;
#pragma unroll
        for (int i = 0; i < EL_NT / 64; ++i) { const unsigned long long x = acc[l][lane + 64 * i]; m = x > m ? x : m; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned hi = __shfl_xor((unsigned)(m >> 32), o), lo = __shfl_xor((unsigned)m, o);
            const unsigned long long x = ((unsigned long long)hi << 32) | lo;
            m = x > m ? x : m;
        }
        if (lane == 0) s_key[l] = m;
    }
    __syncthreads();
    if (wave == 0) {                                        // lane l < n_levels finishes level l; the whole wave takes part in the shuffles
        const int l = lane;
        Best b = {-INFINITY, 0x7fffffff};
        if (l < n_levels) {
            const unsigned long long m = l < EL_NLV ? s_key[l] : 0ull;
            if (m) {
                const unsigned u = (unsigned)(m >> 32);
                b.v = __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
                b.p = 0x7fffffff - (int)(unsigned)m;
            }
            const int fo = filler_pos[l];
            const bool has_c = b.p < n_train, has_f = fo >= 0;
            int win;
            if (has_c && (!has_f || b.v > -1.0f || (b.v == -1.0f && b.p < fo))) win = b.p;
            else win = has_f ? fo : b.p;
            out_level[(int64_t)r * n_levels + l] = train_cols[win];
        }
        const Best top = wave_best(b);                      // unmasked top-1 = best of the level bests
        if (lane == 0 && out_top1) out_top1[r] = top.p < n_train ? train_cols[top.p] : -1;
    }
    if (k <= 0) return;
    // top-k over the test subset: threshold = k-th best of the 256 slice maxima, then sweep 2 collects the candidates
    if (tid < 256) {
        const float mv = s_mv[tid]; const int mp = s_mp[tid];
        int rank = 0, nonempty = 0;
        for (int j = 0; j < 256; ++j) { rank += better(s_mv[j], s_mp[j], mv, mp) ? 1 : 0; nonempty += s_mp[j] != 0x7fffffff ? 1 : 0; }
        // k distinct elements >= t are guaranteed only if at least k slices are non-empty (clustered / tiny test sets can
        // put several test columns into one slice); otherwise every test element is a candidate
        if (nonempty < k) { if (tid == 0) { s_t = -INFINITY; s_tp = 0x7fffffff; } }
        else if (rank == k - 1) { s_t = mv; s_tp = mp; }
    }
    __syncthreads();
    const float t = s_t; const int tp0 = s_tp;
    for (int c0 = tid; c0 < n_nodes; c0 += EL_U * EL_NT) {
        float v4[EL_U]; int te4[EL_U];
#pragma unroll
        for (int u4 = 0; u4 < EL_U; ++u4) {
            const int c = min(c0 + u4 * EL_NT, n_nodes - 1);
            v4[u4] = lr[c]; te4[u4] = test_pos[c];
        }
#pragma unroll
        for (int u4 = 0; u4 < EL_U; ++u4) {
            if (c0 + u4 * EL_NT >= n_nodes) break;
            const int te = te4[u4];
            const float v = v4[u4] + 0.0f;
            if (te >= 0 && !better(t, tp0, v, te)) {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < TOPK_CAND) { s_cv[slot] = v; s_cp[slot] = te; }
            }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt <= TOPK_CAND) {
        for (int c = tid; c < cnt; c += EL_NT) {
            const float v = s_cv[c]; const int p = s_cp[c];
            int rk = 0;
            for (int j = 0; j < cnt; ++j) rk += better(s_cv[j], s_cp[j], v, p) ? 1 : 0;
            if (rk < k) out_topk[(int64_t)r * k + rk] = test_cols[p];
        }
    } else {
        // heavily duplicated data: k rounds of block arg-max; an element is "removed" by requiring it to be worse than the
        // previous winner in (value, position) order, so nothing has to be written back
        Best last = {INFINITY, -1};
        for (int j = 0; j < k; ++j) {
            Best b = {-INFINITY, 0x7fffffff};
            for (int c = tid; c < n_nodes; c += EL_NT) {
                const int te = test_pos[c];
                if (te < 0) continue;
                const float v = lr[c] + 0.0f;
                if (better(last.v, last.p, v, te) && better(v, te, b.v, b.p)) { b.v = v; b.p = te; }
            }
            b = block_best8(b, s_v, s_p, tid);
            if (tid == 0) out_topk[(int64_t)r * k + j] = b.p < n_test ? test_cols[b.p] : -1;
            last = b;
            __syncthreads();
        }
    }
}


// ---- second stage of hgr_logits_eval (first stage: logits_slab in hgr_logits_slab.hip) ------------------------------------------
// One workgroup per image row.  Input: per 32-column slice of the level-sorted class matrix the best train key of this row and the
// largest test value of each of the slice's two 16-column groups (LE_GPS groups per slice).  (1) level arg-max = max of the keys of the level's slices,
// with the reference's -1 filler rule; unmasked top-1 = best of the level bests; (2) threshold t = k-th largest SLICE maximum over
// the test columns (k distinct elements >= t exist; fewer than k non-empty slices: t = -inf); (3) only a 16-column group whose
// maximum reaches t can hold one of the k best (typically k .. k + 5 groups of ~1 400): its 16 logits are recomputed on the matrix
// cores with the first stage's operand roles and K order (bit-identical values), elements >= t are ranked by (value, test
// position).  The [B, N] logits never exist in memory.
constexpr int LE_NT = 512, LE_NW = LE_NT / 64, LE_MAXS = 1024, LE_CAP = 2048, LE_GPS = 2;

template <int DT>
__global__ __launch_bounds__(LE_NT, 4) void logits_eval_rows(const void *__restrict__ feat, const void *__restrict__ zslp, int D, int S,
                                                        const unsigned long long *__restrict__ keys, const float *__restrict__ gmax,
                                                        const int *__restrict__ gp1, const float *__restrict__ gm2,
                                                        const int32_t *__restrict__ level_first, int n_levels, const int32_t *__restrict__ filler_pos,
                                                        const int32_t *__restrict__ train_cols, int n_train, const int32_t *__restrict__ epos,
                                                        const int32_t *__restrict__ test_cols, int n_test, int k,
                                                        int32_t *__restrict__ out_level, int32_t *__restrict__ out_top1, int32_t *__restrict__ out_topk, int dbg, int rows) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    // 58 KB of LDS: two rows per CU, so the 512 rows of a batch are one round on 256 CUs
    __shared__ unsigned long long s_lkey[32];
    __shared__ unsigned long long s_keys[LE_MAXS];
    __shared__ __attribute__((aligned(16))) float s_gm[LE_MAXS * LE_GPS];
    __shared__ __attribute__((aligned(16))) float s_m2[LE_MAXS * LE_GPS];
    __shared__ __attribute__((aligned(16))) int s_p1[LE_MAXS * LE_GPS];
    __shared__ float s_tm[LE_MAXS];
    __shared__ short s_cand[LE_MAXS * LE_GPS];
    __shared__ __attribute__((aligned(16))) E s_feat[1024];
    __shared__ float s_cv[LE_CAP];
    __shared__ int s_cp[LE_CAP];
    __shared__ float s_t;
    __shared__ int s_ncand, s_cnt, s_nonempty;
    __shared__ float s_bv[LE_NW];
    __shared__ int s_bp[LE_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    if (tid < 32) s_lkey[tid] = 0ull;
    if (tid == 0) { s_ncand = 0; s_cnt = 0; s_nonempty = 0; s_t = -INFINITY; }
    // ONE round trip to memory for everything the row needs: the first stage's four records of every slice (S <= 1024: at most two
    // slices per thread, all eight loads requested before the first use) and the image's feature row (the recompute's B operand).
    // Round 3 took five dependent trips (feature row | keys + maxima | ... | second values and positions of the groups above the
    // threshold | class rows of the recomputed groups): 18.5 us for 7 MB.
    {
        // record layout of the first stage: [slab = slice / 3][row][96 bytes: keys x 3 | (m1, m1') x 3 | (p1, p1') x 3 | (m2, m2') x 3]
        const int s0 = tid, s1 = tid + LE_NT;
        const int c0s = min(s0, S - 1), c1s = min(s1, S - 1);
        const char *r0 = (const char *)keys + ((int64_t)(c0s / 3) * rows + row) * 96 + (c0s % 3) * 8;
        const char *r1 = (const char *)keys + ((int64_t)(c1s / 3) * rows + row) * 96 + (c1s % 3) * 8;
        const unsigned long long k0 = *(const unsigned long long *)r0, k1 = *(const unsigned long long *)r1;
        const float2 a0 = *(const float2 *)(r0 + 24), a1 = *(const float2 *)(r1 + 24);
        const int2 c0 = *(const int2 *)(r0 + 48), c1 = *(const int2 *)(r1 + 48);
        const float2 b0 = *(const float2 *)(r0 + 72), b1 = *(const float2 *)(r1 + 72);
        u32x4 fv = (u32x4){0u, 0u, 0u, 0u};
        if (tid < (D >> 3)) fv = ((const u32x4 *)((const E *)feat + (int64_t)row * D))[tid];
        int mine = 0;
        if (s0 < S) {
            s_keys[s0] = k0; *(float2 *)(s_gm + s0 * LE_GPS) = a0; *(float2 *)(s_m2 + s0 * LE_GPS) = b0; *(int2 *)(s_p1 + s0 * LE_GPS) = c0;
            const float t = fmaxf(a0.x, a0.y);
            s_tm[s0] = t;
            mine += t > -INFINITY ? 1 : 0;
        }
        if (s1 < S) {
            s_keys[s1] = k1; *(float2 *)(s_gm + s1 * LE_GPS) = a1; *(float2 *)(s_m2 + s1 * LE_GPS) = b1; *(int2 *)(s_p1 + s1 * LE_GPS) = c1;
            const float t = fmaxf(a1.x, a1.y);
            s_tm[s1] = t;
            mine += t > -INFINITY ? 1 : 0;
        }
        if (tid < (D >> 3)) ((u32x4 *)s_feat)[tid] = fv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if (lane == 0 && mine) atomicAdd(&s_nonempty, mine);     // one LDS atomic per WAVE (same-address LDS atomics serialise)
    }
    __syncthreads();
    const bool want_k = k > 0 && !HGR_LAB_ON(dbg == 1);
    if (wave == 0 && want_k && s_nonempty >= k) {
        // threshold t: any value such that at least k DISTINCT test elements are >= t will do (a smaller t only lengthens the
        // candidate list).  Lane l holds the maxima of slices l, l + 64, ...; m = the largest of them.  The k-th largest of the 64
        // lane maxima is such a value (k lanes reach it, each with an element of its own) and costs one rank count per lane; round 3
        // ran a 32-step bisection over all slice maxima here (ballot + popcount per register and step: ~3.7 us of one wave).  With
        // fewer than k non-empty lanes (a tiny, clustered test set) the exact k-th largest slice maximum is bisected as before.
        unsigned kv[LE_MAXS / 64];
        const int nreg = (S + 63) >> 6;                     // registers that hold real slices (11 at N = 21 841): wave-uniform
        unsigned best = 0;
#pragma unroll
        for (int i = 0; i < LE_MAXS / 64; ++i) {
            const int s = i * 64 + lane;
            const unsigned u = __float_as_uint(s < S ? s_tm[s] : -INFINITY);
            kv[i] = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
            best = max(best, kv[i]);
        }
        const unsigned ninf = 0x007FFFFFu;                  // orderable key of -inf
        const int lanes_live = __popcll(__ballot(best > ninf));
        unsigned x = 0;
        if (lanes_live >= k) {
            int rk = 0;                                     // number of lanes whose maximum beats mine (ties: lower lane first)
            for (int j = 0; j < 64; ++j) {
                const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)best, j);
                rk += (o > best || (o == best && j < lane)) ? 1 : 0;
            }
            const unsigned long long pick = __ballot(rk == k - 1);
            x = (unsigned)__builtin_amdgcn_readlane((int)best, (int)__builtin_ctzll(pick));
        } else {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned c = x | (1u << bit);
                int cntc = 0;
#pragma unroll
                for (int i = 0; i < LE_MAXS / 64; ++i)
                    if (i < nreg) cntc += __popcll(__ballot(kv[i] >= c));
                if (cntc >= k) x = c;
            }
        }
        if (lane == 0) s_t = __uint_as_float(x ^ ((x >> 31) ? 0x80000000u : 0xFFFFFFFFu));
    } else if (wave > 0 || !want_k || s_nonempty < k) {
        // level bests: the slices of level l are the contiguous range [level_first[l], level_first[l + 1]); waves 1 .. 7 (all eight when
        // no threshold is wanted) reduce the levels from the staged keys while wave 0 finds the threshold
        const int w0 = (want_k && s_nonempty >= k) ? 1 : 0, nw = LE_NW - w0;
        for (int l = wave - w0; l < n_levels; l += nw) {
            unsigned long long m = 0ull;
            for (int s = level_first[l] + lane; s < level_first[l + 1]; s += 64) { const unsigned long long x = s_keys[s]; m = x > m ? x : m; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned hi = __shfl_xor((unsigned)(m >> 32), o), lo = __shfl_xor((unsigned)m, o);
                const unsigned long long x = ((unsigned long long)hi << 32) | lo;
                m = x > m ? x : m;
            }
            if (lane == 0) s_lkey[l] = m;
        }
    }
    __syncthreads();
    if (wave == LE_NW - 1) {                                // lane l < n_levels finishes level l (hgr_eval_rows' rule, main.py:162-176)
        const int l = lane;
        Best b = {-INFINITY, 0x7fffffff};
        if (l < n_levels) {
            const unsigned long long m = l < 32 ? s_lkey[l] : 0ull;
            if (m) {
                const unsigned u = (unsigned)(m >> 32);
                b.v = __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
                b.p = 0x7fffffff - (int)(unsigned)m;
            }
            const int fo = filler_pos[l];
            const bool has_c = b.p < n_train, has_f = fo >= 0;
            int win;
            if (has_c && (!has_f || b.v > -1.0f || (b.v == -1.0f && b.p < fo))) win = b.p;
            else win = has_f ? fo : b.p;
            out_level[(int64_t)row * n_levels + l] = train_cols[win];
        }
        const Best top = wave_best(b);
        if (lane == 0 && out_top1) out_top1[row] = top.p < n_train ? train_cols[top.p] : -1;
    }
    if (!want_k) return;
    if (HGR_LAB_ON(dbg == 2)) return;
    const float t = s_t;
    // a group whose maximum reaches t holds a candidate: its (value, position) are already known from the tile stage; only when its
    // SECOND largest value reaches t too (two of the best k in one 16-column group, or a tie at the maximum) the group is recomputed
    for (int q = tid; q < S * LE_GPS; q += LE_NT) {
        const float m1 = s_gm[q];
        if (m1 > -INFINITY && m1 >= t) {
            if (s_m2[q] >= t) s_cand[atomicAdd(&s_ncand, 1)] = (short)q;
            else {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < LE_CAP) { s_cv[slot] = m1; s_cp[slot] = s_p1[q]; }
            }
        }
    }
    __syncthreads();
    int ncand = s_ncand;
    if (HGR_LAB_ON(dbg == 3)) return;
    // visitor(value, test position) over every test element of the candidate groups; a wave takes TWO groups per trip (two
    // independent accumulators, 16 class-row fragments in flight).  Fragments: A operand = 16 class rows x 32 k (lane (r16, g): row
    // r16, k = 32 kk + 8 g ..), B operand = the image's feature for every one of its 16 "rows" m (from LDS, same address for all r16).
    const int r16 = lane & 15, g = lane >> 4;
    const int ks = D >> 5;
    auto scan = [&](auto visit) {
        for (int ci = wave * 2; ci < ncand; ci += 2 * LE_NW) {
            const int q0 = s_cand[ci], q1 = s_cand[min(ci + 1, ncand - 1)];
            const E *w0 = (const E *)zslp + ((int64_t)q0 * 16 + r16) * D + g * 8;
            const E *w1 = (const E *)zslp + ((int64_t)q1 * 16 + r16) * D + g * 8;
            f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
            for (int kk = 0; kk < ks; kk += 8) {                     // 8 k-steps per trip (D % 256 == 0 fast path), all 16 loads first
                vec8 f0[8], f1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kq = min(kk + u, ks - 1);               // D = 128 / 384 / 640: the tail repeats a step and skips its MFMA
                    f0[u] = *(const vec8 *)(w0 + kq * 32); f1[u] = *(const vec8 *)(w1 + kq * 32);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (kk + u < ks) {
                        const vec8 xf = *(const vec8 *)(s_feat + (kk + u) * 32 + g * 8);
                        a0 = T16<DT>::mfma16(f0[u], xf, a0);
                        a1 = T16<DT>::mfma16(f1[u], xf, a1);
                    }
                }
            }
            // lane (r16, g) holds columns 4 g .. 4 g + 3 of the group (identical for every r16): lanes r16 < 4 take element r16
            float v0 = r16 == 0 ? a0[0] : r16 == 1 ? a0[1] : r16 == 2 ? a0[2] : a0[3];
            float v1 = r16 == 0 ? a1[0] : r16 == 1 ? a1[1] : r16 == 2 ? a1[2] : a1[3];
            v0 += 0.0f; v1 += 0.0f;
            if (r16 < 4) {
                const int te0 = epos[q0 * 16 + g * 4 + r16];
                if (te0 >= 0) visit(v0, te0);
                if (ci + 1 < ncand) {
                    const int te1 = epos[q1 * 16 + g * 4 + r16];
                    if (te1 >= 0) visit(v1, te1);
                }
            }
        }
    };
    scan([&](float v, int te) {
        if (v >= t) {
            const int slot = atomicAdd(&s_cnt, 1);
            if (slot < LE_CAP) { s_cv[slot] = v; s_cp[slot] = te; }
        }
    });
    __syncthreads();
    if (HGR_LAB_ON(dbg == 4)) return;
    const int cnt = s_cnt;
    if (cnt <= LE_CAP) {
        for (int c = tid; c < cnt; c += LE_NT) {
            const float v = s_cv[c]; const int p = s_cp[c];
            int rk = 0;
            for (int j = 0; j < cnt; ++j) rk += better(s_cv[j], s_cp[j], v, p) ? 1 : 0;
            if (rk < k) out_topk[(int64_t)row * k + rk] = test_cols[p];
        }
        return;
    }
    // heavily duplicated data (more than LE_CAP elements tie at the threshold): every group that reaches t becomes a recompute
    // candidate, then k rounds of arg-max over them; an element is "removed" by requiring it to be worse than the previous winner
    // in (value, position) order
    __syncthreads();
    if (tid == 0) s_ncand = 0;
    __syncthreads();
    for (int q = tid; q < S * LE_GPS; q += LE_NT)
        if (s_gm[q] > -INFINITY && s_gm[q] >= t) s_cand[atomicAdd(&s_ncand, 1)] = (short)q;
    __syncthreads();
    ncand = s_ncand;
    Best last = {INFINITY, -1};
    for (int j = 0; j < k; ++j) {
        Best b = {-INFINITY, 0x7fffffff};
        scan([&](float v, int te) { if (better(last.v, last.p, v, te) && better(v, te, b.v, b.p)) { b.v = v; b.p = te; } });
        b = wave_best(b);
        __syncthreads();
        if (lane == 0) { s_bv[wave] = b.v; s_bp[wave] = b.p; }
        __syncthreads();
        Best w = {s_bv[0], s_bp[0]};
        for (int i = 1; i < LE_NW; ++i) if (better(s_bv[i], s_bp[i], w.v, w.p)) { w.v = s_bv[i]; w.p = s_bp[i]; }
        if (tid == 0) out_topk[(int64_t)row * k + j] = w.p < n_test ? test_cols[w.p] : -1;
        last = w;
    }
}

}  // namespace

int hgr_logits_eval_rows_launch(const void *feat, const void *zslp, int D, int S, const unsigned long long *keys, const float *tmax,
                                const int *gp1, const float *gm2, const int32_t *level_first, int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                                const int32_t *epos, const int32_t *test_cols, int n_test, int k, int32_t *out_level, int32_t *out_top1,
                                int32_t *out_topk, int rows, int dtype, void *stream) {
    static int dbg = -1;                              // HGR_LE_DBG = 1..4: leave the row stage after level / threshold / candidate list / scan (timing experiments only)
    if (dbg < 0) dbg = hgr_lab_env("HGR_LE_DBG");   // lab builds only
    if (dtype == HGR_BF16) hipLaunchKernelGGL((logits_eval_rows<HGR_BF16>), dim3(rows), dim3(LE_NT), 0, (hipStream_t)stream, feat, zslp, D, S, keys, tmax, gp1, gm2, level_first,
                                              n_levels, filler_pos, train_cols, n_train, epos, test_cols, n_test, k, out_level, out_top1, out_topk, dbg, rows);
    else hipLaunchKernelGGL((logits_eval_rows<HGR_F16>), dim3(rows), dim3(LE_NT), 0, (hipStream_t)stream, feat, zslp, D, S, keys, tmax, gp1, gm2, level_first,
                            n_levels, filler_pos, train_cols, n_train, epos, test_cols, n_test, k, out_level, out_top1, out_topk, dbg, rows);
    HGR_CHECK_LAUNCH("hgr_logits_eval (row stage)");
    return HGR_OK;
}

namespace {
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
                                const int32_t *depth, int n_levels, int32_t *out, int32_t *out_top1, int rows, void *stream) {
    HGR_REQUIRE(logits && depth && out, "hgr_level_argmax: null operand");
    HGR_REQUIRE(rows >= 1 && n_cols >= 1 && n_levels >= 1 && n_levels <= 32, "hgr_level_argmax: rows=%d n_cols=%d n_levels=%d unsupported (n_levels <= 32)", rows, n_cols, n_levels);
    hipStream_t s = (hipStream_t)stream;
    if (n_levels <= 16) hipLaunchKernelGGL((level_argmax<16>), dim3(rows), dim3(256), 0, s, logits, ld, cols, n_cols, depth, n_levels, out, out_top1);
    else hipLaunchKernelGGL((level_argmax<32>), dim3(rows), dim3(256), 0, s, logits, ld, cols, n_cols, depth, n_levels, out, out_top1);
    HGR_CHECK_LAUNCH("hgr_level_argmax");
    return HGR_OK;
}

namespace {
// T1 / T2 / T4 bookkeeping of one batch (main.py:139-148,157-160,177-191) from the outputs of eval_rows: one block, integer
// counts reduced in a fixed order, then the nine double-precision counters of the evaluation are advanced in place.
__global__ __launch_bounds__(1024) void eval_counters(const int32_t *__restrict__ pred, int k, const int64_t *__restrict__ targets, int target,
                                                      const int32_t *__restrict__ top1, const int32_t *__restrict__ lv, int n_levels,
                                                      const int32_t *__restrict__ parents, const int32_t *__restrict__ levels, int L,
                                                      double *__restrict__ acc, int B) {
    // One row per thread (batch 512 = one pass of the 1024-thread block) and every load of a row requested before the first
    // compare: the first version walked two rows per thread with its loads under the compares, a chain of dependent L2 round
    // trips that made this 0.1 MB kernel the 11 us tail of every evaluation step.
    __shared__ unsigned s_red[16][8];
    __shared__ int s_par[32], s_lev[32];
    // the target's path (ancestors, their levels) through LDS: read as uniform values inside the per-level loop they became a chain
    // of scalar loads, one L2 / HBM round trip per level (the arrays are views into a CSR nobody else touches: always cold)
    if (threadIdx.x < 32) {
        s_par[threadIdx.x] = threadIdx.x < L ? parents[threadIdx.x] : -1;
        s_lev[threadIdx.x] = threadIdx.x < L ? levels[threadIdx.x] : 0;
    }
    __syncthreads();
    unsigned cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // hits@1,2,5,10,20, hits_all, edges (or first-level matches when L == 1), points
    for (int r = threadIdx.x; r < B; r += 1024) {
        const int tgt = targets ? (int)targets[r] : target;
        int pr[32], lvv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) pr[i] = i < k ? pred[(int64_t)r * k + i] : -1;
#pragma unroll
        for (int i = 0; i < 32; ++i) lvv[i] = i < L ? lv[(int64_t)r * n_levels + s_lev[i]] : -1;
        const int t1 = top1[r];
        int j = k;
#pragma unroll
        for (int i = 31; i >= 0; --i)
            if (i < k && pr[i] == tgt) j = i;                   // first match (ids are distinct: at most one)
        cnt[0] += j < 1; cnt[1] += j < 2; cnt[2] += j < 5; cnt[3] += j < 10; cnt[4] += j < 20;
        bool prev = false;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int pa = s_par[i];
            const bool live = i < L;
            cnt[5] += live && t1 == pa;
            const bool m = live && lvv[i] == pa;
            cnt[7] += m;
            if (L == 1) cnt[6] += m;
            else if (i > 0) cnt[6] += (m && prev);
            prev = m;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsigned v = cnt[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t[8];
        for (int c = 0; c < 8; ++c) {
            t[c] = 0;
            for (int w = 0; w < 16; ++w) t[c] += s_red[w][c];
        }
        for (int c = 0; c < 6; ++c) acc[c] += (double)t[c];
        acc[6] += L > 1 ? (double)t[6] / (double)(L - 1) : (double)t[6];
        acc[7] += (double)t[7] / (double)L;
        acc[8] += (double)B;
    }
}
}  // namespace

extern "C" int hgr_eval_counters(const int32_t *pred, int k, const int64_t *targets, int target, const int32_t *top1, const int32_t *lv,
                                 int n_levels, const int32_t *parents, const int32_t *levels, int L, double *acc, int rows, void *stream) {
    HGR_REQUIRE(pred && top1 && lv && parents && levels && acc, "hgr_eval_counters: null operand");
    HGR_REQUIRE(rows >= 1 && k >= 1 && k <= 32 && L >= 1 && L <= 32 && n_levels >= 1, "hgr_eval_counters: bad sizes rows=%d k=%d L=%d (k, L <= 32)", rows, k, L);
    hipLaunchKernelGGL(eval_counters, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, k, targets, target, top1, lv, n_levels, parents, levels, L, acc, rows);
    HGR_CHECK_LAUNCH("hgr_eval_counters");
    return HGR_OK;
}

extern "C" int hgr_eval_rows(const float *logits, int64_t ld, int n_nodes, const unsigned char *lvl8, const int32_t *train_pos,
                             const int32_t *train_cols, int n_train, int n_levels, const int32_t *filler_pos, const int32_t *test_pos,
                             const int32_t *test_cols, int n_test, int k, int32_t *out_level, int32_t *out_top1, int32_t *out_topk,
                             int rows, void *stream) {
    HGR_REQUIRE(logits && lvl8 && train_pos && train_cols && filler_pos && out_level, "hgr_eval_rows: null operand");
    HGR_REQUIRE(rows >= 1 && n_nodes >= 1 && ld >= n_nodes && n_train >= 1 && n_levels >= 1 && n_levels <= 32, "hgr_eval_rows: bad sizes (n_levels <= 32)");
    HGR_REQUIRE(k == 0 || (out_topk && test_pos && test_cols && k >= 1 && k <= 32 && n_test >= k), "hgr_eval_rows: bad top-k arguments");
    static const bool regs = getenv("HGR_EVAL_REGS") != nullptr;      // diagnostics: the register-accumulator variant
    if (n_levels <= 16 && !regs) hipLaunchKernelGGL(eval_rows_lds, dim3(rows), dim3(EL_NT), 0, (hipStream_t)stream, logits, ld, n_nodes, lvl8, train_pos, train_cols,
                                                    n_train, n_levels, filler_pos, test_pos, test_cols, n_test, k, out_level, out_top1, out_topk);
    else if (n_levels <= 16) hipLaunchKernelGGL((eval_rows<16>), dim3(rows), dim3(EV_NT), 0, (hipStream_t)stream, logits, ld, n_nodes, lvl8, train_pos, train_cols, n_train,
                                           n_levels, filler_pos, test_pos, test_cols, n_test, k, out_level, out_top1, out_topk);
    else hipLaunchKernelGGL((eval_rows<32>), dim3(rows), dim3(EV_NT), 0, (hipStream_t)stream, logits, ld, n_nodes, lvl8, train_pos, train_cols, n_train,
                            n_levels, filler_pos, test_pos, test_cols, n_test, k, out_level, out_top1, out_topk);
    HGR_CHECK_LAUNCH("hgr_eval_rows");
    return HGR_OK;
}
