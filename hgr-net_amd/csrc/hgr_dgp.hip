// DGP baseline graph propagation: the attention-weighted ancestor / descendant aggregation of
// baseline/DGP/models/gcn_dense_att.py:31-46,
//
//     out = act( sum_d softmax(att)_d * (A_d . support) ),      support = x W + b,
//
// with A_d the in-degree-normalised adjacency of the node pairs at graph distance d (baseline/DGP/utils.py:56-65).
// All D operators are merged into ONE CSR whose edges carry their group id and 1/degree, so a row is a single
// gather-reduce   out[i] = sum_e att[grp_e] * inv_deg_e * support[col_e]   (+ bias * sum_e w_e, folding the "+ b").
//
// HBM-bound gather: a workgroup owns one work item = a run of at most CHUNK edges of one row; its 256 threads each own
// a float4 column slice (a 4 KB coalesced read per neighbour row and pass), 4 neighbour rows in flight per thread.
// Descendant-side operators are extremely skewed (the root aggregates every node), so long rows are cut into several
// items: a row with one item is finished in place (LeakyReLU, row L2 norm via wave shuffles + LDS), a row with several
// writes fp32 partials that `csr_finish` adds in item order - deterministic, no atomics.
#include "hgr_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// finish a row held as float4 slices in registers: activation, optional L2 normalisation, store
template <int NV>
__device__ __forceinline__ void finish_row(f32x4 (&acc)[NV], float *__restrict__ dst, int C, float slope, int normalize, float *red) {
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[v][e];
            x = x >= 0.f ? x : x * slope;
            acc[v][e] = x;
            ss += x * x;
        }
    }
    float scale = 1.f;
    if (normalize) {                                   // F.normalize: x / max(|x|, 1e-12); padding lanes hold zeros
        const float tot = block_sum(ss, red);
        scale = 1.f / fmaxf(sqrtf(tot), 1e-12f);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * NT + threadIdx.x) * 4;
        if (c < C) *(f32x4 *)(dst + c) = acc[v] * scale;
    }
}

template <int NV>
__global__ __launch_bounds__(NT) void csr_aggregate(const float *__restrict__ support, int64_t lds_, const int *__restrict__ item_row,
                                                    const int *__restrict__ item_e0, const int *__restrict__ item_e1,
                                                    const int *__restrict__ item_slot, const int *__restrict__ col,
                                                    const float *__restrict__ inv_deg, const unsigned char *__restrict__ grp,
                                                    const float *__restrict__ att, int D, const float *__restrict__ bias,
                                                    float *__restrict__ out, int64_t ldo, float *__restrict__ partial, int C,
                                                    float slope, int normalize) {
    __shared__ float s_att[32];
    __shared__ float red[4];
    if (threadIdx.x < D) s_att[threadIdx.x] = att[threadIdx.x];
    __syncthreads();
    const int it = blockIdx.x;
    const int row = item_row[it], e0 = item_e0[it], e1 = item_e1[it], slot = item_slot[it];
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
    int e = e0;
    for (; e + 4 <= e1; e += 4) {                      // edge data is block-uniform: scalar loads
        const int j0 = col[e], j1 = col[e + 1], j2 = col[e + 2], j3 = col[e + 3];
        const float w0 = s_att[grp[e]] * inv_deg[e], w1 = s_att[grp[e + 1]] * inv_deg[e + 1];
        const float w2 = s_att[grp[e + 2]] * inv_deg[e + 2], w3 = s_att[grp[e + 3]] * inv_deg[e + 3];
        wsum += (w0 + w1) + (w2 + w3);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * NT + threadIdx.x) * 4;
            if (c < C) {
                const f32x4 s0 = *(const f32x4 *)(support + (int64_t)j0 * lds_ + c), s1 = *(const f32x4 *)(support + (int64_t)j1 * lds_ + c);
                const f32x4 s2 = *(const f32x4 *)(support + (int64_t)j2 * lds_ + c), s3 = *(const f32x4 *)(support + (int64_t)j3 * lds_ + c);
                acc[v] += (s0 * w0 + s1 * w1) + (s2 * w2 + s3 * w3);
            }
        }
    }
    for (; e < e1; ++e) {
        const int j = col[e];
        const float w = s_att[grp[e]] * inv_deg[e];
        wsum += w;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * NT + threadIdx.x) * 4;
            if (c < C) acc[v] += *(const f32x4 *)(support + (int64_t)j * lds_ + c) * w;
        }
    }
    if (bias) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * NT + threadIdx.x) * 4;
            if (c < C) acc[v] += *(const f32x4 *)(bias + c) * wsum;
        }
    }
    if (slot < 0) { finish_row<NV>(acc, out + (int64_t)row * ldo, C, slope, normalize, red); return; }
    float *p = partial + (int64_t)slot * C;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * NT + threadIdx.x) * 4;
        if (c < C) *(f32x4 *)(p + c) = acc[v];
    }
}

// rows cut into several items: out[row] = act(sum of its partials, in item order)
template <int NV>
__global__ __launch_bounds__(NT) void csr_finish(const int *__restrict__ split_row, const int *__restrict__ split_slot0,
                                                 const int *__restrict__ split_n, const float *__restrict__ partial,
                                                 float *__restrict__ out, int64_t ldo, int C, float slope, int normalize) {
    __shared__ float red[4];
    const int row = split_row[blockIdx.x], s0 = split_slot0[blockIdx.x], ns = split_n[blockIdx.x];
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ns; ++s) {
        const float *p = partial + (int64_t)(s0 + s) * C;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * NT + threadIdx.x) * 4;
            if (c < C) acc[v] += *(const f32x4 *)(p + c);
        }
    }
    finish_row<NV>(acc, out + (int64_t)row * ldo, C, slope, normalize, red);
}

}  // namespace

extern "C" int hgr_csr_group_aggregate(const float *support, int64_t ld_support, const int *item_row, const int *item_e0,
                                       const int *item_e1, const int *item_slot, int n_items, const int *col, const float *inv_deg,
                                       const unsigned char *grp, const float *att, int D, const float *bias,
                                       const int *split_row, const int *split_slot0, const int *split_n, int n_split,
                                       float *partial, float *out, int64_t ld_out, int C, float slope, int normalize, void *stream) {
    HGR_REQUIRE(support && item_row && item_e0 && item_e1 && item_slot && col && inv_deg && grp && att && out, "hgr_csr_group_aggregate: null operand");
    HGR_REQUIRE(n_items >= 1 && D >= 1 && D <= 32, "hgr_csr_group_aggregate: bad sizes n_items=%d D=%d (D <= 32)", n_items, D);
    HGR_REQUIRE(C >= 4 && C % 4 == 0 && C <= 4 * NT * 4, "hgr_csr_group_aggregate: C=%d must be a multiple of 4, at most %d", C, 4 * NT * 4);
    HGR_REQUIRE(ld_support % 4 == 0 && ld_out % 4 == 0 && ld_support >= C && ld_out >= C, "hgr_csr_group_aggregate: leading dimensions must be multiples of 4 and >= C");
    HGR_REQUIRE(hgr_aligned(support, 16) && hgr_aligned(out, 16) && (!bias || hgr_aligned(bias, 16)) && (!partial || hgr_aligned(partial, 16)),
                "hgr_csr_group_aggregate: support / out / bias / partial must be 16-byte aligned");
    HGR_REQUIRE(n_split == 0 || (split_row && split_slot0 && split_n && partial), "hgr_csr_group_aggregate: split rows need their tables and the partial buffer");
    hipStream_t s = (hipStream_t)stream;
    const int nv = (C + 4 * NT - 1) / (4 * NT);
#define HGR_DGP_LAUNCH(NV)                                                                                                          \
    do {                                                                                                                            \
        hipLaunchKernelGGL(csr_aggregate<NV>, dim3(n_items), dim3(NT), 0, s, support, ld_support, item_row, item_e0, item_e1,       \
                           item_slot, col, inv_deg, grp, att, D, bias, out, ld_out, partial, C, slope, normalize);                  \
        if (n_split > 0)                                                                                                            \
            hipLaunchKernelGGL(csr_finish<NV>, dim3(n_split), dim3(NT), 0, s, split_row, split_slot0, split_n, partial, out, ld_out, \
                               C, slope, normalize);                                                                                \
    } while (0)
    if (nv == 1) HGR_DGP_LAUNCH(1);
    else if (nv == 2) HGR_DGP_LAUNCH(2);
    else if (nv == 3) HGR_DGP_LAUNCH(3);
    else HGR_DGP_LAUNCH(4);
#undef HGR_DGP_LAUNCH
    HGR_CHECK_LAUNCH("hgr_csr_group_aggregate");
    return HGR_OK;
}
