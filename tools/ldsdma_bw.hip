// Dev tool (standalone, no torch): per-CU operand-staging bandwidth on gfx950 from an L2-resident buffer.
//   mode 0: LDS-DMA  global_load_lds_dwordx4 (16 B per lane straight into LDS; what every GEMM of libhgr stages with)
//   mode 1: register path  global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: register path without the LDS write (global_load_dwordx4 only)
// NOTE: the mode-1 numbers are not trustworthy for the large region (hipcc hoists the pass-invariant loads out of the pass
// loop there: 'bandwidths' of hundreds of GB/s per CU); modes 0 and 2 are the ones quoted in profiles/NOTES.md section 4.1b.
// Each wave keeps DEPTH loads of 1 KB in flight (counted vmcnt) and walks its workgroup's private, L2-hot region
// (`region` bytes, re-read `passes` times).  pattern 0: an instruction reads 1 KB contiguous; pattern 1: 8 rows of
// 128 B, 1 KB apart (one K-tile slice of 8 operand rows with K = 512 - the GEMM loaders' pattern).
// build: hipcc --offload-arch=gfx950 -O3 tools/ldsdma_bw.hip -o tools/_ab/ldsdma_bw ; run: ldsdma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int DEPTH, int PATTERN>
__global__ __launch_bounds__(512) void stage_bw(const char *buf, size_t region, int passes, unsigned *sink) {
    __shared__ __attribute__((aligned(1024))) char smem[8 * 16 * 1024];        // 16 slots of 1 KB per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const char *base = buf + (size_t)blockIdx.x * region;
    const int per_pass = (int)(region / 1024);                                 // 1 KB instructions per pass of the workgroup
    char *my = smem + wave * 16384;
    // lane offset inside an 8 KB group: pattern 0 = instruction i reads bytes [i KB, i KB + 1 KB); pattern 1 = rows of 1 KB, slice i
    const int loff0 = lane * 16;
    const int loff1 = (lane >> 3) * 1024 + (lane & 7) * 16;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[DEPTH];
    int issued = 0;
    for (int p = 0; p < passes; ++p) {
        for (int i = wave; i < per_pass; i += nwaves * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int ii = i + d * nwaves;
                const int iw = ii < per_pass ? ii : wave;
                const char *src = PATTERN == 0 ? base + (size_t)iw * 1024 + loff0
                                               : base + (size_t)(iw >> 3) * 8192 + (iw & 7) * 128 + loff1;
                if (MODE == 0) __builtin_amdgcn_global_load_lds((const AS1 void *)src, (AS3 void *)(my + d * 1024), 16, 0, 0);
                else v[d] = *(const u32x4 *)src;
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if (MODE == 1) *(u32x4 *)(my + d * 1024 + lane * 16) = v[d];
                    else acc ^= v[d];
                }
            }
            ++issued;
        }
    }
    if (MODE == 0 || MODE == 1) { __syncthreads(); acc = *(const u32x4 *)(my + lane * 16); }
    if (acc[0] == 0x12345678u && issued == -1) sink[0] = acc[1];
}

template <int MODE, int DEPTH, int PATTERN>
static void run(const char *name, const char *buf, unsigned *sink, int grid, int threads, size_t region, int passes) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(s, 0);
        hipLaunchKernelGGL((stage_bw<MODE, DEPTH, PATTERN>), dim3(grid), dim3(threads), 0, 0, buf, region, passes, sink);
        hipEventRecord(e, 0); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    const double bytes = (double)grid * region * passes;
    printf("{\"kind\": \"%s\", \"depth\": %d, \"pattern\": %d, \"grid\": %d, \"waves\": %d, \"us\": %.1f, \"GBps_per_cu\": %.1f, \"TBps_total\": %.2f}\n",
           name, DEPTH, PATTERN, grid, threads / 64, best * 1e3, bytes / grid / best / 1e6, bytes / best / 1e9);
    fflush(stdout);
}

int main() {
    char *buf; unsigned *sink;
    hipMalloc(&buf, 256 * (size_t)(512 * 1024)); hipMemset(buf, 1, 256 * (size_t)(512 * 1024)); hipMalloc(&sink, 64);
    // 64 KB per workgroup: 32 workgroups of an XCD share 2 MB of its 4 MB L2 (L2-hot); 512 KB: 16 MB per XCD (MALL / HBM)
    for (size_t region : {(size_t)64 * 1024, (size_t)512 * 1024})
    for (int grid : {1, 256}) {
        const int passes = (int)((16u << 20) / region);     // 16 MB staged per workgroup
        printf("# region %zu KB per workgroup, grid %d\n", region / 1024, grid);
        for (int threads : {256, 512}) {
            run<0, 4, 0>("lds_dma", buf, sink, grid, threads, region, passes);
            run<0, 8, 0>("lds_dma", buf, sink, grid, threads, region, passes);
            run<0, 16, 0>("lds_dma", buf, sink, grid, threads, region, passes);
            run<0, 8, 1>("lds_dma", buf, sink, grid, threads, region, passes);
            run<1, 8, 0>("reg+ds_write", buf, sink, grid, threads, region, passes);
            run<1, 16, 0>("reg+ds_write", buf, sink, grid, threads, region, passes);
            run<1, 8, 1>("reg+ds_write", buf, sink, grid, threads, region, passes);
            run<2, 8, 0>("reg_only", buf, sink, grid, threads, region, passes);
            run<2, 16, 0>("reg_only", buf, sink, grid, threads, region, passes);
        }
    }
    return 0;
}
