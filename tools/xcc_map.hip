// Dev tool: which XCD does workgroup b of a 1-D grid land on?  (s_getreg_b32 HW_REG_XCC_ID, low 4 bits.)  Launched with the
// shapes libhgr's persistent kernels use: 256 workgroups, 384 / 512 threads, 120 KB of LDS (one workgroup per CU).
// build: hipcc --offload-arch=gfx950 -O3 tools/xcc_map.hip -o tools/_ab/xcc_map
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDSB>
__global__ void who(int *out) {
    __shared__ char smem[LDSB];
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    smem[threadIdx.x] = (char)v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 15) + (smem[1] == 77 ? 100 : 0);
}
int main() {
    int *d; hipMalloc(&d, 4096 * 4); static int h[4096];
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int grid = cfg == 2 ? 1024 : 256, threads = cfg == 0 ? 384 : 512;
        hipMemset(d, 0xff, 4096 * 4);
        if (cfg == 2) hipLaunchKernelGGL((who<1024>), dim3(grid), dim3(threads), 0, 0, d);
        else hipLaunchKernelGGL((who<120 * 1024>), dim3(grid), dim3(threads), 0, 0, d);
        hipMemcpy(h, d, grid * 4, hipMemcpyDeviceToHost);
        int ok = 0; for (int b = 0; b < grid; ++b) ok += (h[b] == (b & 7));
        printf("cfg %d grid %d threads %d: xcc(b) == b %% 8 for %d of %d blocks; first 32:", cfg, grid, threads, ok, grid);
        for (int b = 0; b < 32; ++b) printf(" %d", h[b]);
        printf("\n");
    }
    return 0;
}
