// Which XCD does workgroup blockIdx land on?  Prints the XCC_ID hardware register per block for a few grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xF);
}
int main() {
    int *d; hipMalloc(&d, 4096 * sizeof(int));
    for (int grid : {64, 560, 1024}) {
        hipMemset(d, 0xFF, 4096 * sizeof(int)); hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d); hipDeviceSynchronize();
        int h[4096]; hipMemcpy(h, d, grid * sizeof(int), hipMemcpyDeviceToHost);
        int mism = 0; for (int i = 0; i < grid; ++i) mism += (h[i] != (i & 7));
        printf("grid %d: xcc of blocks 0..31:", grid); for (int i = 0; i < 32 && i < grid; ++i) printf(" %d", h[i]);
        printf("  | blocks with xcc != blockIdx %% 8: %d of %d\n", mism, grid);
    }
    return 0;
}
