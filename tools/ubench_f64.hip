// ubench_f64.hip -- instruction-rate microbenchmark used to choose the contraction instruction
// (DESIGN.md "MFMA vs VALU for the 20x20 contraction").  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        } else if (KIND == 3) {          // ONE dependent chain (4 instructions per iteration, each waits for the last)
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
        } else if (KIND == 4) {          // TWO chains interleaved (k_oplist's streamed second-side contraction)
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
        } else {
            s0 = __builtin_fma(a, b, s0); s1 = __builtin_fma(a, b, s1);
            s2 = __builtin_fma(a, b, s2); s3 = __builtin_fma(a, b, s3);
            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double r = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int KIND>
static void run(const char *name, double flop_per_inst, int waves_per_simd) {
    double *d; hipMalloc(&d, 8 * 256 * 4096);
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 1 wave per SIMD per block
    k<KIND><<<blocks, 256>>>(d, 100);
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    const double ninst = (double)iters * 4;
    printf("%-22s waves/SIMD=%d  cycles/inst/wave=%.2f  chip=%.2f TFLOP/s\n", name, waves_per_simd, cyc / ninst,
           flop_per_inst * ninst * blocks * 4 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
__global__ void copyk(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("mfma_f64_16x16x4", 2048, w);
        run<1>("mfma_f64_4x4x4_4b", 512, w);
        run<2>("v_fma_f64", 128, w);
        run<3>("mfma_4x4x4 1 chain", 512, w);
        run<4>("mfma_4x4x4 2 chains", 512, w);
    }
    size_t n = (size_t)1 << 27;   // 2 GiB each
    double2 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    copyk<<<2048, 256>>>(a, b, n);
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) copyk<<<2048, 256>>>(a, b, n); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("copy 16B/lane: %.2f TB/s (read+write)\n", 5.0 * 2 * n * 16 / (ms * 1e-3) / 1e12);
    return 0;
}
