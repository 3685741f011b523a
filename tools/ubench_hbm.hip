// ubench_hbm.hip -- achievable HBM rates on the box for the access mixes the CLV kernel produces
// (read-only, write-only, 1:1 copy, 2:1 read:write), 16 B per lane, 4 loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dvec2 __attribute__((ext_vector_type(2)));
template <int MODE>   // 0 read, 1 write, 2 copy, 3 read2+write1
__global__ __launch_bounds__(256) void k(const dvec2 *__restrict__ a, const dvec2 *__restrict__ a2, dvec2 *__restrict__ b, size_t n, double *sink) {
    dvec2 acc = {0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        dvec2 v0, v1, v2, v3;
        if (MODE != 1) { v0 = a[i]; v1 = a[i + stride]; v2 = a[i + 2 * stride]; v3 = a[i + 3 * stride]; }
        else { v0 = v1 = v2 = v3 = (dvec2){1.0, 2.0}; }
        if (MODE == 3) { v0 += a2[i]; v1 += a2[i + stride]; v2 += a2[i + 2 * stride]; v3 += a2[i + 3 * stride]; }
        if (MODE == 0) acc += v0 + v1 + v2 + v3;
        else { b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3; }
    }
    if (MODE == 0 && acc.x == 123.456) sink[0] = acc.y;
}
template <int MODE>
static void run(const char *name, double bytes_per_elem, int blocks) {
    const size_t n = (size_t)1 << 27;   // 2 GiB per buffer
    static dvec2 *a = nullptr, *a2, *b; static double *sink;
    if (!a) { (void)hipMalloc(&a, n * 16); (void)hipMalloc(&a2, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&sink, 8);
              (void)hipMemset(a, 0, n * 16); (void)hipMemset(a2, 0, n * 16); (void)hipMemset(b, 0, n * 16); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(a, a2, b, n, sink);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<blocks, 256>>>(a, a2, b, n, sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s blocks=%5d  %.2f TB/s\n", name, blocks, 5.0 * n * bytes_per_elem / (ms * 1e-3) / 1e12);
}
int main() {
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0>("read-only", 16, blocks);
        run<1>("write-only", 16, blocks);
        run<2>("copy (1R:1W)", 32, blocks);
        run<3>("2R:1W", 48, blocks);
    }
    return 0;
}
