// ubench_hbm.hip -- achievable HBM rates on the box for the access mixes the CLV kernel produces (read-only, write-only,
// 1:1 copy, 2:1 read:write), 16 B per lane.  Round 1's version (grid-stride, 4 loads in flight per lane, plain accesses)
// topped out at 4.85 TB/s for a copy while MI355X_MICROARCH.md measures 6.29 TB/s; this one sweeps what differs between
// the two: launch shape (persistent grid-stride / one pass per workgroup with a contiguous tile), loads in flight per
// lane (4 / 8), non-temporal loads and stores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dvec2 __attribute__((ext_vector_type(2)));
template <int NT> __device__ __forceinline__ dvec2 ld(const dvec2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT> __device__ __forceinline__ void st(dvec2 *p, dvec2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0 read, 1 write, 2 copy, 3 read2+write1;  TILE: 0 = grid-stride (loads `stride` apart), 1 = each workgroup owns
// contiguous tiles of 256 * U elements;  U loads in flight per lane
template <int MODE, int TILE, int U, int NT>
__global__ __launch_bounds__(256) void k(const dvec2 *__restrict__ a, const dvec2 *__restrict__ a2, dvec2 *__restrict__ b, size_t n, double *sink) {
    dvec2 acc = {0, 0};
    const size_t step = TILE ? (size_t)256 : (size_t)gridDim.x * 256;
    const size_t chunk = TILE ? (size_t)256 * U : (size_t)gridDim.x * 256 * U;
    for (size_t base = TILE ? (size_t)blockIdx.x * 256 * U + threadIdx.x : (size_t)blockIdx.x * 256 + threadIdx.x; base + (U - 1) * step < n;
         base += TILE ? (size_t)gridDim.x * 256 * U : chunk) {
        dvec2 v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) v[i] = (MODE != 1) ? ld<NT>(a + base + i * step) : (dvec2){1.0, 2.0};
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < U; ++i) v[i] += ld<NT>(a2 + base + i * step);
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < U; ++i) acc += v[i];
        } else {
#pragma unroll
            for (int i = 0; i < U; ++i) st<NT>(b + base + i * step, v[i]);
        }
    }
    if (MODE == 0 && acc.x == 123.456) sink[0] = acc.y;
}
static dvec2 *A = nullptr, *A2, *B; static double *sink;
template <int MODE, int TILE, int U, int NT>
static double run(int blocks) {
    const size_t n = (size_t)1 << 27;   // 2 GiB per buffer
    if (!A) { (void)hipMalloc(&A, n * 16); (void)hipMalloc(&A2, n * 16); (void)hipMalloc(&B, n * 16); (void)hipMalloc(&sink, 8);
              (void)hipMemset(A, 0, n * 16); (void)hipMemset(A2, 0, n * 16); (void)hipMemset(B, 0, n * 16); }
    if (blocks == 0) blocks = (int)(n / (256 * U));          // one tile per workgroup
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, TILE, U, NT><<<blocks, 256>>>(A, A2, B, n, sink);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE, TILE, U, NT><<<blocks, 256>>>(A, A2, B, n, sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bpe = MODE == 0 ? 16 : (MODE == 1 ? 16 : (MODE == 2 ? 32 : 48));
    return 5.0 * n * bpe / (ms * 1e-3) / 1e12;
}
template <int MODE> static void sweep(const char *name) {
    printf("%-12s grid-stride 2048 wg: U4 %.2f  U8 %.2f  U4 nt %.2f | tiles, 2048 wg: U4 %.2f  U8 %.2f  U8 nt %.2f | one tile per wg: U4 %.2f  U8 %.2f  U8 nt %.2f  TB/s\n", name,
           run<MODE, 0, 4, 0>(2048), run<MODE, 0, 8, 0>(2048), run<MODE, 0, 4, 1>(2048),
           run<MODE, 1, 4, 0>(2048), run<MODE, 1, 8, 0>(2048), run<MODE, 1, 8, 1>(2048),
           run<MODE, 1, 4, 0>(0), run<MODE, 1, 8, 0>(0), run<MODE, 1, 8, 1>(0));
}
int main() {
    sweep<0>("read-only"); sweep<1>("write-only"); sweep<2>("copy 1R:1W"); sweep<3>("2R:1W");
    return 0;
}
