// Forward-progress probe for ticket-claimed gangs: G workgroups (a "gang") wait for each other (spin on arrival counters);
// gangs are formed from consecutive TICKETS (taken when a workgroup starts), optionally one ticket counter per blockIdx & 7.
// More workgroups than slots, gang size not dividing the slots: does the launch drain?  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Ctl { int ticket[8]; int arrived[4096]; int aborted; long long maxwait; };
__global__ __launch_bounds__(256, 2) void k(Ctl *c, int gang, int per_xcd, long long timeout, int work_iters) {
    extern __shared__ double lds[];
    __shared__ int s_t;
    const int part = per_xcd ? (blockIdx.x & 7) : 0;
    if (threadIdx.x == 0) s_t = __hip_atomic_fetch_add(&c->ticket[part], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = s_t, g = part * 512 + t / gang;
    double x = threadIdx.x;                       // some work before the rendezvous
    // uneven durations (as in the real kernel): gang members differ by up to 2x
    work_iters += (work_iters / 64) * ((t * 37) & 63);
    for (int i = 0; i < work_iters; ++i) x = x * 1.0000001 + 1e-9;
#ifdef USE_SCRATCH
    volatile double sc[24];                       // private segment (scratch), like a kernel with register spills
    for (int i = 0; i < 24; ++i) sc[i] = x + i;
    x += sc[(threadIdx.x + t) % 24];
#endif
    lds[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&c->arrived[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        long long w = 0;
        while (__hip_atomic_load(&c->arrived[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gang) {
            __builtin_amdgcn_s_sleep(8);
            w = wall_clock64() - t0;
            if (w > timeout || __hip_atomic_load(&c->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(&c->aborted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        atomicMax((unsigned long long *)&c->maxwait, (unsigned long long)w);
    }
    __syncthreads();
}
int main(int argc, char **argv) {
    const int gang = argc > 1 ? atoi(argv[1]) : 35, ngangs_per_part = argc > 2 ? atoi(argv[2]) : 2, per_xcd = argc > 3 ? atoi(argv[3]) : 1;
    const int lds = argc > 4 ? atoi(argv[4]) : 77312, work = argc > 5 ? atoi(argv[5]) : 20000;
    Ctl *c; hipMalloc(&c, sizeof(Ctl)); hipMemset(c, 0, sizeof(Ctl)); hipDeviceSynchronize();
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = per_xcd ? 8 * gang * ngangs_per_part : gang * ngangs_per_part;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, c, gang, per_xcd, 50000000LL /* 0.5 s */, work); hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    Ctl h; hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost);
    printf("gang %d x %d per partition, per_xcd %d, lds %d: grid %d, %.3f ms, aborted %d, longest wait %.1f us\n", gang, ngangs_per_part, per_xcd, lds, grid, ms, h.aborted, h.maxwait / 100.0);
    return 0;
}
