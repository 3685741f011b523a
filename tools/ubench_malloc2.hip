// hipMalloc of a C4-shard-sized arena (111 GiB): one block against segments of 16 / 28 GiB, with a full first touch (memset of every
// byte) and the release: does segmenting the arena make the cold first call cheaper?   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_malloc2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    const double total = 111.0;
    for (double seg : {111.0, 28.0, 16.0, 111.0, 28.0}) {
        const int n = (int)(total / seg + 0.999); std::vector<void *> ps(n, nullptr);
        const size_t b = (size_t)(seg * (1ull << 30));
        double t0 = now(); for (auto &p : ps) if (hipMalloc(&p, b) != hipSuccess) printf("malloc failed\n"); double t1 = now();
        for (auto p : ps) hipMemsetAsync(p, 0, b, 0); hipDeviceSynchronize(); double t2 = now();
        for (auto p : ps) hipMemsetAsync(p, 1, b, 0); hipDeviceSynchronize(); double t3 = now();
        for (auto p : ps) hipFree(p); double t4 = now();
        printf("%3d segment(s) of %5.1f GiB: malloc %8.1f ms, first touch of every byte %8.1f ms, second pass %7.1f ms, free %8.1f ms\n", n, seg, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    return 0;
}
