// hipMalloc cost vs size (the driver zero-fills fresh VRAM): one big block vs many chunks
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    for (double gib : {1.0, 2.0, 4.0, 8.0, 12.8, 32.0, 111.0}) {
        size_t b = (size_t)(gib * (1ull << 30)); void *p = nullptr;
        double t0 = now(); hipError_t e = hipMalloc(&p, b); double t1 = now();
        hipMemset(p, 0, 256); hipDeviceSynchronize(); double t2 = now();
        hipFree(p); double t3 = now();
        printf("one block %6.1f GiB: malloc %8.1f ms (%s)  first touch %6.1f ms  free %7.1f ms\n", gib, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2);
    }
    for (double chunk : {1.0, 2.0, 4.0}) {
        const int n = (int)(32.0 / chunk); std::vector<void *> ps(n);
        double t0 = now(); for (auto &p : ps) hipMalloc(&p, (size_t)(chunk * (1ull << 30))); double t1 = now();
        for (auto p : ps) hipFree(p);
        printf("32 GiB as %d chunks of %.0f GiB: %8.1f ms\n", n, chunk, t1 - t0);
    }
    // second time around (driver caches?)
    for (int rep = 0; rep < 2; ++rep) { void *p; double t0 = now(); hipMalloc(&p, (size_t)13 << 30); double t1 = now(); hipFree(p); printf("13 GiB again: %.1f ms\n", t1 - t0); }
    return 0;
}
