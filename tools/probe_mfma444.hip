// probe_mfma444.hip -- determines the lane layout of v_mfma_f64_4x4x4_4b_f64 on the device:
// for every D lane prints the (A lane, B lane) pairs that are multiplied into it.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *a, const double *b, double *d) {
    d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main() {
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
    int pairs[64][8]; int np[64] = {0};
    for (int x = 0; x < 64; ++x) {
        for (int l = 0; l < 64; ++l) { ha[l] = (l == x) ? 1.0 : 0.0; hb[l] = l + 1; }
        (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, dd); (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0 && np[l] < 4) { pairs[l][2 * np[l]] = x; pairs[l][2 * np[l] + 1] = (int)hd[l] - 1; np[l]++; }
    }
    for (int l = 0; l < 64; ++l) {
        printf("D%02d:", l);
        for (int q = 0; q < np[l]; ++q) printf(" (A%02d,B%02d)", pairs[l][2 * q], pairs[l][2 * q + 1]);
        printf("\n");
    }
    return 0;
}
