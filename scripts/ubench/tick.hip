#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long n, long long *out) {
    long long t0 = __builtin_amdgcn_s_memtime(), t;
    long long c0 = __builtin_readcyclecounter();
    do { t = __builtin_amdgcn_s_memtime(); } while (t - t0 < n);
    out[0] = t - t0; out[1] = __builtin_readcyclecounter() - c0;
    out[2] = wall_clock64();
}
int main() {
    long long *d; hipMalloc(&d, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) {
        hipEventRecord(a); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 100000000LL, d); hipEventRecord(b); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("s_memtime ticks %lld in %.3f ms -> %.1f MHz ; readcyclecounter %lld -> %.1f MHz\n", h[0], ms, h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
    int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clockRate attr %d kHz\n", clk);
    return 0;
}
