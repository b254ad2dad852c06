// latency micro-benchmarks for the primitives the NMPC kernel's critical path is made of (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpc_trajectory_generator_amd/csrc/nmpc_device.h"
using namespace nmpc;
#define REP 256
#define PIN(v) do { asm volatile("s_nop 0" : "+v"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ long long TICK() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ void k(double *out, long long *cyc, double seed, int nwaves_dummy)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x & 1023] = seed * threadIdx.x;
    __syncthreads();
    double x = seed + lane, y = seed * 0.5;
    long long t0, t1;
    int r = 0;
    // 1. dependent fma chain
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = fma(x, 1.0000001, y);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 2. 4 independent fma chains
    double x1 = x + 1, x2 = x + 2, x3 = x + 3;
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) { x = fma(x, 1.0000001, y); x1 = fma(x1, 1.0000001, y); x2 = fma(x2, 1.0000001, y); x3 = fma(x3, 1.0000001, y); }
    PIN(x); t1 = TICK(); PIN(x);
    x += x1 + x2 + x3;
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 3. group_sum chain (dependent)
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = group_sum<32>(x * 0.03125, lane);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 4. prefix chain
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = group_prefix<32>(x * 0.03125, lane);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 5. dependent LDS read chain (ds_read_b64, address from previous value)
    int idx = lane;
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) { double v = lds[idx & 1023]; idx = (int)v & 1023; x += v; }
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 6. sincos chain
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 32; ++i) { double s, c; sincos_cw(x, s, c); x = s + c; }
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 7. dependent division chain
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 32; ++i) x = 1.0 / (x + 2.0);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 8. dependent sqrt chain
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 32; ++i) x = sqrt(x + 2.0);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 9. dependent max/min/add mix (v_max_f64, v_min_f64)
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = fmin(fmax(x * 0.999, 0.0), 1.0e9) + y;
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    out[threadIdx.x] = x + idx;
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64 * 8);
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.25, 0); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"dep fma x256", "4 indep fma chains x256 (1024 fma)", "group_sum x64", "group_prefix x64", "dep LDS read x64",
                           "sincos x32", "div x32", "sqrt x32", "mul,max,min,add x256"};
    const double per[] = {256, 1024, 64, 64, 64, 32, 32, 32, 256};
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9? print raw and per-op
    for (int i = 0; i < 9; ++i) printf("%-40s %8lld ticks  %8.2f per op\n", names[i], h[i], h[i] / per[i]);
    return 0;
}
