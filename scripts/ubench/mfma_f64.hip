// mfma_f64.hip -- what v_mfma_f64_4x4x4_4b computes on gfx950, exactly: operand layouts, the order and rounding of the k-sum,
// its latency in dependent chains, and v_fmac_f64 with a DPP row_newbcast source.  The Gram-form L-BFGS of the solve kernels
//  rests on these facts; tests/test_gpu_mfma.py holds the library to them on every GPU run.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o mfma_f64 mfma_f64.hip && ./mfma_f64 out.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define PIN(v) do { asm volatile("s_nop 0" : "+v"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ long long TICK() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

// trials x 64 lanes: D = mfma_4x4x4_4b(A, B, C)
__global__ void k_mfma(const double *A, const double *B, const double *C, double *D, int trials)
{
    const int lane = threadIdx.x;
    for (int s = 0; s < trials; ++s) {
        const double a = A[s * 64 + lane], b = B[s * 64 + lane], c = C[s * 64 + lane];
        D[s * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    }
}
// the same with two chained instructions: D = mfma(A2, B2, mfma(A, B, C))
__global__ void k_mfma2(const double *A, const double *B, const double *C, double *D, int trials)
{
    const int lane = threadIdx.x;
    for (int s = 0; s + 1 < trials; s += 2) {
        const double a = A[s * 64 + lane], b = B[s * 64 + lane], c = C[s * 64 + lane];
        const double a2 = A[(s + 1) * 64 + lane], b2 = B[(s + 1) * 64 + lane];
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, d, 0, 0, 0);
        D[s * 64 + lane] = d;
        D[(s + 1) * 64 + lane] = 0.0;
    }
}
// v_fmac_f64 with DPP row_newbcast:J on src0: acc += x[lane J of my row] * y
template <int J>
__device__ __forceinline__ double fmac_bcast(double acc, double x, double y)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
    return acc;
}
__global__ void k_fmac(const double *X, const double *Y, const double *Z, double *out)
{
    const int lane = threadIdx.x;
    const double x = X[lane], y = Y[lane], z = Z[lane];
    out[lane] = fmac_bcast<0>(z, x, y);
    out[64 + lane] = fmac_bcast<5>(z, x, y);
    out[128 + lane] = fmac_bcast<15>(z, x, y);
}
__global__ void k_lat(double *out, long long *cyc, double seed)
{
    const int lane = threadIdx.x & 63;
    double x = seed + lane, y = seed * 0.5 + 1e-3 * lane, z = 1.0;
    long long t0, t1;
    int r = 0;
    // 0: C-dependent chain of 64 mfma 4x4x4
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_mfma_f64_4x4x4f64(y, z, x, 0, 0, 0);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 1: A-dependent chain of 64
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, z, y, 0, 0, 0);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 2: two interleaved C-dependent chains of 32 each
    double x2 = x + 1.0;
    PIN(x); PIN(x2); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 32; ++i) { x = __builtin_amdgcn_mfma_f64_4x4x4f64(y, z, x, 0, 0, 0); x2 = __builtin_amdgcn_mfma_f64_4x4x4f64(z, y, x2, 0, 0, 0); }
    PIN(x); PIN(x2); t1 = TICK(); PIN(x);
    x += x2;
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 3: 64 independent mfma (throughput): 8 accumulators
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = x + j;
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(y, z, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) PIN(acc[j]);
    t1 = TICK(); PIN(x);
#pragma unroll
    for (int j = 0; j < 8; ++j) x += acc[j];
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 4: C-dependent chain of 32 mfma with 4 independent v_fma_f64 between each (does the VALU run under the MFMA?)
    double w0 = y, w1 = y + 1, w2 = y + 2, w3 = y + 3;
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        x = __builtin_amdgcn_mfma_f64_4x4x4f64(y, z, x, 0, 0, 0);
        w0 = fma(w0, 1.0000001, y); w1 = fma(w1, 1.0000001, y); w2 = fma(w2, 1.0000001, y); w3 = fma(w3, 1.0000001, y);
    }
    PIN(x); PIN(w0); PIN(w1); PIN(w2); PIN(w3); t1 = TICK(); PIN(x);
    x += w0 + w1 + w2 + w3;
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 5: the 4 x 32 v_fma_f64 alone
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 32; ++i) { w0 = fma(w0, 1.0000001, y); w1 = fma(w1, 1.0000001, y); w2 = fma(w2, 1.0000001, y); w3 = fma(w3, 1.0000001, y); }
    PIN(w0); PIN(w1); PIN(w2); PIN(w3); t1 = TICK(); PIN(x);
    x += w0 + w1 + w2 + w3;
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 6: dependent chain of 64 v_fmac_f64_dpp row_newbcast
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fmac_bcast<3>(x, x, y);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 7: dependent chain of 64 plain v_fma_f64 (reference)
    PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fma(x, 1.0000001, y);
    PIN(x); t1 = TICK(); PIN(x);
    if (lane == 0) cyc[r] = t1 - t0; r++;
    // 8: C-dependent chain of 16 v_mfma_f64_16x16x4
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c4 = {x, x + 1, x + 2, x + 3};
        PIN(x); t0 = TICK(); PIN(x);
#pragma unroll
        for (int i = 0; i < 16; ++i) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c4, 0, 0, 0);
        x = c4[0] + c4[1] + c4[2] + c4[3];
        PIN(x); t1 = TICK(); PIN(x);
        if (lane == 0) cyc[r] = t1 - t0; r++;
    }
    out[threadIdx.x] = x;
}

static double rnd_wide(unsigned &s)
{
    s = s * 1664525u + 1013904223u;
    const double m = 1.0 + (double)(s >> 8) / 16777216.0;
    s = s * 1664525u + 1013904223u;
    const int e = (int)((s >> 16) % 41) - 20;
    s = s * 1664525u + 1013904223u;
    const double sg = (s & 0x10000u) ? -1.0 : 1.0;
    s = s * 1664525u + 1013904223u;
    const double m2 = (double)(s >> 8) / 16777216.0 * 5.9604644775390625e-08;     // low mantissa bits
    return sg * ldexp(m + m2, e);
}

int main(int argc, char **argv)
{
    const int T = 256;
    std::vector<double> A(T * 64), B(T * 64), C(T * 64), D(T * 64), D2(T * 64);
    unsigned s = 12345u;
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l) {
            if (t < 64) {            // one-hot A (lane t), B = 1 + lane / 64, C = 0: where does A[lane t] go?
                A[t * 64 + l] = l == t ? 1.0 : 0.0; B[t * 64 + l] = 1.0 + l / 64.0; C[t * 64 + l] = 0.0;
            } else if (t < 128) {    // one-hot B
                B[t * 64 + l] = l == t - 64 ? 1.0 : 0.0; A[t * 64 + l] = 1.0 + l / 64.0; C[t * 64 + l] = 0.0;
            } else {                 // wide-range random: the order and the rounding of the k-sum
                A[t * 64 + l] = rnd_wide(s); B[t * 64 + l] = rnd_wide(s); C[t * 64 + l] = (t & 1) ? rnd_wide(s) : 0.0;
            }
        }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, T * 64 * 8); hipMalloc(&dB, T * 64 * 8); hipMalloc(&dC, T * 64 * 8); hipMalloc(&dD, T * 64 * 8);
    hipMemcpy(dA, A.data(), T * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), T * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), T * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, T);
    hipMemcpy(D.data(), dD, T * 64 * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_mfma2, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, T);
    hipMemcpy(D2.data(), dD, T * 64 * 8, hipMemcpyDeviceToHost);
    std::vector<double> F(192);
    hipLaunchKernelGGL(k_fmac, dim3(1), dim3(64), 0, 0, dA + 200 * 64, dB + 200 * 64, dC + 201 * 64, dD);
    hipMemcpy(F.data(), dD, 192 * 8, hipMemcpyDeviceToHost);
    long long *dcyc, cyc[16] = {0};
    hipMalloc(&dcyc, 16 * 8);
    hipMemset(dcyc, 0, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, dD, dcyc, 1.25);
    hipMemcpy(cyc, dcyc, 16 * 8, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "device error\n"); return 1; }
    const char *names[] = {"mfma4x4x4 C-dep chain /64", "mfma4x4x4 A-dep chain /64", "2 interleaved C-dep chains /64", "64 independent (8 acc) /64",
                           "C-dep chain /32 with 4 fma between", "the 4x32 fma alone /32", "fmac_dpp newbcast chain /64", "fma chain /64", "mfma16x16x4 C-dep /16"};
    const int div[] = {64, 64, 64, 64, 32, 32, 64, 64, 16};
    for (int i = 0; i < 9; ++i) printf("%-40s %8.1f cycles each\n", names[i], (double)cyc[i] / div[i]);
    // fmac check
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15;
        const double *X = A.data() + 200 * 64, *Y = B.data() + 200 * 64, *Z = C.data() + 201 * 64;
        if (F[l] != fma(X[row + 0], Y[l], Z[l])) bad++;
        if (F[64 + l] != fma(X[row + 5], Y[l], Z[l])) bad++;
        if (F[128 + l] != fma(X[row + 15], Y[l], Z[l])) bad++;
    }
    printf("v_fmac_f64_dpp row_newbcast: %d mismatches of 192 (acc += x[row lane J] * y)\n", bad);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        int hdr[2] = {T, 64};
        fwrite(hdr, 4, 2, f);
        fwrite(A.data(), 8, T * 64, f); fwrite(B.data(), 8, T * 64, f); fwrite(C.data(), 8, T * 64, f);
        fwrite(D.data(), 8, T * 64, f); fwrite(D2.data(), 8, T * 64, f);
        fclose(f);
    }
    return 0;
}
