// fetch_cal.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS project's access widths (profiles/r04/traffic.json).
// The solve kernels read p at 8 bytes per lane (global_load_dwordx2, coalesced) and write u / y / status at 8 bytes per lane; the guide's
// "FETCH_SIZE reports half the bytes" is calibrated on 16-byte-per-lane streaming reads only.  Three kernels over one 1 GiB buffer
// (beyond the 256 MiB Infinity Cache): read at 8 B/lane, read at 16 B/lane, write at 8 B/lane; run under
//   rocprofv3 --pmc FETCH_SIZE -- scripts/ubench/fetch_cal      and      --pmc WRITE_SIZE
// and compare the counter (KiB) per kernel with the 1 048 576 KiB each kernel moves.
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/fetch_cal scripts/ubench/fetch_cal.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read8(const double *p, size_t n, double *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678) *sink = acc;
}
__global__ void read16(const double2 *p, size_t n, double *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) *sink = acc;
}
__global__ void write8(double *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 8;
    double *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2 *)buf, n / 2, sink);
        hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, buf, n);
    }
    hipDeviceSynchronize();
    std::printf("each kernel moves %zu KiB\n", bytes >> 10);
    return 0;
}
