// nmpc_device.h -- device-side building blocks of the batched NMPC solver (gfx950 / CDNA4 only).
//
// Mapping: one wavefront (64 lanes) = one workgroup.  A problem instance owns a GROUP of P lanes
// (P = 32 for N_hor <= 32 -> two instances per wave; P = 64 otherwise), lane t of the group owns
// stage t of the horizon: its control pair (v_t, w_t), its post-update state (x_{t+1}, y_{t+1},
// theta_{t+1}) and the adjoints of both.  Everything that is uniform over an instance lives in the
// group's LDS slice; horizon sums / scans are cross-lane operations inside the group.
//
// Arithmetic contract ("canonical arithmetic"): this file is compiled with
// -ffp-contract=off, every fused multiply-add is an explicit fma(), and the cross-lane reductions
// have a fixed shape, so results are reproducible bit for bit by any IEEE-754 f64 machine:
//   group_sum     adjacent-pair binary tree over the P lanes
//   prefix/suffix Kogge-Stone inside 16-lane rows (what DPP row shifts give), then row carries
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmpc {

// ---------------------------------------------------------------------------------------------
// cross-lane primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_get(double v, int src_lane)
{
    // ds_bpermute_b32 x2: pull `v` from an arbitrary lane of the wave
    const int idx = src_lane << 2;
    int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int lane_get_i(int v, int src_lane)
{
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}

// sum over the P lanes of the group, result in every lane (butterfly; a + b == b + a bitwise, so
// every lane holds the value of the adjacent-pair tree)
template <int P>
__device__ __forceinline__ double group_sum(double v, int lane)
{
#pragma unroll
    for (int off = 1; off < P; off <<= 1) v = v + lane_get(v, lane ^ off);
    return v;
}

// inclusive prefix sum over the stages of the group
template <int P>
__device__ __forceinline__ double group_prefix(double v, int lane)
{
    const int r = lane & 15;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const double o = lane_get(v, (lane - off) & 63);
        if (r >= off) v = v + o;
    }
    {   // odd rows add the last lane of the row before them
        const double c = lane_get(v, ((lane & ~15) - 1) & 63);
        if (lane & 16) v = v + c;
    }
    if (P == 64) {
        const double c = lane_get(v, 31);
        if (lane & 32) v = v + c;
    }
    return v;
}

// inclusive suffix sum over the stages of the group (mirror image)
template <int P>
__device__ __forceinline__ double group_suffix(double v, int lane)
{
    const int r = lane & 15;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const double o = lane_get(v, (lane + off) & 63);
        if (r + off <= 15) v = v + o;
    }
    {   // even rows add the first lane of the row after them
        const double c = lane_get(v, ((lane | 15) + 1) & 63);
        if (!(lane & 16)) v = v + c;
    }
    if (P == 64) {
        const double c = lane_get(v, 32);
        if (!(lane & 32)) v = v + c;
    }
    return v;
}

// value held by stage t-1 (fill for t == 0) / stage t+1 (0 for the last lane of the group)
template <int P>
__device__ __forceinline__ double from_prev(double v, int lane, double fill)
{
    const double o = lane_get(v, (lane - 1) & 63);
    return (lane & (P - 1)) == 0 ? fill : o;
}
template <int P>
__device__ __forceinline__ double from_next(double v, int lane)
{
    const double o = lane_get(v, (lane + 1) & 63);
    return (lane & (P - 1)) == P - 1 ? 0.0 : o;
}

// ---------------------------------------------------------------------------------------------
// sin and cos: Cody-Waite reduction by pi/2 in three fma steps + fdlibm minimax kernels (Horner, fma)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_cw(double x, double &s, double &c)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = fma(1.58969099521155010221e-10, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int n = ((int)k) & 3;
    double so = (n & 1) ? cr : sr;
    double co = (n & 1) ? sr : cr;
    if (n & 2) so = -so;
    if ((n + 1) & 2) co = -co;
    s = so;
    c = co;
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

}  // namespace nmpc
