// nmpc_device.h -- device-side building blocks of the batched NMPC solver (gfx950 / CDNA4 only).
//
// Mapping: one wavefront (64 lanes) = one workgroup.  A query point owns a GROUP of lanes (P = 20:
// three groups per wave for N_hor <= 20; P = 32: two groups; P = 64: one), lane t of the group owns
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
// cross-lane primitives: DPP row operations (VALU speed, no LDS traffic) + gfx950 permlane swaps
// ---------------------------------------------------------------------------------------------
// DPP control words (CDNA ISA): quad_perm 0x00-0xFF, row_shl:n 0x100+n, row_shr:n 0x110+n,
// wave_shl:1 0x130, wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141, row_bcast:15 0x142,
// row_bcast:31 0x143, row_newbcast:n 0x150+n
template <int CTRL, int ROW_MASK = 0xF, bool BOUND_CTRL = true>
__device__ __forceinline__ double dpp_mov(double v)
{
    // lanes whose source is out of range (BOUND_CTRL) or whose row is masked off read 0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, BOUND_CTRL);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, BOUND_CTRL);
    return __hiloint2double(hi, lo);
}

// same, but lanes of masked-off rows keep `old` (the destination register is tied to it)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_old(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
// +0.0 in one instruction (as two 32-bit halves the compiler spends two v_mov_b32 on it)
__device__ __forceinline__ double zero_pair()
{
    double z;
    asm("v_mov_b64 %0, 0" : "=v"(z));
    return z;
}

__device__ __forceinline__ double lane_get(double v, int src_lane)
{
    // ds_bpermute_b32 x2: pull `v` from an arbitrary lane of the wave (slow path, rarely used)
    const int idx = src_lane << 2;
    int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int lane_get_i(int v, int src_lane)
{
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}

// v_permlane16_swap / v_permlane32_swap rewrite BOTH their operands, so "the value of the other row / half" needs the value in two
// registers.  They are never handed the same value twice (swap(x, x)): ROCm 7.2's register coalescer then joins the copy with x into one
// wide virtual register and leaves a read-undef flag on the swap's tied second def that declares the other lanes of x dead, and a
// scheduler that moves the copy above the instruction producing x (-amdgpu-sched-strategy=max-ilp does) swaps a stale register -- wrong
// results that depend on what ran before (codegen_check.py finds both the flag and the move).  Instead the
// second register is a full 64-bit definition of its own: the producing addition issued twice (one v_add_f64 instead of the two v_mov_b32
// of a copy -- `opaque` keeps the compiler from merging them), or an opaque copy where there is no producing instruction to repeat.
__device__ __forceinline__ double opaque(double x)
{
    asm("" : "+v"(x));
    return x;
}
// a <- [a.r0 b.r0 a.r2 b.r2], b <- [a.r1 b.r1 a.r3 b.r3] (rows of 16 lanes): odd rows of a change places with even rows of b
__device__ __forceinline__ void swap_rows(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
// a <- [a.h0 b.h0], b <- [a.h1 b.h1] (halves of 32 lanes)
__device__ __forceinline__ void swap_halves(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}

// bitwise OR over all 64 lanes, returned as a wave-uniform scalar
__device__ __forceinline__ unsigned wave_or(unsigned v)
{
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    // one lane per row now holds the row's OR in every lane of the row: combine the four rows
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

// sum over the P lanes of the group, result in every lane.  Butterfly over lane^1, ^2, ^4, ^8, ^16
// (^32); a + b == b + a bitwise, so every lane ends with the value of the adjacent-pair tree.
// the sum over the 32 lanes of each half in TWO registers (every lane of a half holds its half's sum in both)
__device__ __forceinline__ void half_sum_twice(double v, double &s1, double &s2)
{
    v = v + dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);      // row_half_mirror: the other quad of the 8-lane block
    const double o = dpp_mov<0x140>(v);      // row_mirror: the other 8-lane block of the row
    double a = v + o, b = v + opaque(o);
    swap_rows(a, b);                // a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]: the other row of the 32-lane half
    s1 = a + b;
    s2 = a + opaque(b);
}
template <int P>
__device__ __forceinline__ double group_sum(double v, int lane)
{
    (void)lane;
    double s1, s2;
    half_sum_twice(v, s1, s2);
    if (P == 64) {
        swap_halves(s1, s2);        // s1 = [h0 h0], s2 = [h1 h1]
        return s1 + s2;
    }
    return s1;
}

// inclusive prefix sum over the stages of the group: Kogge-Stone inside 16-lane rows, then carries
template <int P>
__device__ __forceinline__ double group_prefix(double v, int lane)
{
    v = v + dpp_mov<0x111>(v);                  // row_shr:1, zero fill
    v = v + dpp_mov<0x112>(v);
    v = v + dpp_mov<0x114>(v);
    v = v + dpp_mov<0x118>(v);
    v = v + dpp_mov<0x142, 0xA, false>(v);      // row_bcast:15 into rows 1 and 3
    if (P == 64) v = v + dpp_mov<0x143, 0xC, false>(v);   // row_bcast:31 into rows 2 and 3
    (void)lane;
    return v;
}

// inclusive suffix sum over the stages of the group (mirror image)
template <int P>
__device__ __forceinline__ double group_suffix(double v, int lane)
{
    v = v + dpp_mov<0x101>(v);                  // row_shl:1, zero fill
    v = v + dpp_mov<0x102>(v);
    v = v + dpp_mov<0x104>(v);
    v = v + dpp_mov<0x108>(v);
    {   // even rows add the first lane of the row after them
        const double first = dpp_mov<0x150, 0xF, false>(v);          // row_newbcast:0
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(first), 0, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(first), 0, false, false);
        v = v + __hiloint2double(hi[1], lo[1]);                      // [r1 0 r3 0]
    }
    if (P == 64) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 32);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 32);
        const double c = __hiloint2double(hi, lo);
        v = v + ((lane & 32) ? 0.0 : c);
    }
    return v;
}

// value held by stage t-1 (fill for t == 0) / stage t+1 (0 for the last lane of the group)
template <int P>
__device__ __forceinline__ double from_prev(double v, int lane, double fill)
{
    const double o = dpp_mov<0x138>(v);         // wave_shr:1
    return (lane & (P - 1)) == 0 ? fill : o;
}
template <int P>
__device__ __forceinline__ double from_next(double v, int lane)
{
    const double o = dpp_mov<0x130>(v);         // wave_shl:1
    return (lane & (P - 1)) == P - 1 ? 0.0 : o;
}

// ---------------------------------------------------------------------------------------------
// lane <-> (group, stage) mapping.  P = 32 / 64: groups of P consecutive lanes.
// P = 20 is the THREE-groups-per-wave layout for N_hor <= 20 ("tri"): group q = 0..2 owns row q of the
// wave (lanes 16q..16q+15, stages 0..15) plus quad q of row 3 (lanes 48+4q..51+4q, stages 16..19);
// lanes 60..63 are stages 20..23 of group 2: beyond the horizon like stages N..31 of a P = 32 group
// (their vectors are zero, the group's scalars reach them), with LDS columns of their own.
// ---------------------------------------------------------------------------------------------
template <int P> __device__ __forceinline__ int lay_group(int lane) { return lane / P; }
template <int P> __device__ __forceinline__ int lay_stage(int lane) { return lane % P; }
template <int P> constexpr int lay_cols() { return P == 20 ? 24 : P; }      // per-stage LDS columns of a group
template <> __device__ __forceinline__ int lay_group<20>(int lane)
{
    return lane < 48 ? lane >> 4 : (lane < 60 ? (lane - 48) >> 2 : 2);
}
template <> __device__ __forceinline__ int lay_stage<20>(int lane)
{
    return lane < 48 ? lane & 15 : (lane < 60 ? 16 + (lane & 3) : 20 + (lane & 3));
}
// lane that holds stage t (< N_hor) of group q
template <int P> __device__ __forceinline__ int lay_lane(int q, int t) { return q * P + t; }
template <> __device__ __forceinline__ int lay_lane<20>(int q, int t) { return t < 16 ? 16 * q + t : 48 + 4 * q + (t - 16); }

// The tri layout computes the SAME canonical tree / scans as P = 32 with stages 20..31 absent (they
// are zero there): levels 1, 2 run in quads everywhere, levels 4, 8 only in rows 0..2 (the tail
// quads add 0.0, as the zero padding does), and the row <-> tail exchange is one ds_bpermute.
template <>
__device__ __forceinline__ double group_sum<20>(double v, int lane)
{
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    {   // lanes 60..63 (stages 20..23 of group 2) take the sum of the group's tail quad: row_shr:4 into bank 3 of row 3
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x114, 0x8, 0x8, false);
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x114, 0x8, 0x8, false);
        v = __hiloint2double(hi, lo);
    }
    double o = dpp_mov_old<0x141, 0x7>(zero_pair(), v);     // rows 0..2 only; row 3 adds 0.0
    v = v + o;
    o = dpp_mov_old<0x140, 0x7>(o, v);                      // (row 3 of `o` is still 0.0)
    v = v + o;
    const int q = lay_group<20>(lane);
    const int partner = lane < 48 ? 48 + 4 * q : 16 * q;
    return v + lane_get(v, partner);        // block 0 + block 1 (commutative: same bits on both sides)
}

template <>
__device__ __forceinline__ double group_prefix<20>(double v, int lane)
{
    const bool tail = lane >= 48;
    double o = dpp_mov<0x111>(v);
    v = v + ((tail && (lane & 3) < 1) ? 0.0 : o);
    o = dpp_mov<0x112>(v);
    v = v + ((tail && (lane & 3) < 2) ? 0.0 : o);
    o = dpp_mov_old<0x114, 0x7>(zero_pair(), v);
    v = v + o;
    o = dpp_mov_old<0x118, 0x7>(o, v);
    v = v + o;
    const int q = (lane - 48) >> 2;
    const double carry = lane_get(v, (tail && lane < 60) ? 16 * q + 15 : lane);     // last entry of block 0
    return v + (tail ? carry : 0.0);
}

template <>
__device__ __forceinline__ double group_suffix<20>(double v, int lane)
{
    const bool tail = lane >= 48;
    double o = dpp_mov<0x101>(v);
    v = v + ((tail && (lane & 3) > 2) ? 0.0 : o);
    o = dpp_mov<0x102>(v);
    v = v + ((tail && (lane & 3) > 1) ? 0.0 : o);
    o = dpp_mov_old<0x104, 0x7>(zero_pair(), v);
    v = v + o;
    o = dpp_mov_old<0x108, 0x7>(o, v);
    v = v + o;
    const double carry = lane_get(v, tail ? lane : 48 + 4 * (lane >> 4));            // first entry of block 1
    return v + (tail ? 0.0 : carry);
}

template <>
__device__ __forceinline__ double from_prev<20>(double v, int lane, double fill)
{
    const int src = (lane >= 48 && (lane & 3) == 0) ? (lane < 60 ? 16 * ((lane - 48) >> 2) + 15 : lane) : lane - 1;
    const double o = lane_get(v, src);
    return (lane < 48 && (lane & 15) == 0) ? fill : o;
}
template <>
__device__ __forceinline__ double from_next<20>(double v, int lane)
{
    const int src = (lane < 48 && (lane & 15) == 15) ? 48 + 4 * (lane >> 4) : lane + 1;
    const double o = lane_get(v, src);
    return (lane >= 48 && (lane & 3) == 3) ? 0.0 : o;
}

// inclusive prefix sum AND the inclusive sum of the stage before (`excl`; 0.0 at stage 0): what from_prev(group_prefix(v)) would
// deliver, without a second exchange -- in the tri layout the value a tail quad's first stage needs from its row (the inclusive sum
// of stage 15) is the very carry the prefix sum has just fetched.
template <int P>
__device__ __forceinline__ double group_prefix_ex(double v, int lane, double &excl)
{
    const double incl = group_prefix<P>(v, lane);
    excl = from_prev<P>(incl, lane, 0.0);
    return incl;
}
template <>
__device__ __forceinline__ double group_prefix_ex<20>(double v, int lane, double &excl)
{
    const bool tail = lane >= 48;
    double o = dpp_mov<0x111>(v);
    v = v + ((tail && (lane & 3) < 1) ? 0.0 : o);
    o = dpp_mov<0x112>(v);
    v = v + ((tail && (lane & 3) < 2) ? 0.0 : o);
    o = dpp_mov_old<0x114, 0x7>(zero_pair(), v);
    v = v + o;
    o = dpp_mov_old<0x118, 0x7>(o, v);
    v = v + o;
    const int q = (lane - 48) >> 2;
    const double carry = lane_get(v, (tail && lane < 60) ? 16 * q + 15 : lane);     // last entry of block 0
    const double incl = v + (tail ? carry : 0.0);
    const double sh = dpp_mov<0x111>(incl);                                          // row_shr:1, zero fill at the start of a row
    excl = (tail && (lane & 3) == 0) ? carry : sh;
    return incl;
}

// The same for a wave whose lanes 60..63 (stages 20..23 of group 2: beyond every horizon) hold ZEROS in `v` -- the hybrid kernel's owner
// path, whose query points arrive through LDS with zero pads in those places.  The inclusive sum of lane 60 is then +0.0 exactly (the
// masked first level adds +0.0 to whichever sign the zero came with), so the lanes of rows 0..2 fetch their "carry" from lane 60 instead
// of from themselves and the addition needs no select: v + (+0.0) is the very operation the select form performs.  Same bits, two
// v_cndmask less per scan.
__device__ __forceinline__ double group_prefix_ex_z60(double v, int lane, double &excl)
{
    const bool tail = lane >= 48;
    double o = dpp_mov<0x111>(v);
    v = v + ((tail && (lane & 3) < 1) ? 0.0 : o);
    o = dpp_mov<0x112>(v);
    v = v + ((tail && (lane & 3) < 2) ? 0.0 : o);
    o = dpp_mov_old<0x114, 0x7>(zero_pair(), v);
    v = v + o;
    o = dpp_mov_old<0x118, 0x7>(o, v);
    v = v + o;
    const int q = (lane - 48) >> 2;
    const double carry = lane_get(v, (tail && lane < 60) ? 16 * q + 15 : 60);       // last entry of block 0 | the +0.0 of lane 60
    const double incl = v + carry;
    const double sh = dpp_mov<0x111>(incl);
    excl = (tail && (lane & 3) == 0) ? carry : sh;
    return incl;
}

// ---------------------------------------------------------------------------------------------
// sin and cos: Cody-Waite reduction by pi/2 in three fma steps + fdlibm minimax kernels (Horner, fma)
// ---------------------------------------------------------------------------------------------
// The ten coefficients that enter a Horner step as the ADDEND can be read from a table in LDS
// (`tab`, filled from CW_COEF_DEV): as literals the compiler keeps them in twenty long-lived VGPRs
// (and copies them before every v_fmac); as LDS operands they are transient.
constexpr int CW_NCOEF = 10;
#define NMPC_CW_COEFS                                                                              \
    -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,           \
    8.33333333332248946124e-03, -1.66666666666666324348e-01, 2.08757232129817482790e-09,            \
    -2.75573143513906633035e-07, 2.48015872894767294178e-05, -1.38888888888741095749e-03,           \
    4.16666666666666019037e-02
__device__ const double CW_COEF_DEV[CW_NCOEF] = {NMPC_CW_COEFS};
struct CwLiteral {
    __device__ __forceinline__ double operator[](int i) const
    {
        constexpr double k[CW_NCOEF] = {NMPC_CW_COEFS};
        return k[i];
    }
};
#undef NMPC_CW_COEFS

template <class TAB>
__device__ __forceinline__ void sincos_cw_t(double x, TAB tab, double &s, double &c)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = fma(1.58969099521155010221e-10, z, tab[0]);
    ps = fma(ps, z, tab[1]);
    ps = fma(ps, z, tab[2]);
    ps = fma(ps, z, tab[3]);
    ps = fma(ps, z, tab[4]);
    const double sr = fma(r * z, ps, r);
    double pc = fma(-1.13596475577881948265e-11, z, tab[5]);
    pc = fma(pc, z, tab[6]);
    pc = fma(pc, z, tab[7]);
    pc = fma(pc, z, tab[8]);
    pc = fma(pc, z, tab[9]);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int n = ((int)k) & 3;
    double so = (n & 1) ? cr : sr;
    double co = (n & 1) ? sr : cr;
    if (n & 2) so = -so;
    if ((n + 1) & 2) co = -co;
    s = so;
    c = co;
}
__device__ __forceinline__ void sincos_cw(double x, double &s, double &c) { sincos_cw_t(x, CwLiteral(), s, c); }

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

}  // namespace nmpc
