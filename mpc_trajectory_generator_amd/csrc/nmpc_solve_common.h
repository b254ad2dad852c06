// nmpc_solve_common.h -- what the solve kernels share: the phase states of their flag-driven state machines, the two-halves reduction
// helpers of the state layout (stage t at lane t of both 32-lane halves), the forward-backward envelope.
// (Until round 5 this was the head of nmpc_solve_dual.h, the two-point kernel for 20 < N_hor <= 32 -- the design the three-point kernels grew
// from.  That kernel is retired: the two-stage kernel of nmpc_solve_hyb2.h serves those horizons, with the Gram-form L-BFGS and the obstacle
// certificate the two-point kernel never got, so that the oracle has ONE arithmetic for every N_hor <= 40.  Measured on MI355X before it went,
// B = 4096: N = 24 / 27 / 32  63.5 / 70.6 / 93.7 ms for the two-point kernel, 70.9 / 80.3 / 96.4 ms for the run-time-shape two-stage kernel.)
#pragma once

namespace nmpc {

// value held by the same stage in half 0 / half 1, delivered to both halves (the second register of the swap is an opaque copy:
// swap_halves in nmpc_device.h has the reason)
__device__ __forceinline__ void both_halves(double v, double &from_h0, double &from_h1)
{
    double a = v, b = opaque(v);
    swap_halves(a, b);
    from_h0 = a;
    from_h1 = b;
}

enum : int { D_INIT = 0, D_LIP, D_ITER, D_LS, D_ALM, D_FB };

// unconditional LDS load of a (v, w) pair, zeroed for lanes beyond the horizon
__device__ __forceinline__ dbl2 ld_pair(const lds_double2 *base, int idx, bool keep)
{
    dbl2 v = base[idx];
    if (!keep) { v.x = 0.0; v.y = 0.0; }
    return v;
}

// two horizon sums for the price of one tree: the solver state is replicated in both 32-lane halves of the wave, so half 0
// reduces `a`, half 1 reduces `b` (each the canonical 32-entry tree of its half) and a permlane32 swap hands both results
// to every lane.  Same bits as two separate group_sum<32>.
__device__ __forceinline__ void pair_sum(double a, double b, int lane, double &sum_a, double &sum_b)
{
    double s1, s2;
    half_sum_twice((lane & 32) ? b : a, s1, s2);
    swap_halves(s1, s2);
    sum_a = s1;
    sum_b = s2;
}

// forward-backward envelope at the point whose cost / gradient / gradient step / half step are given; hig = 0.5 / gamma (formed when gamma changes)
template <int P>
__device__ __forceinline__ double fbe_value(double cost, double gamma, double hig, double sv, double sw, double hv, double hw,
                                            double gv, double gw, int lane)
{
    const double e1 = sv - hv, e2 = sw - hw;
    double dist2, gg;
    pair_sum(fma(e1, e1, e2 * e2), fma(gv, gv, gw * gw), lane, dist2, gg);
    return cost - (0.5 * gamma) * gg + dist2 * hig;
}


}  // namespace nmpc
