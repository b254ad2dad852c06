// nmpc_solve_hyb.h -- the N_hor <= 20 solver: ONE problem instance per wavefront, THREE query points
// per pass, TWO lane layouts.
//
// The evaluation of psi runs in the "tri" layout of nmpc_device.h (rows 0..2 of the wave hold stages
// 0..15 of query points 0..2, the quads of row 3 stages 16..19): three points per pass.  The solver
// state (u, grad, half step, residual, direction, L-BFGS ring) lives in the two-halves layout (the retired two-point
// kernel: stage t at lane t of BOTH 32-lane halves, so every reduction of the PANOC / L-BFGS code is a
// pure DPP + permlane tree (no LDS round trip; the tri layout's row <-> tail exchange costs one per
// reduction, and the two-loop recursion alone chains twenty of them per iteration).  Query points are
// carried from the state layout to the evaluation layout, and gradients back, THROUGH LDS (the "transport" area: the place of the Gram
// batch's four vectors, which are dead between the batch and the next step) -- two ds_write_b128 and two ds_read_b128 where the
// ds_bpermute form took twelve cross-lane fetches and eight selects, and the kernel is bound by instruction issue:
//   X = per-half point (half 0 / half 1 prepare different points), Y = a third point known to both.
//   initialisation   X = (u | u + h)                 Y = u
//   iteration >= 1   X = (u_bar | u+(tau = 1))       Y = u+(tau = 1/2)
//   line search      X = (u+(tau) | u+(tau/2))       Y = u+(tau/4)
// Trials are consumed in order and the first accepted one ends the iteration, exactly as the
// sequential line search would; evaluations past the accepted trial are discarded and not counted.
// Same results and counters as the sequential oracle.
//
// Migration.  With two waves resident per SIMD the hardware serves the older wave slot first: measured on
// MI355X (round 4, profiles/r04/DESIGN_round4.md section 5.5) a pass costs 5.3 us on wave slot 0 and 6.2-7.1 us on slot 1, whatever
// s_setprio says, and a batch ends when its slowest instance does.  So an instance that has already run
// `park_min` passes on an unfavoured wave is PARKED at its next outer-iteration boundary (the state there is
// small: u, y, the previous gradient and a dozen scalars) and pushed to a pool; favoured waves look into the
// pool before they take new work from the queue and resume it.  The arithmetic does not change, only where
// an instance runs; the pusher itself falls back to the pool once the queue is exhausted, so nothing is lost.
//
// Teams.  A workgroup is FOUR such waves (one per SIMD of the CU), each with its own LDS slice and its own instances.  A
// batch ends when its slowest instance does, and long before that most waves have run out of work.  A wave without work
// of its own therefore stays and HELPS its siblings: the owner of an instance publishes (u, r, d) of the iteration it
// is starting in its LDS slice, idle waves of the workgroup claim the line-search trials tau = 2^-2 .. 2^-10 in triples
// (the owner itself evaluates u_bar, tau = 1 and tau = 1/2 in the same pass), evaluate psi, grad psi at them out of the
// owner's tables and leave the results in LDS; the owner consumes trials strictly in order, exactly as before, so the
// first accepted one ends the iteration and everything after it is discarded -- same bits, same counters, but an
// iteration whose line search goes deep costs the owner ONE pass instead of up to four.  With fewer instances than a
// quarter of the resident waves the launch gives every instance a whole workgroup from the start (opengen's own call
// pattern, B = 1, src/path_generator.py:385: the latency mode).  Protocol (all in LDS, workgroup scope): claim word per
// owner = (request sequence << 8 | next unclaimed task), taken by compare-and-swap; one done flag per (owner, task,
// helper) written by that helper only, so a late writer of a stale request can never overwrite a current flag.
#pragma once

#ifndef NMPC_CULL_MIN
#define NMPC_CULL_MIN 16
#endif
#ifndef NMPC_WIN
#define NMPC_WIN 1
#endif

namespace nmpc {

typedef __attribute__((address_space(3))) int lds_int;
// control block of a workgroup, after the four slices
enum { CTL_OWNERS = 0, CTL_HELPERS = 1, CTL_CLAIM = 4, CTL_DONE = 8, CTL_INST = 56 };     // done: [owner][task][helper]; inst: [owner]
__device__ __forceinline__ int ctl_load(lds_int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ctl_store(lds_int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ctl_add(lds_int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The instance id of an owner (ctl[CTL_INST + wave]; -1: none) guards what a helper lane has learnt about the owner's tables (cross-track window,
// obstacle certificate): the owner takes the id away BEFORE it rewrites the tables -- for its next instance, or with the result areas once it
// has turned helper itself -- and puts the new one back AFTER; the helper reads the id before and after its evaluation.  A sequence lock: the
// accesses are atomic (ctl_load / ctl_store: the compiler may neither merge the helper's two reads nor drop the owner's first store) and fenced
// at workgroup scope (the id is released after / acquired before the table accesses it guards).  It lives in the control block, not in the
// slice: a slice is overwritten from offset 0 by the result areas of its wave's helper phase.
// a query point's scalar (psi): every lane of the point's row holds it; rows 0..2 are points 0..2
__device__ __forceinline__ double point_scalar(double v, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * k);
    return __hiloint2double(hi, lo);
}

// gradient pair of query point K's evaluation, delivered to the state layout: one read of the transport area (lanes beyond the
// twentieth stage read the point's zero pad; stages N.. of a shorter horizon hold zeros, as the evaluation left them)
#define NMPC_FETCH_GRAD(K, OV, OW)                                             \
    do {                                                                       \
        const dbl2 fg_ = zgr[(K) * ZS];                                        \
        OV = fg_.x; OW = fg_.y;                                                \
    } while (0)

// gradient step x - gamma g and its projection on U (lanes beyond the horizon stay zero)
#define NMPC_HALF_STEP(xv, xw)                                                 \
    do {                                                                       \
        const double s1_ = fma(-gamma, gv, (xv)), s2_ = fma(-gamma, gw, (xw)); \
        hv = ina ? clampd(s1_, vmin, vmax) : s1_;                              \
        hw = ina ? clampd(s2_, -wmax, wmax) : s2_;                             \
    } while (0)
// FBE at the cached point; the gradient step x - gamma g is recomputed (bitwise the same value)
#define NMPC_FBE(xv, xw) fbe_value<P>(cost, gamma, pk_hig, fma(-gamma, gv, (xv)), fma(-gamma, gw, (xw)), hv, hw, gv, gw, lane)

// parked-instance pool: one lane calls these.  Entries are published with release semantics at agent scope and
// read with acquire semantics (other XCDs' L2s are not coherent with ours: plain loads could see stale lines)
// Each pool is a ring of pool_cap >= B slots (an instance waits in at most one slot at a time; -1 = empty) with three counters: positions
// handed to poppers (head) and to pushers (tail), and the number of published entries nobody has claimed yet (count).  Every operation is a
// fetch-add -- NO compare-and-swap loops: with two thousand waves on one counter a CAS loop collapses (measured at B = 32 768: 33 attempts per
// pop, 75 us each, the waves 65 % of their time in here).  A popper first takes one unit of `count` (giving it back if there was none), which
// entitles it to the next head position; the pusher of that position may still be about to publish it -- a few hundred nanoseconds.
constexpr int POOL_CTRS = 4;       // head | tail | count | (pad) per pool; after the pools: instances alive that are known to be long
__device__ __forceinline__ int pool_pop(const KArgs &a, int c)
{
    unsigned int *ctr = a.pool_ctr + POOL_CTRS * c;
    int *cnt = (int *)(ctr + 2);
    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) return -1;
    if (__hip_atomic_fetch_add(cnt, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
    }
    const unsigned pos = __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int *slot = a.pool + (size_t)c * a.pool_cap + pos % (unsigned)a.pool_cap;
    int inst;
    do { inst = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (inst < 0);
    __hip_atomic_store(slot, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return inst;
}
__device__ __forceinline__ void pool_push(const KArgs &a, int c, int inst)
{
    unsigned int *ctr = a.pool_ctr + POOL_CTRS * c;
    const unsigned pos = __hip_atomic_fetch_add(&ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int *slot = a.pool + (size_t)c * a.pool_cap + pos % (unsigned)a.pool_cap;
    while (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) __builtin_amdgcn_s_sleep(1);      // (the previous lap's tenant: taken long ago)
    __hip_atomic_store(slot, inst, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // publishes the parked state too
    __hip_atomic_fetch_add((int *)(ctr + 2), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// instances waiting in pool c (a snapshot: a hint for the scheduling decisions, never for correctness)
__device__ __forceinline__ int pool_depth(const KArgs &a, int c)
{
    return __hip_atomic_load((int *)(a.pool_ctr + POOL_CTRS * c + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- Gram-form L-BFGS (this kernel's; the oracle follows it for N_hor <= 20: oracle/nmpc_oracle.c, lbfgs_apply_gram) ----
// The two-loop recursion chains twenty horizon reductions per iteration, each waiting for the one before.  Here the twenty inner
// products it needs come from ONE batch of independent ones -- lane (c, q) of the wave runs the fma chain of quarter q (five stages)
// of the products of ring pair c with the iteration's y and r, lanes c = 10..12 those among s, y, r, g themselves (which also gives
// ||r||^2, <g, r>, <s, y>, <s, s>, <y, y>: no tree sums at the head of a step); two permlane swaps add the four quarters -- and the
// products among the ring vectors are kept from the iteration each pair entered (gsy, gyy in LDS).  The coefficients alpha_j, beta_j
// then follow from two short recurrences in lanes 0..9 of every row (the coefficient of step j reaches the row by DPP row_newbcast:j),
// and the direction is updated with them as the two-loop recursion would.  Measured (scripts/ubench/mfma_f64.hip): v_mfma_f64_4x4x4
// shares the f64 pipe with the vector ALU (17 cycles = 4 v_fma_f64) and is an exact ascending fma chain over k, but for this shape
// it would compute four times the products the lanes need -- so the batch is plain v_fma_f64.
// acc - x[lane J of the row] * y in ONE instruction (the recurrences' critical path); the two wait states a DPP read of a fresh VALU
// result needs are spelled out, the compiler does not see into the statement
template <int J>
__device__ __forceinline__ double fnma_row_bcast(double acc, double x, double y)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
    return acc;
}
template <int J>
__device__ __forceinline__ double fma_row_bcast(double acc, double x, double y)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
    return acc;
}
// the same broadcast folded into the updates that FOLLOW a recurrence step (no v_mov_b64_dpp, no register for the broadcast value).  They
// carry no wait states of their own: `after` -- the step's own result, which read x through DPP two wait states after x was written -- orders
// them behind it.
template <int J>
__device__ __forceinline__ void fnma3_row_bcast(double &a1, double &a2, double &a3, double x, double y1, double y2, double y3, double after)
{
    asm("v_fmac_f64_dpp %0, -%3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%3, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, -%3, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y1), "v"(y2), "v"(y3), "v"(after), "n"(J));
}
template <int J>
__device__ __forceinline__ void fma2_row_bcast(double &a1, double &a2, double x, double y1, double y2, double after)
{
    asm("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
        : "+v"(a1), "+v"(a2) : "v"(x), "v"(y1), "v"(y2), "v"(after), "n"(J));
}
template <int J>
__device__ __forceinline__ void fnma2_row_bcast(double &a1, double &a2, double x, double y1, double y2, double after)
{
    asm("v_fmac_f64_dpp %0, -%2, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
        : "+v"(a1), "+v"(a2) : "v"(x), "v"(y1), "v"(y2), "v"(after), "n"(J));
}
// (the two-stage kernel's: two stages per lane)
template <int J>
__device__ __forceinline__ void fnma5_row_bcast(double &a1, double &a2, double &a3, double &a4, double &a5, double x, double y1, double y2, double y3,
                                                double y4, double y5, double after)
{
    asm("v_fmac_f64_dpp %0, -%5, %6 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%5, %7 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, -%5, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, -%5, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, -%5, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
        : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5), "v"(after), "n"(J));
}
template <int J>
__device__ __forceinline__ void fma4_row_bcast(double &a1, double &a2, double &a3, double &a4, double x, double y1, double y2, double y3, double y4,
                                               double after)
{
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
        : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(x), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(after), "n"(J));
}
// the value lane `ln` (a constant) holds, as a wave-uniform scalar
__device__ __forceinline__ double lane_scalar(double v, int ln)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), ln);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), ln);
    return __hiloint2double(hi, lo);
}
// The recurrences, software-pipelined.  A step's coefficient (al_J = rho ga1, ab_J = alpha - rho ga2) is read through DPP by the instruction
// that follows it, which needs two wait states after the write; the updates of the direction and of the other recurrence that belong to the
// step BEFORE are independent of it, so they are issued in that gap instead of an s_nop: every accumulator still receives its updates in
// the two-loop recursion's order (same bits), a step is one instruction shorter and its dependent chain one slot.  The compiler does not
// see into the statements: codegen_check.py checks the wait states of every DPP read in the final assembly.
//   first recurrence, step J >= 1: al = rho ga1 | dv, dw -= al_prev[J-1] (yp.x, yp.y)_prev | ga1 -= al[J] gs | ga2 -= al[J] gy
template <int J>
__device__ __forceinline__ double gram_fwd_step(double &ga1, double &ga2, double &dv, double &dw, double rho, double gs, double gy, double alp,
                                                double ypx, double ypy)
{
    double al;
    asm("v_mul_f64 %0, %5, %1\n\t"
        "v_fmac_f64_dpp %3, -%8, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, -%8, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%0, %6 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, -%0, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
        : "=&v"(al), "+v"(ga1), "+v"(ga2), "+v"(dv), "+v"(dw)
        : "v"(rho), "v"(gs), "v"(gy), "v"(alp), "v"(ypx), "v"(ypy), "n"(J), "n"(J - 1));
    return al;
}
// (step 0 has no step before it: the two wait states are an s_nop)
__device__ __forceinline__ double gram_fwd_first(double &ga1, double &ga2, double rho, double gs, double gy)
{
    double al;
    asm("v_mul_f64 %0, %3, %1\n\t"
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %1, -%0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, -%0, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf"
        : "=&v"(al), "+v"(ga1), "+v"(ga2) : "v"(rho), "v"(gs), "v"(gy));
    return al;
}
//   second recurrence, step J <= 8 (the step before it is J + 1): ab = alpha - rho ga2 | dv, dw += ab_prev[J+1] (sp.x, sp.y)_prev | ga2 += ab[J] gr
template <int J>
__device__ __forceinline__ double gram_bwd_step(double &ga2, double &dv, double &dw, double rho, double alv, double gr, double abp, double spx,
                                                double spy)
{
    double ab;
    asm("v_mul_f64 %0, %4, %1\n\t"
        "v_add_f64 %0, %5, -%0\n\t"
        "v_fmac_f64_dpp %2, %7, %8 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %7, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %0, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
        : "=&v"(ab), "+v"(ga2), "+v"(dv), "+v"(dw)
        : "v"(rho), "v"(alv), "v"(gr), "v"(abp), "v"(spx), "v"(spy), "n"(J), "n"(J + 1));
    return ab;
}
// The pair of age J sits in ring slot (H + J) mod 10: with H, the ring's head, a compile-time constant every LDS address of a step is a per-lane
// base plus an immediate -- so the twenty steps exist once per head position (a ten-way switch on the wave-uniform head), 3 scalar and 2
// vector address instructions per step less than with the head in a register: the kernel is bound by instruction issue of ANY class
// (DESIGN.md section 5.6).
#define NMPC_GRAM_LDF(H, J)                                                                    \
        constexpr int fp##J = ((H) + (J)) % MAXMEM;                                            \
        const double fgs##J = Lgsy[pkrow + fp##J], fgy##J = Lgyy[pkrow + fp##J];               \
        const dbl2 fyp##J = LY[fp##J * NS + tt]
#define NMPC_GRAM_FWD(H, J, JP)                                                                \
        NMPC_GRAM_LDF(H, J);                                                                   \
        const double fal##J = gram_fwd_step<(J)>(ga1, ga2, dv, dw, rho_k, fgs##J, fgy##J, fal##JP, fyp##JP.x, fyp##JP.y)
#define NMPC_GRAM_LDB(H, J)                                                                    \
        constexpr int bp##J = ((H) + (J)) % MAXMEM;                                            \
        const double bgr##J = Lgsy[bp##J * GRAM_LD + pk_];                                     \
        const dbl2 bsp##J = LS[bp##J * NS + tt]
#define NMPC_GRAM_BWD(H, J, JP)                                                                \
        NMPC_GRAM_LDB(H, J);                                                                   \
        const double bab##J = gram_bwd_step<(J)>(ga2, dv, dw, rho_k, alv, bgr##J, bab##JP, bsp##JP.x, bsp##JP.y)
// both recurrences for the head position H
#define NMPC_GRAM_BOTH(H)                                                                                                            \
    do {                                                                                                                             \
        NMPC_GRAM_LDF(H, 0);                                                                                                         \
        const double fal0 = gram_fwd_first(ga1, ga2, rho_k, fgs0, fgy0);                                                             \
        NMPC_GRAM_FWD(H, 1, 0); NMPC_GRAM_FWD(H, 2, 1); NMPC_GRAM_FWD(H, 3, 2); NMPC_GRAM_FWD(H, 4, 3); NMPC_GRAM_FWD(H, 5, 4);      \
        NMPC_GRAM_FWD(H, 6, 5); NMPC_GRAM_FWD(H, 7, 6); NMPC_GRAM_FWD(H, 8, 7); NMPC_GRAM_FWD(H, 9, 8);                              \
        fnma2_row_bcast<9>(dv, dw, fal9, fyp9.x, fyp9.y, ga1);                                                                      \
        const double alv = rho_k * ga1;          /* alpha_k in lane k: entry k of ga1 is final once step k has used it */              \
        ga2 = n_H0 * ga2;                                                                                                            \
        dv = n_H0 * dv; dw = n_H0 * dw;                                                                                              \
        NMPC_GRAM_LDB(H, 9);                                                                                                         \
        const double bab9 = alv - rho_k * ga2;                                                                                       \
        ga2 = fma_row_bcast<9>(ga2, bab9, bgr9);                                                                                     \
        NMPC_GRAM_BWD(H, 8, 9); NMPC_GRAM_BWD(H, 7, 8); NMPC_GRAM_BWD(H, 6, 7); NMPC_GRAM_BWD(H, 5, 6); NMPC_GRAM_BWD(H, 4, 5);      \
        NMPC_GRAM_BWD(H, 3, 4); NMPC_GRAM_BWD(H, 2, 3); NMPC_GRAM_BWD(H, 1, 2); NMPC_GRAM_BWD(H, 0, 1);                              \
        fma2_row_bcast<0>(dv, dw, bab0, bsp0.x, bsp0.y, ga2);                                                                        \
    } while (0)

// The state machine's wave-uniform flags are bits of ONE 32-bit scalar.  As `bool`s each is a 64-bit lane mask (two scalar registers,
// tested through s_and_b64 with exec): seventeen of them across the loop body are a third of the scalar file, and what does not fit is
// spilled to vector lanes, every reload a vector instruction.
struct FlagBit {
    unsigned &w;
    const unsigned bit;
    __device__ __forceinline__ operator bool() const { return (w & bit) != 0u; }
    __device__ __forceinline__ FlagBit &operator=(bool v) { w = v ? (w | bit) : (w & ~bit); return *this; }
    __device__ __forceinline__ FlagBit &operator=(const FlagBit &o) { return *this = (bool)o; }
};

template <class SH>
__global__ __launch_bounds__(64 * TEAM_WAVES, 2) void nmpc_solve_hyb_kernel(KArgs a)
{
    static_assert(MAXMEM == 10, "the Gram-form recurrences are written out for ten pairs");
    constexpr int PE = 20;                      // evaluation layout: three lane groups (nmpc_device.h)
    constexpr int P = 32, COLS = 24;            // state layout: stage t at lane t of both 32-lane halves (lanes 24..31 share LDS column 23: zeros)
    extern __shared__ double lds[];
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave of the team
    const int slice = the_map<SH, PE>(a).total;
    lds_double *L = (lds_double *)lds + wid * slice;
    lds_int *ctl = (lds_int *)((lds_double *)lds + TEAM_WAVES * slice);
    const int lane = threadIdx.x & 63, h = lane >> 5, t = lane & 31;
    const int q = lay_group<PE>(lane), te = lay_stage<PE>(lane);
    // (launch-uniform integers the loop reads -- each a scalar of its own, not a member of the argument block's eight-dword load: scalar_own)
    const int N = shape_N<SH>(a), m = scalar_own(a.op.lbfgs_memory);
    const int k_akkt = scalar_own(a.op.akkt_gradient), k_lsf = scalar_own(a.op.ls_failure);
#ifdef NMPC_EXPERIMENTS
    const int k_help = scalar_own(a.team_help), k_dbg = scalar_own(a.dbg);      // knobs of the experiments build (NMPC_TEAM_HELP, NMPC_DEBUG_PRIO)
#else
    constexpr int k_help = 1, k_dbg = 0;                                          // (the shipped library's host side never sets them otherwise)
#endif
    const bool in = t < N, ina = in;            // state layout: lanes beyond the horizon hold zeros
    const bool ine = te < N;                    // evaluation layout: a real stage
    // with a 20-stage horizon every stage lane of the evaluation layout is inside it (lanes 60..63 may
    // hold don't-care values: no cross-lane operation lets them into other lanes)
    constexpr bool FULL = SH::N == PE;
    constexpr bool CULL = SH::NOBS > NMPC_CULL_MIN || SH::NOBS < 0;      // many circle slots: scan only those the robot can reach (eval_psi)
    constexpr int WIN = NMPC_WIN;                             // windowed cross-track search: half width in segments (eval_psi)
    constexpr bool OBSC = SH::N > 0;                          // obstacle certificate (eval_psi): the shape-specialised kernels only -- the run-time-shape kernel has no registers left for it
    const bool inea = FULL ? true : ine;
    const LdsMap mp = the_map<SH, PE>(a);
    const int n2 = shape_nobs<SH>(a) + shape_ndyn<SH>(a);
    const int f2off = mp.f2 + q * (n2 + 1);
    // Transport area: [3 points][ZS = 24 slots] (v, w) pairs + 8 slots nobody reads = the 80 pairs of the Gram batch's s | y | r | g, which
    // are dead from the end of the batch to the next step.  Slot (k, j) holds stage j of point k; slots 20..23 of a point are ZERO PADS:
    // state lanes 20..23 write their (zero) entries there with the query points, and nothing else ever does -- evaluation lanes 60..63
    // (stages 20..23 of point 2) and state lanes 24..31 write to the spare slots.  Every pointer below is a per-lane constant.
    constexpr int ZS = 24, src0 = 0, src1 = 1, src2 = 2;
    static_assert(3 * ZS + 8 == 4 * GRAM_NST, "the transport area is the place of the Gram batch's four vectors");
    lds_double2 *Lz = (lds_double2 *)(L + mp.nv);
    lds_double2 *zwX = Lz + (t < ZS ? h * ZS + t : 3 * ZS + (t - ZS));          // state lane: its X entry -> point h (X of half 0 | half 1)
    lds_double2 *zwY = Lz + (t < ZS ? 2 * ZS + t : 3 * ZS + (t - ZS));          // ... its Y entry -> point 2
    const lds_double2 *zrd = Lz + q * ZS + te;                                   // evaluation lane: its stage of its point (lanes 60..63: zero pads)
    // ... the stage before it: the last input (prepare_instance leaves it as a pair behind the instance scalars) for stage 0
    const lds_double2 *zpv = te == 0 ? (const lds_double2 *)(L + mp.sc + 18) : (lane < 60 ? zrd - 1 : zrd);
    lds_double2 *zmine = Lz + (lane < 60 ? q * ZS + te : 3 * ZS + (lane - 60));   // evaluation lane: where it leaves (qa, qw), then its gradient pair
    // ... and the slot of the stage after it, as an address of its own: the compiler must not prove the read independent of the write
    const lds_double2 *znext = Lz + opaque_i(lane < 60 ? q * ZS + te + 1 : 3 * ZS + (lane - 60) + 1);
    const lds_double2 *zgr = Lz + (t < 20 ? t : 20);                             // state lane: gradient of point k at zgr[k * ZS]
    // L-BFGS ring: GRAM_NST + 1 columns per slot, the last one all zeros -- lanes beyond GRAM_NST read it (stages N.. are zeros too)
    constexpr int NS = GRAM_NST + 1;
    const int tt = t < GRAM_NST ? t : GRAM_NST;
    lds_double2 *LS = (lds_double2 *)(L + mp.S);
    lds_double2 *LY = (lds_double2 *)(L + mp.Y);
    lds_double *Lrho = L + mp.rho;
    lds_double2 *Lnv = (lds_double2 *)(L + mp.nv);            // Gram-form L-BFGS: s | y | r | g of the iteration, by stage
    lds_double *Lgsy = L + mp.gsy, *Lgyy = L + mp.gyy;        // ... and the kept inner products [slot][slot]
    const int c16 = lane & 15, q4 = lane >> 4;                // ... batch lane = (ring pair / age, quarter of the horizon)
    lds_double2 *Los = (lds_double2 *)(L + mp.vec) + (t < COLS ? t : COLS - 1);      // parked pairs, one column per stage
    lds_double2 *Log = Los + COLS, *Lq = Los + 2 * COLS, *Lyp = Los + 3 * COLS;
    lds_double2 *Lgk = Los + 6 * COLS;                          // gradient at the current iterate (opts.ls_failure = 1 only)
    lds_double2 *LypE = (lds_double2 *)(L + mp.vec) + 3 * COLS + te;      // the same columns, by evaluation lane
    lds_double2 *Ly = (lds_double2 *)(L + mp.vec) + 4 * COLS + te;        // multipliers y (read by every evaluation)
    lds_double *Lvr = L + mp.vec + 2 * 5 * COLS + te;                     // reference speed of this stage
    // an empty L-BFGS buffer is all zeros: the Gram form runs over all ten ages every time (gsy | gyy | S | Y are contiguous)
#define NMPC_LB_ZERO()                                                                                  \
    do {                                                                                                \
        lds_double2 *z_ = (lds_double2 *)(L + mp.gsy);                                                  \
        for (int i_ = lane; i_ < MAXMEM * GRAM_LD + 2 * MAXMEM * NS; i_ += 64) z_[i_] = dbl2{0.0, 0.0};  \
        if (lane < MAXMEM) Lrho[lane] = 0.0;                                                            \
    } while (0)
    if (threadIdx.x < TEAM_CTL_INTS) ctl[threadIdx.x] = threadIdx.x == CTL_OWNERS ? a.team_owners : 0;
    __syncthreads();
    lds_double2 *Lreq = (lds_double2 *)(L + mp.req);      // this wave's request: u | r | d by stage
    unsigned team_seq = 0;                                // sequence number of this wave's requests
    const double vmin = scalar_own(a.pb.vmin), vmax = scalar_own(a.pb.vmax), wmax = scalar_own(a.pb.wmax);
    const EvK ek = {scalar_own(a.pb.ts), scalar_own(a.inv_ts), scalar_own(a.pb.amin), scalar_own(a.pb.amax), scalar_own(a.pb.awmax)};
    const double tol_ = scalar_own(a.op.tolerance);
    const unsigned max_inner = (unsigned)scalar_own(a.op.max_inner);
    const unsigned budget = (unsigned)scalar_own(a.op.max_total_inner);     // 0 = off
    lds_double *Lpar = L + mp.par;
#define pk_eps_nu Lpar[0]
#define pk_dy_norm Lpar[1]
#define pk_f2_norm Lpar[2]
#define pk_dy_norm_plus Lpar[3]
#define pk_f2_norm_plus Lpar[4]
#define pk_last_fpr Lpar[5]
#define pk_last_cost Lpar[6]
#define pk_norm_h Lpar[7]
#define pk_Lc Lpar[8]
#define pk_H0 Lpar[9]
#define pk_sigma Lpar[10]
#define pk_c_lip Lpar[11]
#define pk_fbe_u Lpar[20]       /* FBE at the current iterate */
#define pk_hig Lpar[19]         /* 0.5 / gamma, refreshed when gamma changes (the envelope's last term is dist2 * pk_hig); helpers read it with the request */
#define pk_gr Lpar[12]          /* <grad psi, r> of the current iterate: summed together with ||r||^2, used by the Lipschitz test */
    /* Lpar[13], Lpar[14]: first start (100 MHz clock) and migration count; Lpar[15..17]: c, 1 / max(c, 1) and gamma of the request */


    // wave slot within the SIMD (HW_ID[3:0]): with two resident waves the hardware favours slot 0
    const unsigned hw_slot = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 4);
    const bool unfavoured = hw_slot != 0u;
    const int PS = park_stride(N);

    // first round: the queue's head -- the instances that look hardest -- goes to the favoured wave slots; the other
    // waves hold back until half of the resident waves have fetched.  Bounded by the constant 100 MHz clock (40 us),
    // and skipped when the queue holds barely more than one round (nothing to gain from ordering the first fetches)
    const int n_waves = (int)gridDim.x * TEAM_WAVES;
    if (a.order && unfavoured && a.park_min > 0 && a.B >= n_waves + (n_waves >> 2)) {
        const unsigned want = (unsigned)n_waves / 2;
        const long long t_hold = (long long)__builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want &&
               (long long)__builtin_amdgcn_s_memrealtime() - t_hold < 4000)
            __builtin_amdgcn_s_sleep(8);
    }

    // A wave is first the OWNER of the instances it takes from the queue (the loop right below); once there is nothing left
    // for it -- or from the start, for the waves beyond team_owners in the small-batch mode -- it is a HELPER of its siblings
    // (the loop at the end) until the last of them has finished.
    // (Measured and not taken, round 5: helpers for the K longest instances of a full batch BEFORE the queue is dry -- a sibling that comes
    // back for work helps a flagged long instance instead of fetching.  Same bits; cfg 1 -0.1 .. +0.6 %, cfg 3 -1.4 % for K = 16 .. 256:
    // with the step-aside scheduling the long instances time-share the waves, so the end of a batch is their throughput, not one
    // instance's latency.  profiles/r05/topk_helpers.jsonl, topk_helpers.patch.)
    for (; wid < a.team_owners;) {
        // ------------------------------------------------------------------ next instance: parked long-runners first
        // (favoured waves), else the queue, else -- once the queue is exhausted -- whatever is still parked
        int fetched = -1, from_pool = 0;
        if (lane == 0) {
            const bool pools = a.park_min > 0 || a.sched_mode > 0;
            if (pools && !unfavoured) { fetched = pool_pop(a, POOL_LONG); from_pool = fetched >= 0; }      // favoured waves: waiting long-runners first
            if (fetched < 0) {
                const unsigned nxt = atomicAdd(a.queue, 1u);
                if (nxt < (unsigned)a.B) fetched = a.order ? a.order[nxt] : (int)nxt;
                else if (pools) {
                    for (int c = 0; c < NPOOLS && fetched < 0; ++c) fetched = pool_pop(a, c);
                    from_pool = fetched >= 0;
                }
            }
        }
        const int inst = __builtin_amdgcn_readfirstlane(fetched);
        const bool resumed = __builtin_amdgcn_readfirstlane(from_pool) != 0;
        if (inst < 0) break;     // nothing left for this wave: a helper from now on

        const long long dbg_t0 = __builtin_amdgcn_s_memtime();
        // (experiments, NMPC_DEBUG_PRIO: first start and migration count travel with the instance; parked in LDS)
        Lpar[13] = (double)__builtin_amdgcn_s_memrealtime(); Lpar[14] = 0.0;
        if (k_dbg == 1) { if (hw_slot & 1u) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
        double vref_;
        DynStage dyn;
        // (the instance id goes away BEFORE the tables change and comes back after: a helper still evaluating a cancelled request of the previous
        // instance then finds another id after its evaluation than before it and throws away what it learnt about its window)
        if (lane == 0) ctl_store(ctl + CTL_INST + wid, -1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        prepare_instance<PE, SH>(a, L, a.p + (size_t)inst * a.n_p, te, vref_, dyn);
        *Lvr = vref_;
        WinState ws = {te < N - 1 ? te : N - 2, 0.0, 0.0, 0.0};      // this lane's cross-track window (eval_psi): nothing known yet
        ObsCert oc = {0.0, 0.0, 0.0, 0, 0, 0};                      // ... and its obstacle certificate
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) ctl_store(ctl + CTL_INST + wid, inst);         // (helpers tell by it whether their own windows are still this instance's)
        unsigned long long near = ~0ull;            // static circles worth scanning (eval_psi, CULL)
        if constexpr (CULL) {
            near = circle_near_mask(a.p + (size_t)inst * a.n_p, N, shape_nobs<SH>(a), lane, a.cull_radius);
            if (lane == 0) { Lpar[18] = __hiloint2double((int)(near >> 32), (int)near); }
        }

        // horizon vectors: lane t holds the (v_t, w_t) pair, identically in both halves unless noted
        // a fresh instance starts from the caller's u0 / y0; a resumed one from its parked state (acquired by pool_pop)
        const double *pk = a.park + (size_t)inst * PS;
        const double *u0 = resumed ? pk : a.u + (size_t)inst * a.n_u;
        double uv = in ? u0[2 * t] : 0.0, uw = in ? u0[2 * t + 1] : 0.0;
        {
            double yv0, yw0;
            if (resumed) {
                yv0 = ine ? pk[2 * N + 2 * te] : 0.0;
                yw0 = ine ? pk[2 * N + 2 * te + 1] : 0.0;
            } else {
                yv0 = (a.y0 && ine) ? a.y0[(size_t)inst * a.n1 + te] : 0.0;
                yw0 = (a.y0 && ine) ? a.y0[(size_t)inst * a.n1 + N + te] : 0.0;
            }
            *LypE = dbl2{yv0, yw0};
            *Ly = dbl2{yv0, yw0};
        }
        // gradient_u_previous (AKKT residual): zero at the start of a solve, carried across its inner solves
        *Lq = (resumed && in) ? dbl2{pk[4 * N + 2 * t], pk[4 * N + 2 * t + 1]} : dbl2{0.0, 0.0};
        double gv = 0, gw = 0, hv = 0, hw = 0, rv = 0, rw = 0, dv = 0, dw = 0;
        // (the query points of a pass -- X of this half -> evaluation points 0 and 1, Y -> point 2 -- and the trial point being consumed live
        // inside one pass: they are declared in the loop body, so that they do not occupy registers from one pass to the next)
        unsigned fl = 0u;                         // the flags below (FlagBit)
        FlagBit need_grad{fl, 1u << 0}; need_grad = true;
        double cost = 0, gamma = 0, tau = 1, rhs_ls = 0;
        pk_fbe_u = 0.0; pk_hig = 0.0;
        pk_Lc = 0.0; pk_sigma = 0.0; pk_H0 = 1.0;
        FlagBit fbe_ok{fl, 1u << 1};                      // pk_fbe_u holds the FBE at the current iterate (an accepted trial's FBE is the next iteration's: same operands, same bits)
        int iteration = 0, lip_it = 0, ls_n = 0, lb_active = 0, lb_head = 0;
        FlagBit lb_first{fl, 1u << 2}; lb_first = true;
        // tentative L-BFGS update of the current iteration (committed when the Lipschitz test passes)
        int n_active = 0, n_head = 0;
        FlagBit n_first{fl, 1u << 3}, n_take_old{fl, 1u << 4}; n_first = true;
        double n_H0 = 1;
        unsigned num_iter = 0;
        const double c0 = a.c0 ? a.c0[inst] : 0.0;
        double pen_c = c0 > 0.0 ? c0 : a.op.initial_penalty;
        double cbar_inv = 1.0 / fmax(pen_c, 1.0);          // 1 / max(c, 1), refreshed when c changes
        pk_c_lip = 0.0;                                     // 0.95 / (2 gamma), refreshed when gamma changes
        // scalars touched once per inner solve / outer iteration are parked in LDS (every lane writes the
        // same value) instead of occupying a VGPR pair each for the whole solve
        pk_eps_nu = a.op.initial_tolerance;
        pk_dy_norm = 0.0; pk_f2_norm = 0.0; pk_dy_norm_plus = DBL_MAX; pk_f2_norm_plus = 0.0; pk_last_fpr = 0.0; pk_last_cost = 0.0; pk_norm_h = 0.0;
        int nu = 0, inner_status = 0, state = D_INIT, final_status = 0;
        unsigned inner_total = 0, n_cost = 0, n_grad = 0, n_pass = 0;
        FlagBit parked{fl, 1u << 5};
        int park_cls = POOL_LONG;
        unsigned q_pass = 0;                      // n_pass at the last outer-iteration boundary (or at the start of this leg)
        FlagBit long_counted{fl, 1u << 6};                 // this instance is in the count of long instances alive (KArgs.pool_ctr[2 NPOOLS])
        if (resumed) {                            // parked scalars
            const double *pks = a.park + (size_t)inst * PS + 6 * N;
            pen_c = pks[0];
            cbar_inv = 1.0 / fmax(pen_c, 1.0);
            pk_eps_nu = pks[1]; pk_dy_norm = pks[2]; pk_f2_norm = pks[3]; pk_dy_norm_plus = pks[4]; pk_f2_norm_plus = pks[5];
            pk_last_fpr = pks[6]; pk_last_cost = pks[7];
            nu = (int)pks[8]; inner_total = (unsigned)pks[9]; n_cost = (unsigned)pks[10]; n_grad = (unsigned)pks[11]; n_pass = (unsigned)pks[12];
            Lpar[13] = pks[13]; Lpar[14] = pks[14] + 1.0;
            long_counted = pks[15] != 0.0; q_pass = n_pass;
        }

        // phase flags (wave-uniform): set by the state handlers, consumed at the top of the loop
        FlagBit f_start{fl, 1u << 7}, f_back{fl, 1u << 8}, f_trials{fl, 1u << 9}, f_begin{fl, 1u << 11}, f_done{fl, 1u << 12}, f_fb{fl, 1u << 13};
        FlagBit running{fl, 1u << 14}, timed_out{fl, 1u << 15};
        f_start = true; running = true;
        FlagBit posted{fl, 1u << 16};                      // a request of the current iteration is open for the helpers
#ifdef NMPC_TL
        int tl_it = 0;                            // PANOC steps of this instance so far
#endif
        // -DNMPC_PROF2 (scripts/sections.py): cycles of this instance by section of the loop -- 0 phase handlers in front of the batch, 1 the batch
        // of inner products, 2 exit test / L-BFGS update, 3 the recurrences and the direction, 4 envelope, trial points, request, 5 the
        // evaluation, 6 the consumption of the trials.  Every mark drains the LDS queue, so the sum is a little above the plain build's time.
#ifdef NMPC_PROF2
        long long pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0, pf4 = 0, pf5 = 0, pf6 = 0, pf_last;
        long long pe[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // NMPC_PROF2 == 2: the evaluation's own sections (eval_psi's NMPC_EVTICK marks), pe[7] = last mark
#define NMPC_SEC_RAW(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)
#define NMPC_SEC(acc) do { long long t_; NMPC_SEC_RAW(t_); acc += t_ - pf_last; pf_last = t_; } while (0)
        NMPC_SEC_RAW(pf_last);
#elif defined(NMPC_MARKS)      // section markers in the ISA dump (scripts/isa_stats.py)
#define NMPC_SEC(acc) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK S_" #acc); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NMPC_SEC(acc) do { } while (0)
#endif

        // an iteration finished (raised by the consumption of a pass; the next pass starts the next step or returns from the inner solver).
        // Written out where it is raised: as a flag it was a test at the top of every pass and a set / clear pair per iteration.
        // OpEn: while step() && num_iter < max_iter { num_iter++ };  opts.max_total_inner: the deterministic max_duration
#define NMPC_END_ITERATION()                                                                                   \
        do {                                                                                                   \
            iteration++;                                                                                       \
            if (!(num_iter < max_inner)) f_done = true;                                                        \
            else {                                                                                             \
                num_iter++;                                                                                    \
                if (budget > 0u && inner_total + num_iter >= budget) { timed_out = true; f_done = true; }      \
                else f_begin = true;                                                                           \
            }                                                                                                  \
        } while (0)
        for (;;) {
            double xv = 0, xw = 0, yqv = 0, yqw = 0;       // query points of this pass: a phase handler below sets them
            double nr2 = 0, norm_r = 0;                    // ||r||^2, ||r|| of the step this pass starts (the batch below); the last ||r|| stays in pk_last_fpr
            double pv = 0, pw = 0;                         // line-search trial point being consumed (after the evaluation)
            // (Measured and not taken, round 6: these without their initialisers -- each is written before it is read on every path, and the
            // zeros are fifteen moves per pass in front of the phase handlers.  Without those of xv..yqw or of nr2, norm_r ROCm 7.2's
            // register allocator segfaults or spills to scratch; those of pv, pw, lhs the compiler drops by itself.)
            bool lb_batch = false;                     // this pass starts with the batch of inner products (f_back, f_begin)
            // (the two rare phases behind ONE test: a pass of the bulk looks at one flag word instead of two flags)
            if (fl & (f_back.bit | f_fb.bit)) {
                // ---------------------------------------------------------------- backtrack: L <- 2L, gamma <- gamma/2
                if (f_back) {
                    if (posted) { posted = false; if (lane == 0) ctl_store(ctl + CTL_CLAIM + wid, 0); }      // speculation discarded
                    lb_active = 0; lb_first = true;                         // L-BFGS buffer invalidated
                    NMPC_LB_ZERO();
                    fbe_ok = false;
                    pk_Lc *= 2.0; gamma /= 2.0;
                    pk_sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * gamma);
                    pk_c_lip = GAMMA_L_COEFF / (2.0 * gamma);
                    pk_hig = 0.5 / gamma;
                    NMPC_HALF_STEP(uv, uw);
                    lb_batch = true;
                }
                // ---------------------------------------------------------------- every trial failed (opts.ls_failure = 1):
                // tau = 0, the forward-backward step from the current iterate
                if (f_fb) {
                    f_fb = false;
                    tau = 0.0;
                    { const dbl2 gk_ = *Lgk; gv = gk_.x; gw = gk_.y; }
                    NMPC_HALF_STEP(uv, uw);
                    xv = yqv = hv; xw = yqw = hw; need_grad = true; state = D_FB;
                }
            }
            // ---------------------------------------------------------------- line-search trials (tau, ls_n) | (tau/2, ls_n+1) | (tau/4, ls_n+2)
            if (f_trials) {
                f_trials = false;
                const double th_ = h == 0 ? tau : tau / 2.0, omt = 1.0 - th_;
                xv = fma(-th_, dv, fma(-omt, rv, uv));
                xw = fma(-th_, dw, fma(-omt, rw, uw));
                const double t4_ = tau / 4.0, om4_ = 1.0 - t4_;
                yqv = fma(-t4_, dv, fma(-om4_, rv, uv));
                yqw = fma(-t4_, dw, fma(-om4_, rw, uw));
                need_grad = true; state = D_LS;
            }
            // ---------------------------------------------------------------- start of a PANOC step
            if (f_begin) lb_batch = true;
#ifdef NMPC_TL
            if (f_begin) { tl_it++; NMPC_TL_EV(tl_it, 0); }
#endif
            NMPC_SEC(pf0);
            // ---- the batch of inner products of this step (Gram-form L-BFGS, see the top of the file)
            double gU = 0.0, gs1 = 0.0, gs2 = 0.0, gy1 = 0.0, gy2 = 0.0;
            if (lb_batch) {
                rv = uv - hv; rw = uw - hw;                // the residual of the step (a back-off has just halved gamma and renewed the half step)
                if (f_begin && iteration >= 1 && !lb_first) {
                    const dbl2 os_ = *Los, og_ = *Log;
                    gs1 = uv - os_.x; gs2 = uw - os_.y; gy1 = rv - og_.x; gy2 = rw - og_.y;
                }
                if (t < GRAM_NST && h == 0) {
                    Lnv[t] = dbl2{gs1, gs2}; Lnv[GRAM_NST + t] = dbl2{gy1, gy2};
                    Lnv[2 * GRAM_NST + t] = dbl2{rv, rw}; Lnv[3 * GRAM_NST + t] = dbl2{gv, gw};
                }
                // lane (c, q): X = (S_c, Y_c), Z = (y, r) for the ring pairs c < 10; (s, y) x (s, y) for c = 10; (r, g) x (r, r) for
                // c = 11; (s, y) x (r, r) for c >= 12.  V1 = X1.Z1, V2 = X1.Z2, V3 = X2.Z1, V4 = X2.Z2 over the stages of quarter q.
                const lds_double2 *X1 = (c16 < MAXMEM ? LS + c16 * NS : (c16 == 11 ? Lnv + 2 * GRAM_NST : Lnv)) + 5 * q4;
                const lds_double2 *X2 = (c16 < MAXMEM ? LY + c16 * NS : (c16 == 11 ? Lnv + 3 * GRAM_NST : Lnv + GRAM_NST)) + 5 * q4;
                const lds_double2 *Z1 = (c16 < MAXMEM ? Lnv + GRAM_NST : (c16 == 10 ? Lnv : Lnv + 2 * GRAM_NST)) + 5 * q4;
                const lds_double2 *Z2 = (c16 < MAXMEM ? Lnv + 2 * GRAM_NST : (c16 == 10 ? Lnv + GRAM_NST : Lnv + 2 * GRAM_NST)) + 5 * q4;
                double V1 = 0.0, V2 = 0.0, V3 = 0.0, V4 = 0.0;
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    const dbl2 x1 = X1[e], x2 = X2[e], z1 = Z1[e], z2 = Z2[e];
                    V1 = fma(x1.x, z1.x, V1); V1 = fma(x1.y, z1.y, V1);
                    V2 = fma(x1.x, z2.x, V2); V2 = fma(x1.y, z2.y, V2);
                    V3 = fma(x2.x, z1.x, V3); V3 = fma(x2.y, z1.y, V3);
                    V4 = fma(x2.x, z2.x, V4); V4 = fma(x2.y, z2.y, V4);
                }
                // (q0 + q1) + (q2 + q3) of all four at once: row i of gU ends up with V(i + 1) summed over the quarters
                swap_rows(V1, V2);
                const double W12 = V1 + V2;
                swap_rows(V3, V4);
                double W34 = V3 + V4;
                double W12b = W12;
                swap_halves(W12b, W34);
                gU = W12b + W34;
                nr2 = lane_scalar(gU, 11);                       // <r, r>
                pk_gr = lane_scalar(gU, 32 + 11);                // <g, r>
                norm_r = sqrt(nr2);
                pk_last_fpr = norm_r;
#ifdef NMPC_PROF2
                { double keep = norm_r + gU; asm volatile("" : "+v"(keep)); }
#endif
            }
            NMPC_SEC(pf1);
#ifdef NMPC_TL
            if (lb_batch) { NMPC_TL_KEEP(norm_r + gU); NMPC_TL_EV(tl_it, 1); }
#endif
            if (f_begin) {
                f_begin = false;
                bool exit_now = false;
                if (__any(norm_r < tol_)) {                    // fpr test, then the AKKT test (opts.akkt_gradient)
                    if (k_akkt == 2) exit_now = true;
                    else if (k_akkt == 1 && iteration >= 1) {
                        // grad_prev was copied from grad at the top of this step: the residual r / gamma + grad - grad_prev is r / gamma
                        exit_now = __any(norm_r < pk_eps_nu * gamma);
                    } else {
                        const dbl2 q_ = *Lq;
                        const bool top = k_akkt == 1;       // iteration 0: grad_prev is still the zero vector
                        const double b1 = top ? gv : gv - q_.x;
                        const double b2 = top ? gw : gw - q_.y;
                        const double a1 = rv / gamma + b1, a2 = rw / gamma + b2;
                        exit_now = __any(sqrt(group_sum<P>(fma(a1, a1, a2 * a2), lane)) < pk_eps_nu);
                    }
                }
                if (exit_now) {
                    f_done = true;
                } else if (iteration == 0) {
                    // psi and grad psi at u_bar serve both the Lipschitz test and the first FB step
                    lip_it = 0;
                    xv = yqv = hv; xw = yqw = hw; need_grad = true; state = D_LIP;
                } else {
                    lip_it = 0;
                    // ---- tentative L-BFGS update with (s, y) = (u - u_old, r - r_old) ----
                    n_first = lb_first; n_head = lb_head; n_active = lb_active; n_H0 = pk_H0; n_take_old = false;
                    bool took = false;                  // the pair of this step entered the buffer (as its newest: age 0)
                    if (lb_first) {
                        n_first = false; n_take_old = true;
                    } else {
                        const double ss = lane_scalar(gU, 10), ys = lane_scalar(gU, 16 + 10);
                        bool ok = !(ss <= DBL_MIN || ys <= LBFGS_SY_EPSILON);
                        if (ok) ok = ys > (LBFGS_CBFGS_EPSILON * norm_r) * ss;      // C-BFGS: <y, s> / ||s||^2 > eps ||r||
                        if (__any(ok)) {
                            took = true;
                            n_take_old = true;
                            // the ring always turns over its ten slots: the pair of age k sits in slot (head + k) mod 10
                            n_head = lb_head == 0 ? MAXMEM - 1 : lb_head - 1;
                            if (in && h == 0) { LS[n_head * NS + t] = dbl2{gs1, gs2}; LY[n_head * NS + t] = dbl2{gy1, gy2}; }
                            if (lane == 0) Lrho[n_head] = 1.0 / ys;
                            const double yy = lane_scalar(gU, 48 + 10);
                            n_H0 = ys / yy;
                            if (n_active < m) n_active++;
                            // kept inner products of the new pair (slot n_head) with the pairs that stay: column n_head of gsy
                            // (<s_c, y>, row 0 of gU), row and column of gyy (<y_c, y>, row 2); the new pair is the newest, so its
                            // own row of gsy is zero (gsy is strictly lower triangular by age), and so is the diagonal
                            if (c16 < MAXMEM && q4 < 3) {
                                const bool dg = c16 == n_head;
                                lds_double *wa = q4 == 0 ? Lgsy + c16 * GRAM_LD + n_head : (q4 == 1 ? Lgsy + n_head * GRAM_LD + c16 : Lgyy + c16 * GRAM_LD + n_head);
                                const double wv = q4 == 1 ? 0.0 : (dg ? (q4 == 0 ? 0.0 : yy) : gU);
                                *wa = wv;
                                if (q4 == 2) Lgyy[n_head * GRAM_LD + c16] = wv;
                            }
                            if (m < MAXMEM) {
                                // a shorter memory (opts.lbfgs_memory < 10): the pair that has just reached age m leaves -- its slot goes back
                                // to zeros, as every slot that holds no pair is
                                const int ev = n_head + m >= MAXMEM ? n_head + m - MAXMEM : n_head + m;
                                if (t <= GRAM_NST && h == 0) { LS[ev * NS + t] = dbl2{0.0, 0.0}; LY[ev * NS + t] = dbl2{0.0, 0.0}; }
                                if (lane == 0) Lrho[ev] = 0.0;
                                if (c16 < MAXMEM && q4 < 2) {
                                    (q4 == 0 ? Lgsy : Lgyy)[c16 * GRAM_LD + ev] = 0.0;
                                    (q4 == 0 ? Lgsy : Lgyy)[ev * GRAM_LD + c16] = 0.0;
                                }
                            }
                        }
                    }
                    // ---- d = H r over the tentative buffer ----
                    dv = rv; dw = rw;
#ifdef NMPC_TL
                    NMPC_TL_EV(tl_it, 2);
#endif
                    NMPC_SEC(pf2);
                    if (n_active > 0) {
                        // age k = lane & 15 of every row (lanes 10..15 idle along on slot 9; what they compute is never looked at)
                        const int pk_ = c16 < MAXMEM ? (n_head + c16 >= MAXMEM ? n_head + c16 - MAXMEM : n_head + c16) : MAXMEM - 1;
                        const int pkrow = pk_ * GRAM_LD;
                        double ga1 = lane_get(gU, 16 + pk_), ga2 = lane_get(gU, 48 + pk_);      // <s_k, r>, <y_k, r>
                        if (took && c16 == 0) { ga1 = lane_scalar(gU, 12); ga2 = lane_scalar(gU, 32 + 12); }      // (the ring held the evicted pair)
                        const double rho_k = Lrho[pk_];
                        switch (n_head) {
                        case 0: NMPC_GRAM_BOTH(0); break;
                        case 1: NMPC_GRAM_BOTH(1); break;
                        case 2: NMPC_GRAM_BOTH(2); break;
                        case 3: NMPC_GRAM_BOTH(3); break;
                        case 4: NMPC_GRAM_BOTH(4); break;
                        case 5: NMPC_GRAM_BOTH(5); break;
                        case 6: NMPC_GRAM_BOTH(6); break;
                        case 7: NMPC_GRAM_BOTH(7); break;
                        case 8: NMPC_GRAM_BOTH(8); break;
                        default: NMPC_GRAM_BOTH(9); break;
                        }
                    }
#ifdef NMPC_PROF2
                    { double keep = dv + dw; asm volatile("" : "+v"(keep)); }
#endif
#ifdef NMPC_TL
                    NMPC_TL_KEEP(dv + dw); NMPC_TL_EV(tl_it, 3);
#endif
                    NMPC_SEC(pf3);
                    if (!fbe_ok) { pk_fbe_u = NMPC_FBE(uv, uw); fbe_ok = true; }
                    rhs_ls = pk_fbe_u - pk_sigma * nr2;
                    tau = 1.0; ls_n = 0;
                    xv = h ? fma(-1.0, dv, fma(-0.0, rv, uv)) : hv;      // X: u_bar | u+(tau = 1) = u - 0 r - d
                    xw = h ? fma(-1.0, dw, fma(-0.0, rw, uw)) : hw;
                    yqv = fma(-0.5, dv, fma(-0.5, rv, uv));               // Y: u+(tau = 1/2)
                    yqw = fma(-0.5, dw, fma(-0.5, rw, uw));
                    need_grad = true; state = D_ITER;
                    // team: idle waves of this workgroup evaluate the trials tau = 2^-2 .. 2^-10 of this direction meanwhile
                    if (k_help && __builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_HELPERS)) > 0) {
                        if (in && h == 0) { Lreq[t] = dbl2{uv, uw}; Lreq[24 + t] = dbl2{rv, rw}; Lreq[48 + t] = dbl2{dv, dw}; }
                        if (lane == 0) { Lpar[15] = pen_c; Lpar[16] = cbar_inv; Lpar[17] = gamma; }
#ifdef NMPC_TL
                        if (lane == 0) Lpar[22] = (double)tl_it;
                        NMPC_TL_EV(tl_it, 4);
#endif
                        team_seq = team_seq >= 0xffff0u ? 1u : team_seq + 1u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) ctl_store(ctl + CTL_CLAIM + wid, (int)(team_seq << 8));
#ifdef NMPC_TL
                        NMPC_TL_EV(tl_it, 5);
#endif
                        posted = true;
                    }
                }
            }
            // (back-off, return of the inner solver, start of an inner solve: rare, behind one test)
            if (fl & (f_back.bit | f_done.bit | f_start.bit)) {
                if (f_back) {
                    f_back = false;
                    lip_it++;
                    xv = yqv = hv; xw = yqw = hw; need_grad = iteration == 0; state = D_LIP;
                }
                // ---------------------------------------------------------------- the inner solver returned
                if (f_done) {
                    f_done = false;
                    inner_status = timed_out ? NMPC_NOT_CONVERGED_OUT_OF_TIME
                                             : (num_iter < max_inner ? NMPC_CONVERGED : NMPC_NOT_CONVERGED_ITERATIONS);
                    inner_total += num_iter;
                    pk_last_cost = cost;                                     // (pk_last_fpr: the norm of the last step started)
                    uv = hv; uw = hw;                                        // PANOC returns the feasible half step
                    const bool fin = __builtin_isfinite(uv) && __builtin_isfinite(uw) && __builtin_isfinite(cost) && __builtin_isfinite(pk_last_fpr);
                    if (__any(in && !fin)) { final_status = NMPC_NOT_CONVERGED_NOT_FINITE; running = false; }
                    else { xv = yqv = uv; xw = yqw = uw; need_grad = false; state = D_ALM; }
                }
                // ---------------------------------------------------------------- start an inner solve
                if (f_start) {
                    f_start = false;
                    { const dbl2 y_ = *Ly; *Ly = dbl2{clampd(y_.x, -1e12, 1e12), clampd(y_.y, -1e12, 1e12)}; }      // y <- Pi_Y(y)
                    lb_active = 0; lb_first = true; iteration = 0; num_iter = 0; tau = 1.0;
                    NMPC_LB_ZERO();
                    // init evaluates u (points 0, 2) and u + h (point 1), h_i = max(1e-6 u_i, 1e-12)
                    const double h1 = EPSILON_LIPSCHITZ * uv > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uv : DELTA_LIPSCHITZ;
                    const double h2 = EPSILON_LIPSCHITZ * uw > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uw : DELTA_LIPSCHITZ;
                    pk_norm_h = sqrt(group_sum<P>(ina ? fma(h1, h1, h2 * h2) : 0.0, lane));
                    xv = h == 1 ? (in ? uv + h1 : 0.0) : uv;
                    xw = h == 1 ? (in ? uw + h2 : 0.0) : uw;
                    yqv = uv; yqw = uw;
                    need_grad = true; state = D_INIT;
                }
            }
            if (!running) break;

            // ================================================================ one pass: psi at two points
            double psi, pen, egv = 0, egw = 0, eav, eaw;
            n_pass++;
            // Iteration counts are heavy-tailed: an instance that has already run long is likely the
            // one the whole batch will end up waiting for.  Raise its wave's issue priority while it shares its SIMD with another
            // wave: worth 1-2 % of the headline batch (39.2 against 39.9 ms without; levels at 1k / 2k / 3k or 0.5k / 1k / 1.5k passes
            // instead of 2k / 4k / 6k: the same within noise) -- the older wave slot is still served first (top of the file).
            if (k_dbg == 0 && (n_pass & 1023u) == 0u) {
                const unsigned lvl = n_pass >> 11;
                if (lvl == 1u) __builtin_amdgcn_s_setprio(1);
                else if (lvl == 2u) __builtin_amdgcn_s_setprio(2);
                else if (lvl >= 3u) __builtin_amdgcn_s_setprio(3);
            }
            NMPC_SEC(pf4);
#ifdef NMPC_TL
            if (state == D_ITER) NMPC_TL_EV(tl_it, 6);
#endif
#ifdef NMPC_MARKS
            asm volatile("; MARK 10");
#endif
            const dbl2 ycur = *Ly;
            const double yv = ycur.x, yw = ycur.y;
            // query points: state layout -> evaluation layout (transport area; LDS serves the accesses of a wave in order)
            *zwX = dbl2{xv, xw};
            *zwY = dbl2{yqv, yqw};
            const dbl2 zq_ = *zrd, zp_ = *zpv;
            const double zv = zq_.x, zw = zq_.y;
            const EvX evx = {zp_.x, zp_.y, zmine, znext};
#if defined(NMPC_PROF2) && NMPC_PROF2 == 2
            NMPC_SEC_RAW(pe[7]);
            eval_psi<PE, SH, false, CULL, WIN, true>(a, L, f2off, lane, te, zv, zw, pen_c, cbar_inv, yv, yw, *Lvr, dyn, need_grad, psi, pen, egv, egw, eav, eaw, near, &ws, OBSC ? &oc : nullptr, pe, &ek, &evx);
#else
            eval_psi<PE, SH, false, CULL, WIN, true>(a, L, f2off, lane, te, zv, zw, pen_c, cbar_inv, yv, yw, *Lvr, dyn, need_grad, psi, pen, egv, egw, eav, eaw, near, &ws, OBSC ? &oc : nullptr, nullptr, &ek, &evx);
#endif
            *zmine = dbl2{egv, egw};                 // gradients: evaluation layout -> state layout (NMPC_FETCH_GRAD)
#ifdef NMPC_PROF2
            { double keep = psi + egv; asm volatile("" : "+v"(keep)); }
#endif
            NMPC_SEC(pf5);
#ifdef NMPC_TL
            if (state == D_ITER) { NMPC_TL_KEEP(psi + egv); NMPC_TL_EV(tl_it, 7); }
#endif
#ifdef NMPC_MARKS
            asm volatile("; MARK 11");
#endif
            const double psiA = point_scalar(psi, 0), psiB = point_scalar(psi, 1), psiC = point_scalar(psi, 2);
            // one trial of the current direction: psi, grad psi were evaluated by query point K at step tau
            // (Measured and not taken: the envelopes of all three points formed in the evaluation layout right after the evaluation -- as the
            // helpers form theirs -- so that the search is three comparisons and one gradient fetch.  It saves 64 vector instructions per
            // trial after the first but costs 70 on EVERY pass; the bulk of a batch accepts its first trial: headline 40.2 -> 43 ms.)
#define NMPC_TAKE_TRIAL(PSI, SRC) NMPC_TAKE_TRIAL_(PSI, NMPC_FETCH_GRAD((SRC), gv, gw))
#define NMPC_TAKE_TRIAL_(PSI, FETCH)                                                   \
            do {                                                                       \
                n_grad++;                                                              \
                *Lq = dbl2{gv, gw};                  /* cache_previous_gradient */     \
                cost = (PSI);                                                          \
                FETCH;                                                                 \
                const double omt_ = 1.0 - tau;                                         \
                pv = fma(-tau, dv, fma(-omt_, rv, uv));                                \
                pw = fma(-tau, dw, fma(-omt_, rw, uw));                                \
                NMPC_HALF_STEP(pv, pw);                                                \
                lhs = NMPC_FBE(pv, pw);                                                \
                const bool bad_ = __any(lhs > rhs_ls);                                 \
                rejected = bad_ && ls_n < MAX_LINESEARCH_ITERATIONS;                   \
                exhausted = bad_ && !rejected && k_lsf == 1;                 \
                if (rejected) { tau /= 2.0; ls_n++; }                                  \
            } while (0)
            double lhs = 0.0;
            bool rejected = false, exhausted = false;

            if (state == D_INIT) {
                n_grad += 2;
                cost = psiA;
                NMPC_FETCH_GRAD(src0, gv, gw);
                double g1v_, g1w_;
                NMPC_FETCH_GRAD(src1, g1v_, g1w_);
                const double d1 = g1v_ - gv, d2 = g1w_ - gw;
                pk_Lc = sqrt(hdot<P>(d1, d2, d1, d2, lane)) / pk_norm_h;
                gamma = GAMMA_L_COEFF / fmax(pk_Lc, MIN_LIPSCHITZ_CONSTANT);
                pk_sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * gamma);
                pk_c_lip = GAMMA_L_COEFF / (2.0 * gamma);
                pk_hig = 0.5 / gamma;
                NMPC_HALF_STEP(uv, uw);
                fbe_ok = false;
                f_begin = true;
            } else if (state == D_LIP || state == D_ITER) {
                // Lipschitz test on psi(u_bar) (half 0).  D_LIP: sequential (iteration 0, or after a
                // failed speculative pass; the L-BFGS buffer is empty there).  D_ITER: half 1 holds the
                // speculative trial u+(tau = 1) on the tentative direction.
                n_cost++;
                const double rhs = cost + LIPSCHITZ_UPDATE_EPSILON * fabs(cost) - pk_gr + pk_c_lip * nr2;
                if (lip_it < MAX_LIPSCHITZ_UPDATE_ITERATIONS && __any(pk_Lc < MAX_LIPSCHITZ_CONSTANT && psiA > rhs)) {
                    f_back = true;                                       // (speculation discarded)
                } else {
                    if (state == D_LIP) {
                        lb_first = false; *Los = dbl2{uv, uw}; *Log = dbl2{rv, rw};      // first pair after a reset: only remembered
                        if (iteration == 0) {
                            // first iteration: plain forward-backward step; psi, grad psi at u_bar are at hand
                            n_grad++;
                            uv = hv; uw = hw;
                            cost = psiA;
                            NMPC_FETCH_GRAD(src0, gv, gw);
                            NMPC_HALF_STEP(uv, uw);
                            fbe_ok = false;
                            NMPC_END_ITERATION();
                        } else {
                            dv = rv; dw = rw;                            // empty buffer: d = r
                            rhs_ls = NMPC_FBE(uv, uw) - pk_sigma * nr2;
                            tau = 1.0; ls_n = 0;
                            if (k_lsf == 1) *Lgk = dbl2{gv, gw};
                            f_trials = true;
                        }
                    } else {
                        lb_first = n_first; lb_head = n_head; lb_active = n_active; pk_H0 = n_H0;      // commit
                        if (n_take_old) { *Los = dbl2{uv, uw}; *Log = dbl2{rv, rw}; }
                        if (k_lsf == 1) *Lgk = dbl2{gv, gw};
                        NMPC_TAKE_TRIAL(psiB, src1);                     // tau = 1
                        if (rejected) NMPC_TAKE_TRIAL(psiC, src2);       // tau = 1/2
#ifdef NMPC_TL
                        NMPC_TL_KEEP(lhs + hv); NMPC_TL_EV(tl_it, 8);
#endif
                        if (posted) {
                            // tasks 0..2 of the request hold trials ls_n = 2 + 3k .. 4 + 3k.  A task a helper has claimed is
                            // waited for and consumed in order; the first one nobody has claimed is closed (with everything
                            // after it) and this wave goes on by itself, as it would without a team.
                            bool have_prev = false;                      // prev_g: gradient of the last trial a helper's area rejected
                            dbl2 prev_g = dbl2{0.0, 0.0};
                            for (int k = 0; k < 3 && rejected; ++k) {
                                int hid = -1;
                                lds_int *dn = ctl + CTL_DONE + (wid * 3 + k) * TEAM_WAVES;
                                {   // first look, ONE LDS read for the task's four done flags: by now the helper has usually finished
                                    // (its flag is up 600 cycles before this wave is through with its own two trials)
                                    const int fl_ = lane < TEAM_WAVES ? ctl_load(dn + lane) : 0;
                                    const unsigned long long dm_ = __ballot(lane < TEAM_WAVES && fl_ == (int)team_seq);
                                    if (dm_ != 0ull) hid = __builtin_ctzll(dm_);
                                }
                                if (hid < 0) {
                                    if (lane == 0) {
                                        lds_int *cl = ctl + CTL_CLAIM + wid;
                                        bool served = false;
                                        for (;;) {
                                            int v = ctl_load(cl);
                                            if ((v & 0xff) > k) { served = true; break; }
                                            if (__hip_atomic_compare_exchange_strong(cl, &v, (int)(team_seq << 8) | 3, __ATOMIC_RELAXED,
                                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                                        }
                                        if (served) {
                                            for (;;) {
#pragma unroll
                                                for (int w2 = 0; w2 < TEAM_WAVES; ++w2) if (ctl_load(dn + w2) == (int)team_seq) hid = w2;
                                                if (hid >= 0) break;
                                                __builtin_amdgcn_s_sleep(1);
                                            }
                                        }
                                    }
                                    hid = __builtin_amdgcn_readfirstlane(hid);
                                }
#ifdef NMPC_TL
                                if (k == 0) NMPC_TL_EV(tl_it, 9);
#endif
                                if (hid < 0) break;
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                n_pass++;                                // the pass these three trials would have cost this wave
                                // The helper has formed each trial's envelope value too (same canonical sums), so the sequential
                                // decision over its three trials is three comparisons.  Values and gradients of all three come in with
                                // the envelopes -- one LDS round trip instead of two on the critical path of a helped iteration.
                                const lds_double *ar = (const lds_double *)lds + hid * slice + (wid * 3 + k) * TEAM_AREA_DOUBLES;
                                const lds_double2 *ag = (const lds_double2 *)ar + (in ? t : 0);
                                const dbl2 g0_ = ag[0], g1_ = ag[24], g2_ = ag[48];
                                const double c0_ = ar[2 * 72], c1_ = ar[2 * 72 + 1], c2_ = ar[2 * 72 + 2];
                                const double l0_ = ar[2 * 72 + 4], l1_ = ar[2 * 72 + 5], l2_ = ar[2 * 72 + 6];
                                int jstop = 3;
                                if (!(__any(l2_ > rhs_ls) && ls_n + 2 < MAX_LINESEARCH_ITERATIONS)) jstop = 2;
                                if (!(__any(l1_ > rhs_ls) && ls_n + 1 < MAX_LINESEARCH_ITERATIONS)) jstop = 1;
                                if (!(__any(l0_ > rhs_ls) && ls_n < MAX_LINESEARCH_ITERATIONS)) jstop = 0;      // ends as the FIRST trial that is not rejected
                                if (jstop == 3) {                        // all three rejected: on to the next task
                                    n_grad += 3; ls_n += 3; tau *= 0.125;
                                    have_prev = true; prev_g = g2_;
                                } else {
                                    n_grad += (unsigned)jstop + 1u; ls_n += jstop;
                                    tau *= jstop == 0 ? 1.0 : (jstop == 1 ? 0.5 : 0.25);
                                    // cache_previous_gradient: the gradient of the trial before the one that stops the search
                                    if (jstop > 0) { have_prev = true; prev_g = jstop == 1 ? g0_ : g1_; }
                                    if (have_prev) *Lq = dbl2{in ? prev_g.x : 0.0, in ? prev_g.y : 0.0};
                                    else *Lq = dbl2{gv, gw};
                                    cost = jstop == 0 ? c0_ : (jstop == 1 ? c1_ : c2_);
                                    { const dbl2 g_ = jstop == 0 ? g0_ : (jstop == 1 ? g1_ : g2_); gv = in ? g_.x : 0.0; gw = in ? g_.y : 0.0; }
                                    const double omt_ = 1.0 - tau;
                                    pv = fma(-tau, dv, fma(-omt_, rv, uv));
                                    pw = fma(-tau, dw, fma(-omt_, rw, uw));
                                    NMPC_HALF_STEP(pv, pw);
                                    lhs = jstop == 0 ? l0_ : (jstop == 1 ? l1_ : l2_);
                                    exhausted = __any(lhs > rhs_ls) && k_lsf == 1;      // (only the eleventh trial can stop the search while bad)
                                    rejected = false;
                                }
                            }
                            if (rejected && have_prev) {                 // going on alone: the state holds the last rejected trial's gradient
                                gv = in ? prev_g.x : 0.0; gw = in ? prev_g.y : 0.0;
                            }
#ifdef NMPC_TL
                            NMPC_TL_KEEP(uv + pv + gv + hv); NMPC_TL_EV(tl_it, 10);
#endif
                            posted = false;
                            if (lane == 0) ctl_store(ctl + CTL_CLAIM + wid, 0);      // the request is over
                        }
                        if (rejected) f_trials = true;
                        else if (exhausted) f_fb = true;
                        else { uv = pv; uw = pw; pk_fbe_u = lhs; fbe_ok = true; NMPC_END_ITERATION(); }
                    }
                }
            } else if (state == D_LS) {
                // points 0..2 evaluated trials (tau, ls_n), (tau/2, ls_n + 1), (tau/4, ls_n + 2)
                NMPC_TAKE_TRIAL(psiA, src0);
                if (rejected) NMPC_TAKE_TRIAL(psiB, src1);
                if (rejected) NMPC_TAKE_TRIAL(psiC, src2);
                if (rejected) f_trials = true;
                else if (exhausted) f_fb = true;
                else { uv = pv; uw = pw; pk_fbe_u = lhs; fbe_ok = true; NMPC_END_ITERATION(); }
            } else if (state == D_FB) {
                // psi, grad psi at u_bar: the plain forward-backward step (as in iteration 0)
                n_grad++;
                uv = hv; uw = hw;
                cost = psiA;
                NMPC_FETCH_GRAD(src0, gv, gw);
                NMPC_HALF_STEP(uv, uw);
                fbe_ok = false;
                NMPC_END_ITERATION();
            } else {    // D_ALM: F1, F2 at the inner solution
                n_cost++;
                const double tv = fma(yv, cbar_inv, eav), tw = fma(yw, cbar_inv, eaw);
                const double ypv = inea ? fma(pen_c, eav - clampd(tv, ek.amin, ek.amax), yv) : 0.0;
                const double ypw = inea ? fma(pen_c, eaw - clampd(tw, -ek.awmax, ek.awmax), yw) : 0.0;
                *LypE = dbl2{ypv, ypw};
                const double d1 = ypv - yv, d2 = ypw - yw;
                pk_dy_norm_plus = sqrt(group_sum<PE>(inea ? fma(d1, d1, d2 * d2) : 0.0, lane));
                pk_f2_norm_plus = sqrt(pen);
                const double SMALL = DBL_EPSILON;
                const bool crit1 = nu > 0 && __any(pk_dy_norm_plus <= pen_c * a.op.delta_tolerance + SMALL);
                const bool crit2 = a.n2 == 0 || __any(pk_f2_norm_plus <= a.op.delta_tolerance + SMALL);
                const bool crit3 = __any(pk_eps_nu <= a.op.tolerance + SMALL);
                if (crit1 && crit2 && crit3) {
                    final_status = a.op.inner_status == 1 ? NMPC_CONVERGED : inner_status; running = false;
                } else {
                    const bool stall = nu == 0 || __any(pk_dy_norm_plus <= a.op.sufficient_decrease * pk_dy_norm + SMALL &&
                                                        pk_f2_norm_plus <= a.op.sufficient_decrease * pk_f2_norm + SMALL);
                    if (!stall) { pen_c *= a.op.penalty_update; cbar_inv = 1.0 / fmax(pen_c, 1.0); }
                    pk_eps_nu = fmax(a.op.tolerance_update * pk_eps_nu, a.op.tolerance);
                    *Ly = dbl2{ypv, ypw};
                    pk_dy_norm = pk_dy_norm_plus; pk_f2_norm = pk_f2_norm_plus;
                    nu++;
                    if (nu == a.op.max_outer) { final_status = NMPC_NOT_CONVERGED_ITERATIONS; running = false; }
                    else if (timed_out) { final_status = NMPC_NOT_CONVERGED_OUT_OF_TIME; running = false; nu--; }      // (the report adds the one back)
                    else {
                        // ---- an outer-iteration boundary: does this instance keep its wave?
                        // What the iteration just finished revealed splits the batch exactly where it matters: COLD = every outer criterion holds
                        // already, the next outer iteration is the last one, and short (two thirds of the headline batch, median 370 passes);
                        // LONG = a criterion is still open (median 2 800 - 4 600 passes, up to 10 000).  A batch ends when its last instance does and
                        // the waves are busy 70-80 % of that time (scripts/utilisation.py): long instances queue up behind each other on some waves
                        // while others have run out of work.  So (1) a cold instance steps aside while anything else waits for a wave -- fresh
                        // instances show early what they are -- and (2) once more long instances are alive than `sched_long_cap` (a fraction of the
                        // waves), long instances TIME-SHARE: each gives up its wave after every outer iteration while others wait, so that they all
                        // advance together instead of two of them taking turns on one wave (config 4: 111 -> 103 ms; configs 1 and 3: +-1 %).
                        // Only WHERE and WHEN an instance runs changes; its arithmetic, hence every bit of the result, does not (tests: permutation
                        // invariance, repeated solves, parity with the oracle on every path).
                        bool yield_ = false;
                        if (a.sched_mode > 0 || a.park_min > 0) {
                            int dec = 0;
                            if (lane == 0) {
                                const bool fresh_left = (int)__hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.B;
                                int cls = POOL_LONG, y = 0;
                                if (a.sched_mode > 0) {
                                    const bool long_now = !(crit2 && pk_dy_norm_plus <= pen_c * a.op.delta_tolerance + SMALL);
                                    cls = long_now ? POOL_LONG : POOL_COLD;
                                    unsigned int *n_long = a.pool_ctr + POOL_CTRS * NPOOLS;
                                    if (long_now != long_counted) __hip_atomic_fetch_add(n_long, long_now ? 1u : ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    const bool long_wait = pool_depth(a, POOL_LONG) > 0;
                                    const int alive = (int)__hip_atomic_load(n_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    // (a cold instance only steps aside in a batch that HAS long instances to make room for, sched_cold_cap of them; a warm-started
                                    // closed-loop step has next to none, and parking its many tiny instances would cost 5 %)
                                    if (cls == POOL_COLD) y = (fresh_left || long_wait) && alive >= a.sched_cold_cap && n_pass >= 150u;      // (short solves: parking costs more than it gains)
                                    else y = (fresh_left || long_wait) && alive >= a.sched_long_cap && n_pass - q_pass >= 150u;      // (an outer iteration of a few passes is not worth a hand-over)
                                    dec = (long_now ? 2 : 0);
                                }
                                // a long-runner on the unfavoured wave slot while favoured waves will still come back for work: hand it over
                                if (!y && a.park_min > 0 && unfavoured && n_pass >= (unsigned)a.park_min && fresh_left && pool_depth(a, POOL_LONG) < a.park_depth) {
                                    y = 1; cls = POOL_LONG;
                                }
                                dec |= y | (cls << 2);
                            }
                            dec = __builtin_amdgcn_readfirstlane(dec);
                            yield_ = (dec & 1) != 0;
                            if (a.sched_mode > 0) long_counted = (dec & 2) != 0;
                            park_cls = dec >> 2;
                        }
                        q_pass = n_pass;
                        if (yield_) { parked = true; running = false; }
                        else f_start = true;
                    }
                }
            }
#ifdef NMPC_PROF2
            { double keep = uv + gv + hv + cost; asm volatile("" : "+v"(keep)); }
#endif
            NMPC_SEC(pf6);
        }

        // ------------------------------------------------------------------ parked: state out, into the pool
        if (parked) {
            double *po = a.park + (size_t)inst * PS;
            if (in && h == 0) {
                po[2 * t] = uv; po[2 * t + 1] = uw;
                const dbl2 q_ = *Lq;
                po[4 * N + 2 * t] = q_.x; po[4 * N + 2 * t + 1] = q_.y;
            }
            if (ine && q == 0) { const dbl2 y_ = *Ly; po[2 * N + 2 * te] = y_.x; po[2 * N + 2 * te + 1] = y_.y; }
            if (lane == 0) {
                double *ps_ = po + 6 * N;
                ps_[0] = pen_c; ps_[1] = pk_eps_nu; ps_[2] = pk_dy_norm; ps_[3] = pk_f2_norm; ps_[4] = pk_dy_norm_plus;
                ps_[5] = pk_f2_norm_plus; ps_[6] = pk_last_fpr; ps_[7] = pk_last_cost; ps_[8] = (double)nu;
                ps_[9] = (double)inner_total; ps_[10] = (double)n_cost; ps_[11] = (double)n_grad; ps_[12] = (double)n_pass;
                ps_[13] = Lpar[13]; ps_[14] = Lpar[14]; ps_[15] = long_counted ? 1.0 : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) pool_push(a, park_cls, inst);
            if (k_dbg == 0) __builtin_amdgcn_s_setprio(0);
            NMPC_WAVE_SYNC();
            continue;
        }
        // ------------------------------------------------------------------ results
        if (in && h == 0) {
            double *uo = a.u + (size_t)inst * a.n_u;
            uo[2 * t] = uv; uo[2 * t + 1] = uw;
            if (a.y_out) { const dbl2 yp_ = *Lyp; a.y_out[(size_t)inst * a.n1 + t] = yp_.x; a.y_out[(size_t)inst * a.n1 + N + t] = yp_.y; }
        }
        if (lane == 0 && long_counted) __hip_atomic_fetch_add(a.pool_ctr + POOL_CTRS * NPOOLS, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0 && a.st) {
            nmpc_status s;
            s.exit_status = final_status;
            s.num_outer_iterations = (uint32_t)(final_status == NMPC_NOT_CONVERGED_NOT_FINITE ? nu + 1 : (nu < a.op.max_outer ? nu + 1 : nu));
            s.num_inner_iterations = inner_total;
            s.num_cost_evals = n_cost;
            s.num_grad_evals = n_grad;
            s.reserved = n_pass;                 // evaluation passes actually executed (diagnostic)
            s.last_problem_norm_fpr = pk_last_fpr;
            s.delta_y_norm_over_c = pk_dy_norm_plus / pen_c;
            s.f2_norm = pk_f2_norm_plus;
            s.penalty = pen_c;
            s.cost = pk_last_cost;
            // first start -> finish on the constant 100 MHz clock (the parked time of a migrated instance included): what the
            // reference reads per solve (src/mpc/mpc_generator.py:214)
            s.solve_time_ms = (double)((long long)__builtin_amdgcn_s_memrealtime() - (long long)Lpar[13]) * 1e-5;
            if (k_dbg) {       // cycles spent on this instance, wave slot, finish time on the 100 MHz reference clock
                s.last_problem_norm_fpr = (double)(__builtin_amdgcn_s_memtime() - dbg_t0);
                s.f2_norm = (double)hw_slot;
                s.cost = (double)__builtin_amdgcn_s_memrealtime();
                s.delta_y_norm_over_c = Lpar[13]; s.penalty = Lpar[14];
            }
#ifdef NMPC_PROF2
            s.last_problem_norm_fpr = (double)pf0; s.delta_y_norm_over_c = (double)pf1; s.f2_norm = (double)pf2;
            s.penalty = (double)pf3; s.cost = (double)pf4; s.solve_time_ms = (double)pf5;
            s.num_cost_evals = (uint32_t)(pf6 / 64);
#if NMPC_PROF2 == 2
            // rollout | stage cost + cross-track | accelerations, ALM term, cost sum | touched obstacles + adjoint head | adjoint sweep | circle scan | ellipse scan
            s.last_problem_norm_fpr = (double)pe[0]; s.delta_y_norm_over_c = (double)pe[1]; s.f2_norm = (double)pe[2];
            s.penalty = (double)pe[3]; s.cost = (double)pe[4]; s.solve_time_ms = (double)pe[5];
            s.num_cost_evals = (uint32_t)(pe[6] / 64);
#endif
#endif
            a.st[inst] = s;
        }
        if (k_dbg == 0) __builtin_amdgcn_s_setprio(0);
        NMPC_WAVE_SYNC();          // the LDS slice is reused by the next instance
    }

    // ====================================================================== helper: no work of its own (any more)
    // This wave's slice is free now; it holds the result areas, one per (owner, task).
    if (k_dbg == 0) __builtin_amdgcn_s_setprio(0);
    // (a sibling may still be evaluating a cancelled request out of this slice: the instance id goes away before the first result area lands in it)
    if (lane == 0) ctl_store(ctl + CTL_INST + wid, -1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        if (wid < a.team_owners) ctl_add(ctl + CTL_OWNERS, -1);
        ctl_add(ctl + CTL_HELPERS, 1);
    }
    WinState ws_h = {te < N - 1 ? te : N - 2, 0.0, 0.0, 0.0};      // this helper lane's cross-track window, valid for the instance `ws_inst`
    ObsCert oc_h = {0.0, 0.0, 0.0, 0, 0, 0};                      // ... and its obstacle certificate, likewise
    int ws_inst = -1;
    for (;;) {
        if (!k_help || __builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_OWNERS)) <= 0) break;      // (nobody will ask: NMPC_TEAM_HELP=0)
        // claim the next open task of some sibling's request
        int got = -1;
        if (lane == 0) {
            for (int w = 0; w < TEAM_WAVES && got < 0; ++w) {
                if (w == wid) continue;
                int v = ctl_load(ctl + CTL_CLAIM + w);
                if ((v >> 8) != 0 && (v & 0xff) < 3 &&
                    __hip_atomic_compare_exchange_strong(ctl + CTL_CLAIM + w, &v, v + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP))
                    got = (w << 28) | (v & 0x0fffffff);
            }
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (got < 0) { __builtin_amdgcn_s_sleep(2); continue; }
        const int w = got >> 28, k = got & 0xff;
#ifdef NMPC_TL
        const int tl_h = k == 0 ? (int)((lds_double *)lds + w * slice)[mp.par + 22] : -1;
        NMPC_TL_EV(tl_h, 11);
#endif
        const int seq = (got & 0x0fffffff) >> 8;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the owner's slice: tables of its instance, multipliers, reference speeds, the request
        lds_double *Lw = (lds_double *)lds + w * slice;
        const lds_double2 *rq = (const lds_double2 *)(Lw + mp.req) + te;
        const dbl2 u_ = rq[0], r_ = rq[24], d_ = rq[48];
        const double c_w = Lw[mp.par + 15], cbar_w = Lw[mp.par + 16];
        const dbl2 y_w = ((const lds_double2 *)(Lw + mp.vec) + 4 * COLS)[te];
        const double vref_w = Lw[mp.vec + 2 * 5 * COLS + te];
        DynStage dyn_w;
        dyn_w.col = Lw + mp.dyn + te;
        dyn_w.stride = lay_cols<PE>();
        // point q of this pass is trial ls_n = 2 + 3k + q: tau = 2^-ls_n, u+ = u - (1 - tau) r - tau d
        const double tau_w = __hiloint2double((1023 - (2 + 3 * k + q)) << 20, 0), omt_w = 1.0 - tau_w;
        const double zv = fma(-tau_w, d_.x, fma(-omt_w, r_.x, u_.x)), zw = fma(-tau_w, d_.y, fma(-omt_w, r_.y, u_.y));
        double psi, pen, egv = 0, egw = 0, eav, eaw;
        unsigned long long near_w = ~0ull;
        if constexpr (CULL) { const double nb_ = Lw[mp.par + 18]; near_w = ((unsigned long long)(unsigned)__double2hiint(nb_) << 32) | (unsigned)__double2loint(nb_); }
        {                                   // another instance's tables: what this lane knew about its window and its obstacles is void
            const int inst_w = __builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_INST + w));
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (inst_w != ws_inst || inst_w < 0) { ws_inst = inst_w; ws_h.mo2 = 0.0; oc_h.m2 = 0.0; }
        }
#ifdef NMPC_TL
        NMPC_TL_KEEP(zv + zw); NMPC_TL_EV(tl_h, 12);
#endif
        eval_psi<PE, SH, false, CULL, WIN>(a, Lw, f2off, lane, te, zv, zw, c_w, cbar_w, y_w.x, y_w.y, vref_w, dyn_w, true, psi, pen, egv, egw, eav, eaw, near_w, &ws_h, OBSC ? &oc_h : nullptr, nullptr, &ek);
#ifdef NMPC_TL
        NMPC_TL_KEEP(psi + egv); NMPC_TL_EV(tl_h, 13);
#endif
        {                                   // the owner moved on meanwhile: the scans may have seen half-rewritten tables
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (__builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_INST + w)) != ws_inst) { ws_inst = -1; ws_h.mo2 = 0.0; oc_h.m2 = 0.0; }
        }
        // the trial's forward-backward envelope, formed here in the evaluation layout: the tri-layout sums are the same canonical
        // trees as the state layout's (nmpc_device.h), so the value has the bits the owner would compute
        const double gam_w = Lw[mp.par + 17], hig_w = Lw[mp.par + 19];
        const double s1_ = fma(-gam_w, egv, zv), s2_ = fma(-gam_w, egw, zw);
        const double e1_ = s1_ - (inea ? clampd(s1_, vmin, vmax) : s1_), e2_ = s2_ - (inea ? clampd(s2_, -wmax, wmax) : s2_);
        const double dist2_ = group_sum<PE>(inea ? fma(e1_, e1_, e2_ * e2_) : 0.0, lane);
        const double gg_ = group_sum<PE>(inea ? fma(egv, egv, egw * egw) : 0.0, lane);
        const double lhs_ = psi - (0.5 * gam_w) * gg_ + dist2_ * hig_w;
        lds_double *ar = L + (w * 3 + k) * TEAM_AREA_DOUBLES;
        ((lds_double2 *)ar)[24 * q + te] = dbl2{egv, egw};
        if (te == 0) { ar[2 * 72 + q] = psi; ar[2 * 72 + 4 + q] = lhs_; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ctl_store(ctl + CTL_DONE + (w * 3 + k) * TEAM_WAVES + wid, seq);
#ifdef NMPC_TL
        NMPC_TL_EV(tl_h, 14);
#endif
    }
}
#undef pk_eps_nu
#undef pk_dy_norm
#undef pk_f2_norm
#undef pk_dy_norm_plus
#undef pk_f2_norm_plus
#undef pk_last_fpr
#undef pk_last_cost
#undef pk_norm_h
#undef pk_Lc
#undef pk_H0
#undef pk_sigma
#undef pk_c_lip
#undef pk_gr
#undef pk_fbe_u
#undef pk_hig

#undef NMPC_FETCH_GRAD
#undef NMPC_LB_ZERO
#undef NMPC_GRAM_FWD
#undef NMPC_GRAM_LDF
#undef NMPC_GRAM_LDB
#undef NMPC_GRAM_BWD
#undef NMPC_GRAM_BOTH
#undef NMPC_TAKE_TRIAL
#undef NMPC_END_ITERATION
#undef NMPC_HALF_STEP
#undef NMPC_FBE

}  // namespace nmpc
