// nmpc_solve_dual.h -- the 20 < N_hor <= 32 solver (any N_hor <= 32 with NMPC_LAYOUT=dual): ONE problem instance per wavefront, the two 32-lane
// halves of the wave evaluate psi at TWO query points per pass.
//
// Why: at B = 8192 the batch is bounded by the latency of its slowest instance (thousands of
// PANOC iterations), not by chip throughput, so the kernel minimises evaluation passes per
// iteration.  PANOC's iteration k >= 1 needs psi(u_bar) for the Lipschitz test and then
// psi, grad psi at the line-search trial u+ = u - (1-tau) r - tau d.  Both are known before either
// is evaluated (the L-BFGS update is computed tentatively), so half 0 evaluates u_bar while half 1
// evaluates u+(tau = 1); further line-search trials are evaluated two at a time (tau, tau/2), and
// PANOC's initialisation evaluates u and u + h together.  If the Lipschitz test fails the
// speculative half is discarded and the sequential path is followed.  Results, iteration counts
// and evaluation counters are exactly those of the sequential algorithm (oracle/nmpc_oracle.c).
//
// Lane (h, t): h = lane >> 5 selects the query point, t = lane & 31 is the stage.  All solver state
// is replicated in both halves; results cross halves with v_permlane32_swap.
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

// forward-backward envelope at the point whose cost / gradient / gradient step / half step are given
template <int P>
__device__ __forceinline__ double fbe_value(double cost, double gamma, double sv, double sw, double hv, double hw,
                                            double gv, double gw, int lane)
{
    const double e1 = sv - hv, e2 = sw - hw;
    double dist2, gg;
    pair_sum(fma(e1, e1, e2 * e2), fma(gv, gv, gw * gw), lane, dist2, gg);
    return cost - (0.5 * gamma) * gg + (0.5 * dist2) / gamma;
}

// gradient step x - gamma g and its projection on U (lanes beyond the horizon stay zero)
#define NMPC_HALF_STEP(xv, xw)                                                 \
    do {                                                                       \
        const double s1_ = fma(-gamma, gv, (xv)), s2_ = fma(-gamma, gw, (xw)); \
        hv = in ? clampd(s1_, vmin, vmax) : s1_;                               \
        hw = in ? clampd(s2_, -wmax, wmax) : s2_;                              \
    } while (0)
// FBE at the cached point; the gradient step x - gamma g is recomputed (bitwise the same value)
#define NMPC_FBE(xv, xw) fbe_value<P>(cost, gamma, fma(-gamma, gv, (xv)), fma(-gamma, gw, (xw)), hv, hw, gv, gw, lane)

__global__ __launch_bounds__(64, 2) void nmpc_solve_dual_kernel(KArgs a)
{
    constexpr int P = 32;
    extern __shared__ double lds[];
    lds_double *L = (lds_double *)lds;
    const int lane = threadIdx.x, h = lane >> 5, t = lane & 31;
    const int N = a.pb.N, m = a.op.lbfgs_memory;
    const bool in = t < N;
    const int f2off = a.map.f2 + h * (a.n2 + 1);
    lds_double2 *LS = (lds_double2 *)(L + a.map.S);
    lds_double2 *LY = (lds_double2 *)(L + a.map.Y);
    lds_double *Lrho = L + a.map.rho;
    lds_double2 *Los = (lds_double2 *)(L + a.map.vec) + t;      // parked pairs, one column per stage
    lds_double2 *Log = Los + P, *Lq = Los + 2 * P, *Lyp = Los + 3 * P;
    lds_double2 *Lgk = Los + 6 * P;                             // gradient at the current iterate (opts.ls_failure = 1 only)
    const double vmin = a.pb.vmin, vmax = a.pb.vmax, wmax = a.pb.wmax;
    const unsigned max_inner = (unsigned)a.op.max_inner;
    const unsigned budget = (unsigned)a.op.max_total_inner;     // 0 = off

    for (;;) {
        // ------------------------------------------------------------------ next instance from the queue
        unsigned nxt = 0;
        if (lane == 0) nxt = atomicAdd(a.queue, 1u);
        nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)nxt);
        if (nxt >= (unsigned)a.B) break;
        const int inst = a.order ? a.order[nxt] : (int)nxt;
        const long long t_start = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz: per-instance solve_time_ms

        double vref;
        DynStage dyn;
        prepare_instance<P>(a, L, a.p + (size_t)inst * a.n_p, t, vref, dyn);
        WinState ws = {t < N - 1 ? t : N - 2, 0.0, 0.0, 0.0};      // this lane's cross-track window (eval_psi): nothing known yet

        // horizon vectors: lane t holds the (v_t, w_t) pair, identically in both halves unless noted
        const double *u0 = a.u + (size_t)inst * a.n_u;
        double uv = in ? u0[2 * t] : 0.0, uw = in ? u0[2 * t + 1] : 0.0;
        double yv = (a.y0 && in) ? a.y0[(size_t)inst * a.n1 + t] : 0.0;
        double yw = (a.y0 && in) ? a.y0[(size_t)inst * a.n1 + N + t] : 0.0;
        *Lyp = dbl2{yv, yw};
        *Lq = dbl2{0.0, 0.0};                    // gradient_u_previous (AKKT residual) starts at zero
        double gv = 0, gw = 0, hv = 0, hw = 0, rv = 0, rw = 0, dv = 0, dw = 0;
        double pv = 0, pw = 0;                    // line-search trial point of THIS half
        double zv = 0, zw = 0;                    // query point of THIS half
        bool need_grad = true;
        double cost = 0, Lc = 0, gamma = 0, sigma = 0, nr2 = 0, norm_r = 0, tau = 1, rhs_ls = 0, norm_h = 0, H0 = 1;
        double fbe_u = 0;                         // FBE at the current iterate, valid while fbe_ok (an accepted
        bool fbe_ok = false;                      // trial's FBE is the next iteration's: same operands, same bits)
        int iteration = 0, lip_it = 0, ls_n = 0, lb_active = 0, lb_head = 0;
        bool lb_first = true;
        // tentative L-BFGS update of the current iteration (committed when the Lipschitz test passes)
        int n_active = 0, n_head = 0;
        bool n_first = true, n_take_old = false;
        double n_H0 = 1;
        unsigned num_iter = 0;
        const double c0 = a.c0 ? a.c0[inst] : 0.0;
        double pen_c = c0 > 0.0 ? c0 : a.op.initial_penalty;
        double cbar_inv = 1.0 / fmax(pen_c, 1.0);          // 1 / max(c, 1), refreshed when c changes
        double c_lip = 0;                                   // 0.95 / (2 gamma), refreshed when gamma changes
        double eps_nu = a.op.initial_tolerance;
        double dy_norm = 0, f2_norm = 0, dy_norm_plus = DBL_MAX, f2_norm_plus = 0, last_fpr = 0, last_cost = 0;
        int nu = 0, inner_status = 0, state = D_INIT, final_status = 0;
        unsigned inner_total = 0, n_cost = 0, n_grad = 0, n_pass = 0;

        // phase flags (wave-uniform): set by the state handlers, consumed at the top of the loop
        bool f_start = true, f_back = false, f_trials = false, f_end = false, f_begin = false, f_done = false, f_fb = false;
        bool running = true, timed_out = false;
#ifdef NMPC_PROFILE
        { extern __shared__ long long nmpc_prof_lds[]; if (lane < 16) nmpc_prof_lds[4096 + lane] = 0; }
        long long cyc_eval = 0, cyc_top = 0, cyc_post = 0, tk0 = 0, tk1 = 0;
#define NMPC_TICK(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)
        NMPC_TICK(tk0);
#endif

        for (;;) {
            // ---------------------------------------------------------------- backtrack: L <- 2L, gamma <- gamma/2
            if (f_back) {
                f_back = false;
                lb_active = 0; lb_first = true;                         // L-BFGS buffer invalidated
                fbe_ok = false;
                Lc *= 2.0; gamma /= 2.0;
                sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * gamma);
                c_lip = GAMMA_L_COEFF / (2.0 * gamma);
                NMPC_HALF_STEP(uv, uw);
                rv = uv - hv; rw = uw - hw;
                nr2 = hdot<P>(rv, rw, rv, rw, lane);
                norm_r = sqrt(nr2);
                lip_it++;
                zv = hv; zw = hw; need_grad = iteration == 0; state = D_LIP;
            }
            // ---------------------------------------------------------------- line-search trials (tau, ls_n) | (tau/2, ls_n+1)
            if (f_trials) {
                f_trials = false;
                const double th_ = h ? tau / 2.0 : tau;
                const double omt = 1.0 - th_;
                pv = fma(-th_, dv, fma(-omt, rv, uv));
                pw = fma(-th_, dw, fma(-omt, rw, uw));
                zv = pv; zw = pw; need_grad = true; state = D_LS;
            }
            // ---------------------------------------------------------------- every trial failed (opts.ls_failure = 1):
            // tau = 0, the forward-backward step from the current iterate
            if (f_fb) {
                f_fb = false;
                tau = 0.0;
                { const dbl2 gk_ = *Lgk; gv = gk_.x; gw = gk_.y; }
                NMPC_HALF_STEP(uv, uw);
                zv = hv; zw = hw; need_grad = true; state = D_FB;
            }
            // ---------------------------------------------------------------- an iteration finished
            if (f_end) {
                f_end = false;
                iteration++;
                // OpEn: while step() && num_iter < max_iter { num_iter++ }
                if (!(num_iter < max_inner)) f_done = true;
                else {
                    num_iter++;
                    // opts.max_total_inner: the deterministic max_duration
                    if (budget > 0u && inner_total + num_iter >= budget) { timed_out = true; f_done = true; }
                    else f_begin = true;
                }
            }
            // ---------------------------------------------------------------- start of a PANOC step
            if (f_begin) {
                f_begin = false;
                rv = uv - hv; rw = uw - hw;
                nr2 = hdot<P>(rv, rw, rv, rw, lane);
                norm_r = sqrt(nr2);
                bool exit_now = false;
                if (__any(norm_r < a.op.tolerance)) {                    // fpr test, then the AKKT test (opts.akkt_gradient)
                    if (a.op.akkt_gradient == 2) exit_now = true;
                    else {
                        const dbl2 q_ = *Lq;
                        const bool top = a.op.akkt_gradient == 1;       // grad_prev = grad (iteration >= 1) or 0 (iteration 0)
                        const double b1 = top ? (iteration >= 1 ? 0.0 : gv) : gv - q_.x;
                        const double b2 = top ? (iteration >= 1 ? 0.0 : gw) : gw - q_.y;
                        const double a1 = rv / gamma + b1, a2 = rw / gamma + b2;
                        exit_now = __any(sqrt(group_sum<P>(fma(a1, a1, a2 * a2), lane)) < eps_nu);
                    }
                }
                if (exit_now) {
                    f_done = true;
                } else if (iteration == 0) {
                    // psi and grad psi at u_bar serve both the Lipschitz test and the first FB step
                    lip_it = 0;
                    zv = hv; zw = hw; need_grad = true; state = D_LIP;
                } else {
                    lip_it = 0;
                    // ---- tentative L-BFGS update with (s, y) = (u - u_old, r - r_old) ----
                    n_first = lb_first; n_head = lb_head; n_active = lb_active; n_H0 = H0; n_take_old = false;
                    if (lb_first) {
                        n_first = false; n_take_old = true;
                    } else {
                        const dbl2 os_ = *Los, og_ = *Log;
                        const double s1 = uv - os_.x, s2 = uw - os_.y, y1 = rv - og_.x, y2 = rw - og_.y;
                        double ys, ss;
                        pair_sum(fma(s1, y1, s2 * y2), fma(s1, s1, s2 * s2), lane, ys, ss);
                        bool ok = !(ss <= DBL_MIN || ys <= LBFGS_SY_EPSILON);
                        if (ok) ok = ys / ss > LBFGS_CBFGS_EPSILON * norm_r;
                        if (__any(ok)) {
                            n_take_old = true;
                            n_head = lb_head == 0 ? m - 1 : lb_head - 1;
                            if (in && h == 0) { LS[n_head * N + t] = dbl2{s1, s2}; LY[n_head * N + t] = dbl2{y1, y2}; }
                            if (lane == 0) Lrho[n_head] = 1.0 / ys;
                            n_H0 = ys / hdot<P>(y1, y2, y1, y2, lane);
                            if (n_active < m) n_active++;
                            NMPC_WAVE_SYNC();
                        }
                    }
                    // ---- d = H r, two-loop recursion over the tentative buffer ----
                    dv = rv; dw = rw;
                    if (n_active == MAXMEM && m == MAXMEM) {
                        // full buffer (the steady state): branch-free, all pairs addressed statically from the head
                        double alpha[MAXMEM];
                        const int tt = in ? t : 0;
#pragma unroll
                        for (int k = 0; k < MAXMEM; ++k) {
                            int slot = n_head + k; if (slot >= MAXMEM) slot -= MAXMEM;
                            const dbl2 s_ = ld_pair(LS, slot * N + tt, in), y_ = ld_pair(LY, slot * N + tt, in);
                            const double al = Lrho[slot] * hdot<P>(s_.x, s_.y, dv, dw, lane);
                            alpha[k] = al;
                            dv = fma(-al, y_.x, dv); dw = fma(-al, y_.y, dw);
                        }
                        dv = n_H0 * dv; dw = n_H0 * dw;
#pragma unroll
                        for (int k = MAXMEM - 1; k >= 0; --k) {
                            int slot = n_head + k; if (slot >= MAXMEM) slot -= MAXMEM;
                            const dbl2 s_ = ld_pair(LS, slot * N + tt, in), y_ = ld_pair(LY, slot * N + tt, in);
                            const double be = Lrho[slot] * hdot<P>(y_.x, y_.y, dv, dw, lane);
                            const double ab = alpha[k] - be;
                            dv = fma(ab, s_.x, dv); dw = fma(ab, s_.y, dw);
                        }
                    } else if (n_active > 0) {
                        // pair k lives in ring slot (n_head + k) mod m; each trip fetches the NEXT pair from
                        // LDS before it reduces the current one, so the LDS latency hides under the reduction
                        double alpha[MAXMEM];
                                                const int tt = in ? t : 0;               // lanes beyond the horizon read lane 0's pair and drop it
                        int slot = n_head;
                        dbl2 sc_ = ld_pair(LS, slot * N + tt, in), yc_ = ld_pair(LY, slot * N + tt, in);
                        double rc_ = Lrho[slot];
#pragma unroll
                        for (int k = 0; k < MAXMEM; ++k) {
                            alpha[k] = 0.0;
                            if (k < n_active) {
                                dbl2 sn_ = {0.0, 0.0}, yn_ = {0.0, 0.0};
                                double rn_ = 0.0;
                                if (k + 1 < n_active) {
                                    slot = slot + 1 == m ? 0 : slot + 1;
                                    sn_ = ld_pair(LS, slot * N + tt, in); yn_ = ld_pair(LY, slot * N + tt, in); rn_ = Lrho[slot];
                                }
                                const double al = rc_ * hdot<P>(sc_.x, sc_.y, dv, dw, lane);
                                alpha[k] = al;
                                dv = fma(-al, yc_.x, dv); dw = fma(-al, yc_.y, dw);
                                if (k + 1 < n_active) { sc_ = sn_; yc_ = yn_; rc_ = rn_; }
                            }
                        }
                        dv = n_H0 * dv; dw = n_H0 * dw;
                        // (sc_, yc_, rc_) now hold the oldest pair, k = n_active - 1; walk back to the newest
#pragma unroll
                        for (int k = MAXMEM - 1; k >= 0; --k) {
                            if (k < n_active) {
                                dbl2 sn_ = {0.0, 0.0}, yn_ = {0.0, 0.0};
                                double rn_ = 0.0;
                                if (k > 0) {
                                    slot = slot == 0 ? m - 1 : slot - 1;
                                    sn_ = ld_pair(LS, slot * N + tt, in); yn_ = ld_pair(LY, slot * N + tt, in); rn_ = Lrho[slot];
                                }
                                const double be = rc_ * hdot<P>(yc_.x, yc_.y, dv, dw, lane);
                                const double ab = alpha[k] - be;
                                dv = fma(ab, sc_.x, dv); dw = fma(ab, sc_.y, dw);
                                if (k > 0) { sc_ = sn_; yc_ = yn_; rc_ = rn_; }
                            }
                        }
                    }
                    if (!fbe_ok) { fbe_u = NMPC_FBE(uv, uw); fbe_ok = true; }
                    rhs_ls = fbe_u - sigma * nr2;
                    tau = 1.0; ls_n = 0;
                    const double omt = 1.0 - tau;
                    pv = fma(-tau, dv, fma(-omt, rv, uv));
                    pw = fma(-tau, dw, fma(-omt, rw, uw));
                    zv = h ? pv : hv; zw = h ? pw : hw;                  // half 0: u_bar, half 1: u+(tau = 1)
                    need_grad = true; state = D_ITER;
                }
            }
            // ---------------------------------------------------------------- the inner solver returned
            if (f_done) {
                f_done = false;
                inner_status = timed_out ? NMPC_NOT_CONVERGED_OUT_OF_TIME
                                         : (num_iter < max_inner ? NMPC_CONVERGED : NMPC_NOT_CONVERGED_ITERATIONS);
                inner_total += num_iter;
                last_fpr = norm_r; last_cost = cost;
                uv = hv; uw = hw;                                        // PANOC returns the feasible half step
                const bool fin = __builtin_isfinite(uv) && __builtin_isfinite(uw) && __builtin_isfinite(cost) && __builtin_isfinite(norm_r);
                if (__any(in && !fin)) { final_status = NMPC_NOT_CONVERGED_NOT_FINITE; running = false; }
                else { zv = uv; zw = uw; need_grad = false; state = D_ALM; }
            }
            // ---------------------------------------------------------------- start an inner solve
            if (f_start) {
                f_start = false;
                yv = clampd(yv, -1e12, 1e12); yw = clampd(yw, -1e12, 1e12);      // y <- Pi_Y(y)
                lb_active = 0; lb_first = true; iteration = 0; num_iter = 0; tau = 1.0;
                // init evaluates u (half 0) and u + h (half 1), h_i = max(1e-6 u_i, 1e-12)
                const double h1 = EPSILON_LIPSCHITZ * uv > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uv : DELTA_LIPSCHITZ;
                const double h2 = EPSILON_LIPSCHITZ * uw > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uw : DELTA_LIPSCHITZ;
                norm_h = sqrt(group_sum<P>(in ? fma(h1, h1, h2 * h2) : 0.0, lane));
                zv = h ? (in ? uv + h1 : 0.0) : uv;
                zw = h ? (in ? uw + h2 : 0.0) : uw;
                need_grad = true; state = D_INIT;
            }
            if (!running) break;

            // ================================================================ one pass: psi at two points
            double psi, pen, egv = 0, egw = 0, eav, eaw;
            n_pass++;
            // Iteration counts are heavy-tailed: an instance that has already run long is likely the
            // one the whole batch will end up waiting for.  Raise its wave's issue priority so that
            // it runs at (nearly) single-wave speed while it still shares its SIMD with another wave.
            if ((n_pass & 1023u) == 0u) {
                const unsigned lvl = n_pass >> 11;
                if (lvl == 1u) __builtin_amdgcn_s_setprio(1);
                else if (lvl == 2u) __builtin_amdgcn_s_setprio(2);
                else if (lvl >= 3u) __builtin_amdgcn_s_setprio(3);
            }
#ifdef NMPC_PROFILE
            NMPC_TICK(tk1); cyc_top += tk1 - tk0; tk0 = tk1;
#endif
            eval_psi<P, ShapeAny, false, false, NMPC_WIN>(a, L, f2off, lane, t, zv, zw, pen_c, cbar_inv, yv, yw, vref, dyn, need_grad, psi, pen, egv, egw, eav, eaw, ~0ull, &ws);
#ifdef NMPC_PROFILE
            { double keep = psi + egv; asm volatile("" : "+v"(keep)); }
            NMPC_TICK(tk1); cyc_eval += tk1 - tk0; tk0 = tk1;
#endif
            double psiA, psiB, gAv, gBv, gAw, gBw;
            both_halves(psi, psiA, psiB);
            both_halves(egv, gAv, gBv);
            both_halves(egw, gAw, gBw);

            if (state == D_INIT) {
                n_grad += 2;
                cost = psiA; gv = gAv; gw = gAw;
                const double d1 = gBv - gAv, d2 = gBw - gAw;
                Lc = sqrt(hdot<P>(d1, d2, d1, d2, lane)) / norm_h;
                gamma = GAMMA_L_COEFF / fmax(Lc, MIN_LIPSCHITZ_CONSTANT);
                sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * gamma);
                c_lip = GAMMA_L_COEFF / (2.0 * gamma);
                NMPC_HALF_STEP(uv, uw);
                fbe_ok = false;
                f_begin = true;
            } else if (state == D_LIP || state == D_ITER) {
                // Lipschitz test on psi(u_bar) (half 0).  D_LIP: sequential (iteration 0, or after a
                // failed speculative pass; the L-BFGS buffer is empty there).  D_ITER: half 1 holds the
                // speculative trial u+(tau = 1) on the tentative direction.
                n_cost++;
                const double rhs = cost + LIPSCHITZ_UPDATE_EPSILON * fabs(cost) - hdot<P>(gv, gw, rv, rw, lane)
                                 + c_lip * nr2;
                if (lip_it < MAX_LIPSCHITZ_UPDATE_ITERATIONS && __any(Lc < MAX_LIPSCHITZ_CONSTANT && psiA > rhs)) {
                    f_back = true;                                       // (speculation discarded)
                } else {
                    if (state == D_LIP) {
                        lb_first = false; *Los = dbl2{uv, uw}; *Log = dbl2{rv, rw};      // first pair after a reset: only remembered
                        if (iteration == 0) {
                            // first iteration: plain forward-backward step; psi, grad psi at u_bar are at hand
                            n_grad++;
                            uv = hv; uw = hw;
                            cost = psiA; gv = gAv; gw = gAw;
                            NMPC_HALF_STEP(uv, uw);
                            fbe_ok = false;
                            f_end = true;
                        } else {
                            dv = rv; dw = rw;                            // empty buffer: d = r
                            rhs_ls = NMPC_FBE(uv, uw) - sigma * nr2;
                            tau = 1.0; ls_n = 0;
                            if (a.op.ls_failure == 1) *Lgk = dbl2{gv, gw};
                            f_trials = true;
                        }
                    } else {
                        lb_first = n_first; lb_head = n_head; lb_active = n_active; H0 = n_H0;      // commit
                        if (n_take_old) { *Los = dbl2{uv, uw}; *Log = dbl2{rv, rw}; }
                        n_grad++;
                        if (a.op.ls_failure == 1) *Lgk = dbl2{gv, gw};
                        *Lq = dbl2{gv, gw};                              // cache_previous_gradient
                        cost = psiB; gv = gBv; gw = gBw;
                        NMPC_HALF_STEP(pv, pw);
                        const double lhs = NMPC_FBE(pv, pw);
                        if (__any(lhs > rhs_ls) && ls_n < MAX_LINESEARCH_ITERATIONS) { tau /= 2.0; ls_n++; f_trials = true; }
                        else { uv = pv; uw = pw; fbe_u = lhs; fbe_ok = true; f_end = true; }      // (tau = 1 is never the last trial)
                    }
                }
            } else if (state == D_LS) {
                // half 0 evaluated trial (tau, ls_n), half 1 trial (tau/2, ls_n + 1)
                double pAv, pBv, pAw, pBw;
                both_halves(pv, pAv, pBv);
                both_halves(pw, pAw, pBw);
                n_grad++;
                *Lq = dbl2{gv, gw};
                cost = psiA; gv = gAv; gw = gAw;
                pv = pAv; pw = pAw;
                NMPC_HALF_STEP(pv, pw);
                double lhs = NMPC_FBE(pv, pw);
                bool accept = true;
                bool bad = __any(lhs > rhs_ls);
                if (bad && ls_n < MAX_LINESEARCH_ITERATIONS) {
                    tau /= 2.0; ls_n++;
                    n_grad++;
                    *Lq = dbl2{gv, gw};
                    cost = psiB; gv = gBv; gw = gBw;
                    pv = pBv; pw = pBw;
                    NMPC_HALF_STEP(pv, pw);
                    lhs = NMPC_FBE(pv, pw);
                    bad = __any(lhs > rhs_ls);
                    if (bad && ls_n < MAX_LINESEARCH_ITERATIONS) { tau /= 2.0; ls_n++; f_trials = true; accept = false; }
                }
                if (accept && bad && a.op.ls_failure == 1) f_fb = true;        // the last trial failed too: tau = 0
                else if (accept) { uv = pv; uw = pw; fbe_u = lhs; fbe_ok = true; f_end = true; }
            } else if (state == D_FB) {
                // psi, grad psi at u_bar: the plain forward-backward step (as in iteration 0)
                n_grad++;
                uv = hv; uw = hw;
                cost = psiA; gv = gAv; gw = gAw;
                NMPC_HALF_STEP(uv, uw);
                fbe_ok = false;
                f_end = true;
            } else {    // D_ALM: F1, F2 at the inner solution
                n_cost++;
                const double tv = fma(yv, cbar_inv, eav), tw = fma(yw, cbar_inv, eaw);
                const double ypv = in ? fma(pen_c, eav - clampd(tv, a.pb.amin, a.pb.amax), yv) : 0.0;
                const double ypw = in ? fma(pen_c, eaw - clampd(tw, -a.pb.awmax, a.pb.awmax), yw) : 0.0;
                *Lyp = dbl2{ypv, ypw};
                const double d1 = ypv - yv, d2 = ypw - yw;
                dy_norm_plus = sqrt(group_sum<P>(in ? fma(d1, d1, d2 * d2) : 0.0, lane));
                f2_norm_plus = sqrt(pen);
                const double SMALL = DBL_EPSILON;
                const bool crit1 = nu > 0 && __any(dy_norm_plus <= pen_c * a.op.delta_tolerance + SMALL);
                const bool crit2 = a.n2 == 0 || __any(f2_norm_plus <= a.op.delta_tolerance + SMALL);
                const bool crit3 = __any(eps_nu <= a.op.tolerance + SMALL);
                if (crit1 && crit2 && crit3) {
                    final_status = a.op.inner_status == 1 ? NMPC_CONVERGED : inner_status; running = false;
                } else {
                    const bool stall = nu == 0 || __any(dy_norm_plus <= a.op.sufficient_decrease * dy_norm + SMALL &&
                                                        f2_norm_plus <= a.op.sufficient_decrease * f2_norm + SMALL);
                    if (!stall) { pen_c *= a.op.penalty_update; cbar_inv = 1.0 / fmax(pen_c, 1.0); }
                    eps_nu = fmax(a.op.tolerance_update * eps_nu, a.op.tolerance);
                    yv = ypv; yw = ypw;
                    dy_norm = dy_norm_plus; f2_norm = f2_norm_plus;
                    nu++;
                    if (nu == a.op.max_outer) { final_status = NMPC_NOT_CONVERGED_ITERATIONS; running = false; }
                    else if (timed_out) { final_status = NMPC_NOT_CONVERGED_OUT_OF_TIME; running = false; nu--; }      // (the report adds the one back)
                    else f_start = true;
                }
            }
#ifdef NMPC_PROFILE
            { double keep = uv + gv + hv + cost; asm volatile("" : "+v"(keep)); }
            NMPC_TICK(tk1); cyc_post += tk1 - tk0; tk0 = tk1;
#endif
        }

        // ------------------------------------------------------------------ results
        if (in && h == 0) {
            double *uo = a.u + (size_t)inst * a.n_u;
            uo[2 * t] = uv; uo[2 * t + 1] = uw;
            if (a.y_out) { const dbl2 yp_ = *Lyp; a.y_out[(size_t)inst * a.n1 + t] = yp_.x; a.y_out[(size_t)inst * a.n1 + N + t] = yp_.y; }
        }
        if (lane == 0 && a.st) {
            nmpc_status s;
            s.exit_status = final_status;
            s.num_outer_iterations = (uint32_t)(final_status == NMPC_NOT_CONVERGED_NOT_FINITE ? nu + 1 : (nu < a.op.max_outer ? nu + 1 : nu));
            s.num_inner_iterations = inner_total;
            s.num_cost_evals = n_cost;
            s.num_grad_evals = n_grad;
            s.reserved = n_pass;                 // evaluation passes actually executed (diagnostic)
            s.last_problem_norm_fpr = last_fpr;
            s.delta_y_norm_over_c = dy_norm_plus / pen_c;
            s.f2_norm = f2_norm_plus;
            s.penalty = pen_c;
            s.cost = last_cost;
            s.solve_time_ms = (double)((long long)__builtin_amdgcn_s_memrealtime() - t_start) * 1e-5;
#ifdef NMPC_PROFILE
            {
                extern __shared__ long long nmpc_prof_lds[];
                const long long *e = nmpc_prof_lds + 4096;
                s.last_problem_norm_fpr = (double)cyc_eval; s.delta_y_norm_over_c = (double)cyc_top; s.f2_norm = (double)cyc_post;
                s.penalty = (double)e[0]; s.cost = (double)e[1]; s.solve_time_ms = (double)e[2];
                s.num_cost_evals = (uint32_t)(e[3] / 100); s.num_grad_evals = (uint32_t)(e[4] / 100);
                s.num_outer_iterations = (uint32_t)(e[5] / 100); s.num_inner_iterations = (uint32_t)(e[6] / 100);
            }
#endif
            a.st[inst] = s;
        }
        __builtin_amdgcn_s_setprio(0);
        NMPC_WAVE_SYNC();          // the LDS slice is reused by the next instance
    }
}
#undef NMPC_HALF_STEP
#undef NMPC_FBE

}  // namespace nmpc
