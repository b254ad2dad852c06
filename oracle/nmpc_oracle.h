/*
 * nmpc_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C, f64 restatement of the NMPC solve path of wljungbergh/mpc-trajectory-generator:
 *   - cost / constraint definition: reference src/mpc/mpc_generator.py:66-171 (in tree, pinned by
 *     tests/golden/cost_*.npz which were produced by executing that very file);
 *   - the solver the reference delegates to (OpEn: ALM/penalty outer loop + PANOC inner loop +
 *     L-BFGS), which is NOT in the reference tree (opengen==0.6.4, env/environment.yml:14; Rust
 *     crates optimization_engine / lbfgs, unpinned).  It is restated from its published
 *     algorithm.  SOLVER-LEVEL PARITY WITH OpEn IS UNPINNED: the reference
 *     holds no tests / golden trajectories and OpEn cannot be built here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (mpc_trajectory_generator_amd/) never links, imports or calls it.
 */
#ifndef NMPC_ORACLE_H
#define NMPC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Quantities the reference bakes into the generated solver at build() time
 * (configs/default.yaml:6-13,18,34-40; src/mpc/mpc_generator.py:70-71,151-168). */
typedef struct orc_problem {
    int32_t N;        /* N_hor                                  */
    int32_t nobs;     /* Nobs    : static circle slots          */
    int32_t ndyn;     /* Ndynobs : dynamic ellipse slots        */
    int32_t reserved;
    double ts;
    double vmin, vmax, wmax;     /* U = ([vmin,vmax] x [-wmax,wmax])^N      (:151-153) */
    double amin, amax, awmax;    /* C = [amin,amax]^N x [-awmax,awmax]^N    (:164-168) */
} orc_problem;

/* OpEn solver knobs (mpc_generator.py:184-186 sets tolerance; the rest are opengen defaults,
 * SURVEY.md App. C.1). */
typedef struct orc_opts {
    double tolerance;          /* epsilon        1e-4 */
    double initial_tolerance;  /* epsilon_0      1e-4 */
    double delta_tolerance;    /* delta          1e-4 */
    double initial_penalty;    /* c0             1.0  */
    double penalty_update;     /* rho            5.0  */
    double tolerance_update;   /* beta           0.1  */
    double sufficient_decrease;/* theta          0.1  */
    int32_t lbfgs_memory;      /* m              10   */
    int32_t max_inner;         /*                500  */
    int32_t max_outer;         /*                10   */
    int32_t max_total_inner;   /* 0 = off.  Deterministic stand-in for max_duration (mpc_generator.py:9,186):
                                  the solve stops once this many PANOC iterations have been spent in total and
                                  reports NotConvergedOutOfTime with the feasible half step                    */
    /* restatement switches: choices of the OpEn restatement that cannot be verified here (DESIGN.md
     * section 9); 0 is what round 1 shipped and what every figure is quoted for unless stated        */
    int32_t akkt_gradient;     /* AKKT residual ||r/gamma + grad - grad_prev||: grad_prev is
                                  0 the gradient before the last overwrite (cached before EVERY line-search
                                    trial, carried across the inner solves of one call)
                                  1 cached once at the top of step(), before the exit test, from iteration 1 on
                                    and zero at iteration 0 (residual = ||r||/gamma from iteration 1 on)
                                  2 no AKKT test: PANOC stops on ||r|| < epsilon alone                          */
    int32_t ls_failure;        /* all 11 line-search trials fail: 0 the last one is taken anyway,
                                  1 tau = 0: plain forward-backward step from the current iterate             */
    int32_t inner_status;      /* outer criteria hold: 0 report the last inner solve's status,
                                  1 report Converged                                                          */
    int32_t reserved;
} orc_opts;

typedef struct orc_status {
    int32_t  exit_status;            /* 0 Converged, 1 NotConvergedIterations, 2 NotConvergedOutOfTime,
                                        3 NotConvergedCost, 4 NotConvergedNotFiniteComputation */
    uint32_t num_outer_iterations;
    uint32_t num_inner_iterations;
    uint32_t num_cost_evals;         /* forward-only evaluations of psi            */
    uint32_t num_grad_evals;         /* forward + adjoint evaluations of psi, grad */
    uint32_t reserved;
    double last_problem_norm_fpr;
    double delta_y_norm_over_c;
    double f2_norm;
    double penalty;
    double cost;
    double solve_time_ms;
} orc_status;

int orc_n_u(const orc_problem *pb);
int orc_n_p(const orc_problem *pb);
int orc_n1(const orc_problem *pb);
int orc_n2(const orc_problem *pb);
void orc_default_opts(orc_opts *o);

/* psi(u; c, y, p), grad_u psi, F1, F2.  c = 0, y = NULL gives f and grad f.
 * grad / F1 / F2 may be NULL.  Returns 0, or <0 on a bad descriptor. */
int orc_eval(const orc_problem *pb, const double *p, const double *u, double c, const double *y,
             double *psi, double *grad, double *F1, double *F2);

/* One ALM/PANOC solve.  u: in = initial guess, out = solution.  y: in = initial multipliers
 * (NULL = zeros), y_out (may be NULL) = final multipliers.  c0 <= 0 selects opts->initial_penalty. */
int orc_solve(const orc_problem *pb, const orc_opts *opts, const double *p, double *u,
              const double *y0, double c0, double *y_out, orc_status *st);

/* B independent solves on `threads` host threads (pthreads pulling from a shared work queue). */
int orc_solve_batch(const orc_problem *pb, const orc_opts *opts, int B, const double *p, double *u,
                    const double *y0, const double *c0, double *y_out, orc_status *st, int threads);

/* primitives, exported so the GPU's can be checked bit-for-bit */
void orc_sincos(double x, double *s, double *c);
void orc_sincos_n(int n, const double *x, double *s, double *c);     /* the same, element by element */
double orc_tree_sum(const double *v, int n);
/* the Gram-form L-BFGS alone (tests compare it with a two-loop recursion): npush pairs enter an empty buffer of memory m, d_io: r -> H r */
int orc_test_lbfgs_gram(int N, int m, int npush, const double *s_list, const double *y_list, double *d_io);

#ifdef __cplusplus
}
#endif
#endif
