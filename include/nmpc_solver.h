/*
 * nmpc_solver.h -- C ABI of the MI355X batched NMPC solver (libnmpc_hip.so).
 *
 * This is the drop-in boundary for the one hot path of wljungbergh/mpc-trajectory-generator that
 * this project accelerates: the NMPC solve that the reference performs through OpEn's generated
 * solver.  Reference interface each entry point replaces (paths relative to the reference repo):
 *
 *   nmpc_new / nmpc_free      og.tcp.OptimizerTcpManager(path).start() / .kill()
 *                             src/path_generator.py:218-220,408,417 ; src/mpc/mpc_generator.py:220
 *                             (and the offline code generation of src/mpc/mpc_generator.py:173-193:
 *                             the quantities that build() bakes into the generated crate are the
 *                             fields of nmpc_problem / nmpc_opts)
 *   nmpc_ping                 mng.ping()                       src/path_generator.py:222
 *   nmpc_solve_batch_host     mng.call(parameters)             src/mpc/mpc_generator.py:206
 *   nmpc_solve_batch_device   same, operands already in HBM    (B parameter vectors per call)
 *   nmpc_status fields        solution_data.exit_status / .solve_time_ms / ...
 *                             src/mpc/mpc_generator.py:211-214
 *   nmpc_eval_batch_*         the generated cost / grad / mapping_f1 / mapping_f2 C functions that
 *                             build() emits (declared by src/mpc/mpc_generator.py:66-175)
 *
 * The shape follows OpEn's own generated C bindings (<name>_new / <name>_solve / <name>_free with a
 * status struct), widened from one instance to a batch.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on success
 * or a negative nmpc_error, never throws; buffers are caller-allocated; a handle is not
 * thread-safe, distinct handles are.  A handle owns device scratch (work queue, launch order, the pools
 * instances wait in between outer iterations): its asynchronous calls must be ordered on ONE stream
 * at a time; two batches in flight take two handles (bench.py's `pipelined` leg).
 * Data layout (all IEEE f64, row-major):
 *   p  [B][n_p]   parameter vectors, layout of reference src/path_generator.py:378-379
 *   u  [B][n_u]   decision vectors (v_0, w_0, v_1, w_1, ...)   src/mpc/mpc_generator.py:83,157-158
 *   y  [B][n1]    ALM multipliers of F1 = [acc ; omega_acc]    src/mpc/mpc_generator.py:162
 */
#ifndef NMPC_SOLVER_H
#define NMPC_SOLVER_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMPC_ABI_VERSION 3
#define NMPC_MAX_HORIZON 40     /* longest N_hor served: the reference ships N_hor = 20 and 15 (configs/default.yaml:7,
                                   configs/smooth_velocity.yaml:7); BASELINE's long-horizon case is 40.  nmpc_new returns
                                   NMPC_ERR_BAD_PROBLEM beyond it (the one-point kernel that took 40 < N <= 64 until ABI 3's
                                   first release -- another L-BFGS arithmetic, no obstacle certificate -- is gone) */

typedef enum nmpc_error {
    NMPC_OK = 0,
    NMPC_ERR_BAD_PROBLEM = -1,      /* unsupported dims / non-positive ts                     */
    NMPC_ERR_BAD_OPTS = -2,
    NMPC_ERR_BAD_ARG = -3,          /* NULL pointer, B < 0, B > max_batch                     */
    NMPC_ERR_NO_DEVICE = -4,        /* no HIP device / wrong architecture                     */
    NMPC_ERR_HIP = -5,              /* a HIP runtime call failed; see nmpc_last_error()       */
    NMPC_ERR_DEAD_HANDLE = -6       /* handle was killed                                      */
} nmpc_error;

/* What the reference bakes into the generated solver at build() time
 * (configs/default.yaml:6-13,18,34-40 ; src/mpc/mpc_generator.py:70-71,151-168). */
typedef struct nmpc_problem {
    int32_t N;        /* N_hor, 2..NMPC_MAX_HORIZON                 */
    int32_t nobs;     /* Nobs    : static circle slots, 0..64       */
    int32_t ndyn;     /* Ndynobs : dynamic ellipse slots, 0..3      */
    int32_t reserved;
    double ts;
    double vmin, vmax, wmax;     /* U = ([vmin,vmax] x [-wmax,wmax])^N      (:151-153) */
    double amin, amax, awmax;    /* C = [amin,amax]^N x [-awmax,awmax]^N    (:164-168) */
} nmpc_problem;

/* OpEn solver configuration (src/mpc/mpc_generator.py:184-186; opengen defaults otherwise).
 * Deviation: the reference stops on wall-clock (max_duration 0.5 s, :9,186); a batch must be
 * deterministic, so the iteration caps apply, plus -- when asked for -- max_total_inner, the
 * deterministic stand-in for max_duration. */
typedef struct nmpc_opts {
    double tolerance;            /* 1e-4 */
    double initial_tolerance;    /* 1e-4 */
    double delta_tolerance;      /* 1e-4 */
    double initial_penalty;      /* 1.0  */
    double penalty_update;       /* 5.0  */
    double tolerance_update;     /* 0.1  */
    double sufficient_decrease;  /* 0.1  */
    int32_t lbfgs_memory;        /* 10 (1..10) */
    int32_t max_inner;           /* 500  */
    int32_t max_outer;           /* 10   */
    int32_t max_total_inner;     /* 0 = off.  PANOC iterations one solve may spend in total; beyond it the
                                    solve returns the feasible half step with NMPC_NOT_CONVERGED_OUT_OF_TIME
                                    (what max_duration does in the reference, src/mpc/mpc_generator.py:9,186,
                                    configs/default.yaml:49, counted in iterations instead of microseconds) */
    /* Restatement switches: choices of the PANOC/ALM restatement that cannot be checked against OpEn
     * here.  0 = what every published figure of this project uses unless it
     * says otherwise; oracle and kernels implement all of them, bit for bit. */
    int32_t akkt_gradient;       /* "previous gradient" of the AKKT residual ||r/gamma + grad - grad_prev||:
                                    0 cached before every line-search trial, carried across inner solves
                                    1 cached at the top of step() from iteration 1 on, zero at iteration 0
                                    2 no AKKT test (PANOC stops on ||r|| < tolerance alone)                 */
    int32_t ls_failure;          /* all 11 line-search trials fail: 0 take the last one, 1 tau = 0 (FB step) */
    int32_t inner_status;        /* outer criteria hold: 0 report the last inner status, 1 report Converged */
    int32_t reserved;
} nmpc_opts;

typedef enum nmpc_exit {
    NMPC_CONVERGED = 0,
    NMPC_NOT_CONVERGED_ITERATIONS = 1,
    NMPC_NOT_CONVERGED_OUT_OF_TIME = 2,      /* opts.max_total_inner spent (deterministic max_duration) */
    NMPC_NOT_CONVERGED_COST = 3,
    NMPC_NOT_CONVERGED_NOT_FINITE = 4        /* Deviation from OpEn, which checks the returned u only: raised as well when
                                                the cost or the residual norm ||r|| of an inner solve is not finite
                                                while the projected half step u still is (e.g. an initial penalty that
                                                overflows psi).  Through tcp_shim this is error 2000; the reference
                                                would go on with clamped controls (tests: test_nonfinite_cost_*) */
} nmpc_exit;

/* One per instance: the fields of OpEn's solver status plus evaluation
 * counters.  72 bytes. */
typedef struct nmpc_status {
    int32_t  exit_status;            /* nmpc_exit                                   */
    uint32_t num_outer_iterations;
    uint32_t num_inner_iterations;
    uint32_t num_cost_evals;         /* forward-only evaluations of psi             */
    uint32_t num_grad_evals;         /* forward + adjoint evaluations               */
    uint32_t reserved;               /* diagnostic: evaluation passes this solve needs in the kernel's schedule --
                                        three query points per pass (nmpc_solve_hyb_kernel, N_hor <= 20; a pass whose
                                        trials were evaluated by helper waves of the team counts like one the owner
                                        ran itself, so the figure is deterministic; nmpc_solve_hyb2_kernel likewise for
                                        20 < N_hor <= 40), one (nmpc_solve_kernel<64>: = num_cost_evals + num_grad_evals) */
    double last_problem_norm_fpr;
    double delta_y_norm_over_c;
    double f2_norm;
    double penalty;
    double cost;
    double solve_time_ms;            /* THIS instance: first start -> finish on the device's constant 100 MHz clock
                                        (what the reference reads per solve, src/mpc/mpc_generator.py:214); the
                                        wall time of a whole host-path batch is nmpc_last_batch_ms()          */
} nmpc_status;

typedef struct nmpc_handle nmpc_handle;

void nmpc_default_opts(nmpc_opts *opts);
int nmpc_n_u(const nmpc_problem *pb);
int nmpc_n_p(const nmpc_problem *pb);
int nmpc_n1(const nmpc_problem *pb);
int nmpc_n2(const nmpc_problem *pb);

/* Creates a solver for one problem shape on HIP device `device_id`, with room for `max_batch`
 * instances.  opts == NULL selects nmpc_default_opts. */
int nmpc_new(const nmpc_problem *pb, const nmpc_opts *opts, int device_id, int max_batch,
             nmpc_handle **out);
void nmpc_free(nmpc_handle *h);
int nmpc_ping(const nmpc_handle *h);
const char *nmpc_last_error(const nmpc_handle *h);
int nmpc_abi_version(void);
/* 0 for the shipped library: it reads no environment variable and has no tuning knob beyond nmpc_opts.  1 for the experiments build
 * (-DNMPC_EXPERIMENTS, csrc/variants/libnmpc_experiments.so), which tests and measurement scripts use to force alternative kernels and
 * scheduling policies and check that they give the same bits.  (The interface being replaced has no knobs: src/path_generator.py:218-222.) */
int nmpc_experiments_build(void);
/* Diagnostic: the solve kernel this handle launches (the name a rocprofv3 kernel trace shows). */
const char *nmpc_kernel_name(const nmpc_handle *h);

/* Device path: every pointer is device memory on the handle's device; the launch is enqueued on
 * `stream` (a hipStream_t, NULL = default stream) and the call returns without synchronising.
 *   d_u      [B][n_u]  in: initial guess, out: solution
 *   d_y0     [B][n1]   initial multipliers, or NULL (zeros)
 *   d_c0     [B]       initial penalties, or NULL (opts.initial_penalty)
 *   d_y_out  [B][n1]   final multipliers, or NULL
 *   d_status [B]       or NULL                                                              */
int nmpc_solve_batch_device(nmpc_handle *h, int B, const double *d_p, double *d_u,
                            const double *d_y0, const double *d_c0, double *d_y_out,
                            nmpc_status *d_status, void *stream);

/* Host path: same operands in host memory; copies in, solves, copies out, synchronises. */
int nmpc_solve_batch_host(nmpc_handle *h, int B, const double *p, double *u, const double *y0,
                          const double *c0, double *y_out, nmpc_status *status);
/* Kernel time (HIP events around the launch) of the last nmpc_solve_batch_host call on this handle, in ms. */
double nmpc_last_batch_ms(const nmpc_handle *h);

/* psi(u; c, y, p), grad_u psi, F1, F2 for B instances (c == NULL: zeros -> plain f; y == NULL: zeros).
 * Outputs may be NULL.  psi [B], grad [B][n_u], F1 [B][n1], F2 [B][n2]. */
int nmpc_eval_batch_device(nmpc_handle *h, int B, const double *d_p, const double *d_u,
                           const double *d_c, const double *d_y, double *d_psi, double *d_grad,
                           double *d_F1, double *d_F2, void *stream);
int nmpc_eval_batch_host(nmpc_handle *h, int B, const double *p, const double *u, const double *c,
                         const double *y, double *psi, double *grad, double *F1, double *F2);

/* ---- the receding-horizon loop on device ------------------------------------------------------
 * B robots follow one route in lock step; one step = assemble p (the body of the reference's
 * PathGenerator.run loop, src/path_generator.py:290-382), solve the batch warm-started from the
 * previous controls and multipliers (mng.call, src/mpc/mpc_generator.py:206), advance the states
 * over num_steps_taken controls (src/mpc/mpc_generator.py:223-235) and evaluate the terminal test
 * (src/path_generator.py:397).  Nothing crosses PCIe between steps. */
typedef struct nmpc_route {
    int32_t n_ref;             /* samples of rough_ref                       src/mpc/mpc_generator.py:17-57  */
    int32_t n_vert;            /* vertices the route bends around            src/visibility/visibility.py:126-139 */
    int32_t n_brake;           /* entries of the braking tables (>= 1)       src/path_generator.py:439-477   */
    int32_t num_steps_taken;   /* controls applied per solve                 configs/default.yaml:16         */
    const double *x_ref, *y_ref, *theta_ref;    /* [n_ref]   host memory, copied by nmpc_loop_new */
    const double *vert_xy;                      /* [n_vert][2]                                       */
    const double *brake_vel, *brake_dist;       /* [n_brake]                                         */
    double end[3];             /* goal pose                                  src/path_generator.py:327      */
    double base_speed;         /* lin_vel_max * throttle_ratio               src/path_generator.py:284      */
    double radius;             /* static circle radius                       src/path_generator.py:300-301  */
    double dyn_pad;            /* added to the ellipse radii                 src/visibility/visibility.py:208-209 */
    double weights[10];        /* p[10:20]                                   src/path_generator.py:226-227  */
} nmpc_route;

typedef struct nmpc_loop nmpc_loop;

/* starts [B][3] (x, y, theta); idx0 [B] = reference sample each robot starts at, or NULL (0, as
 * the reference).  K <= Ndynobs moving ellipses per robot, dyn [B][K][10] = (p1x, p1y, p2x, p2y,
 * freq, rx, ry, angle, sinus, direction): the reference's linear law (visibility.py:156-166), or
 * with sinus != 0 its sinusoidal law (:177-196, amplitude 1.5; direction = atan2(p2y - p1y,
 * p2x - p1x), computed by the caller); NULL with K = 0.
 * max_steps > 0 also records the trajectory on device. */
int nmpc_loop_new(nmpc_handle *h, const nmpc_route *route, int B, const double *starts,
                  const int32_t *idx0, int K, const double *dyn, int max_steps, nmpc_loop **out);
void nmpc_loop_free(nmpc_loop *l);
/* Enqueues assemble -> solve -> advance on `stream`; does not synchronise. */
int nmpc_loop_step(nmpc_loop *l, void *stream);
/* Synchronises, then copies what is asked for (NULL = skip) to host memory:
 * state [B][3], last_u [B][2], idx [B], done [B], status [B] of the last solve. */
int nmpc_loop_read(nmpc_loop *l, double *state, double *last_u, int32_t *idx, uint8_t *done,
                   nmpc_status *status);
/* The parameter vectors of the last step p [B][n_p], the controls u [B][n_u] and multipliers y [B][n1]. */
int nmpc_loop_params(nmpc_loop *l, double *p, double *u, double *y);
/* Recorded trajectory: rows [steps * num_steps_taken + 1][B][3]; returns the row count (or < 0). */
int nmpc_loop_trajectory(nmpc_loop *l, double *rows, int max_rows);

/* Arithmetic primitives of the kernels, exported for bit-level checks: out_s/out_c [n]. */
int nmpc_test_sincos_host(nmpc_handle *h, int n, const double *x, double *out_s, double *out_c);
/* a/b and sqrt(a) as the device computes them: out_div/out_sqrt [n]. */
int nmpc_test_divsqrt_host(nmpc_handle *h, int n, const double *a, const double *b,
                           double *out_div, double *out_sqrt);

#ifdef __cplusplus
}
#endif
#endif
