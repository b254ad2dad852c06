/*
 * nmpc_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See nmpc_oracle.h.
 *
 * PARITY STATUS
 *   cost layer  (f, F1, F2, grad f, psi, grad psi): PINNED by tests/golden/cost_*.npz, produced by
 *               executing the reference's own MpcModule.build() (src/mpc/mpc_generator.py:66-193).
 *   solver layer (PANOC / L-BFGS / ALM):            PARITY UNPINNED.  OpEn is a third-party
 *               dependency absent from the reference tree (opengen==0.6.4 -> Rust crates
 *               optimization_engine + lbfgs, version not pinned by the reference); it is restated
 *               here from its published algorithm (Stella et al., CDC 2017; Sathya et al., ECC 2018;
 *               SURVEY.md App. C).  Anchors: the reference's call sites
 *               src/mpc/mpc_generator.py:173-193,206-221 and src/path_generator.py:218-222.
 *
 * CANONICAL ARITHMETIC
 *   Every floating-point operation below is written out explicitly (fma() where a fused
 *   multiply-add is meant; the file is compiled with -ffp-contract=off) and all reductions /
 *   scans over the horizon use a fixed, hardware-independent shape:
 *     tree_sum : zero-pad to P (32 for N <= 20, 64 for 20 < N <= 40); adjacent-pair binary tree
 *     prefix / suffix sums : Kogge-Stone inside blocks of 16 stages, then block carries; for 20 < N <= 40 the
 *                            PAIR form (two consecutive stages are summed first, the Kogge-Stone scan runs over the
 *                            32 pair sums, the first stage of a pair adds its own value to the exclusive result):
 *                            what a kernel that keeps two stages per lane computes (nmpc_solve_hyb2.h)
 *     quarter dot : the inner products of the Gram-form L-BFGS: qdot() below
 *   Horizons: 2 <= N <= 40, what the kernels behind include/nmpc_solver.h accept (the reference ships N = 20 and 15, configs/default.yaml:7).
 *   ONE L-BFGS arithmetic: the Gram form of nmpc_solve_hyb.h / nmpc_solve_hyb2.h (algebraically the two-loop recursion of OpEn's lbfgs
 *   crate; tests/test_oracle_solver.py holds a numpy two-loop recursion and compares).
 *   Three formulas are written the cheap way round (same algebra, another rounding):
 *     - the AKKT residual under akkt_gradient = 1 from iteration 1 on is ||r|| / gamma: tested as ||r|| < eps * gamma;
 *     - the envelope's last term ||w - u_bar||^2 / (2 gamma) is dist2 * (0.5 / gamma), the factor formed when gamma changes;
 *     - the C-BFGS safeguard <y, s> / ||s||^2 > eps ||r|| is tested as <y, s> > (eps ||r||) ||s||^2.
 *   so that an implementation on any machine with IEEE-754 f64 add/mul/fma/div/sqrt can
 *   reproduce the results bit for bit.  sin/cos are computed by orc_sincos() (Cody-Waite
 *   reduction + fdlibm kernels written with fma), never by libm.
 */
#include "nmpc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXP 64     /* padded horizon (lanes)        */
#define MAXN 40     /* longest horizon served (include/nmpc_solver.h) */
#define MAXOBS 64   /* static circle slots           */
#define MAXDYN 8    /* dynamic ellipse slots         */
#define MAXMEM 10   /* L-BFGS memory (= GRAM_M: the ring the kernels are built for) */
#define NZ 20       /* reference configs/default.yaml:35 ; mpc_generator.py:73-75 unpack z0[0..19] */

/* ------------------------------------------------------------------------------------------ */
/* primitives                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* sin and cos of x.  k = rint(x*2/pi); r = x - k*pi/2 in three fma steps (fdlibm's pio2_1,
 * pio2_2, pio2_2t); fdlibm __kernel_sin/__kernel_cos minimax polynomials in Horner form. */
void orc_sincos(double x, double *s, double *c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_2 = 6.07710050630396597660e-11;
    const double PIO2_2T = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, PIO2_1, x);
    r = fma(-k, PIO2_2, r);
    r = fma(-k, PIO2_2T, r);
    double z = r * r;
    double ps = fma(S6, z, S5);
    ps = fma(ps, z, S4);
    ps = fma(ps, z, S3);
    ps = fma(ps, z, S2);
    ps = fma(ps, z, S1);
    double sr = fma(r * z, ps, r);
    double pc = fma(C6, z, C5);
    pc = fma(pc, z, C4);
    pc = fma(pc, z, C3);
    pc = fma(pc, z, C2);
    pc = fma(pc, z, C1);
    double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    int n = ((int)k) & 3;
    double so = (n & 1) ? cr : sr;
    double co = (n & 1) ? sr : cr;
    if (n == 2 || n == 3) so = -so;
    if (n == 1 || n == 2) co = -co;
    *s = so;
    *c = co;
}

void orc_sincos_n(int n, const double *x, double *s, double *c)
{
    for (int i = 0; i < n; ++i) orc_sincos(x[i], s + i, c + i);
}

/* padded horizon: 32 stages for N <= 32, else 64 */
static int pad_pow2(int n) { return n <= 32 ? 32 : 64; }
static int horizon_pad(int N);

/* canonical horizon reduction: adjacent-pair binary tree over P (power of two) entries,
 * ((v0+v1)+(v2+v3))+...; entries >= N are zero.  Destroys v. */
static double tree_sum_p(double *v, int P)
{
    for (int off = 1; off < P; off <<= 1)
        for (int j = 0; j < P; j += 2 * off) v[j] = v[j] + v[j + off];
    return v[0];
}

double orc_tree_sum(const double *v, int n)
{
    double t[MAXP];
    int P = pad_pow2(n);
    if (P > MAXP) return NAN;
    for (int j = 0; j < P; ++j) t[j] = j < n ? v[j] : 0.0;
    return tree_sum_p(t, P);
}

/* canonical inclusive prefix sum over the horizon (P = 32 or 64 entries, zero padded):
 *   1. Kogge-Stone with offsets 1, 2, 4, 8 inside each block of 16 entries (zero fill at the block start);
 *   2. every odd block adds the last entry of the block before it;
 *   3. (P = 64) entries 32..63 add entry 31. */
static void ks_prefix(double *v, int P)
{
    double t[MAXP];
    for (int off = 1; off < 16 && off < P; off <<= 1) {
        for (int j = 0; j < P; ++j) t[j] = (j & 15) >= off ? v[j] + v[j - off] : v[j];
        memcpy(v, t, sizeof(double) * P);
    }
    for (int b = 1; 16 * b < P; b += 2) {
        const double carry = v[16 * b - 1];
        for (int j = 16 * b; j < 16 * b + 16; ++j) v[j] = v[j] + carry;
    }
    if (P == 64) {
        const double carry = v[31];
        for (int j = 32; j < 64; ++j) v[j] = v[j] + carry;
    }
}

/* canonical inclusive suffix sum: the mirror image of ks_prefix */
static void ks_suffix(double *v, int P)
{
    double t[MAXP];
    for (int off = 1; off < 16 && off < P; off <<= 1) {
        for (int j = 0; j < P; ++j) t[j] = (j & 15) + off <= 15 ? v[j] + v[j + off] : v[j];
        memcpy(v, t, sizeof(double) * P);
    }
    for (int b = 0; 16 * (b + 1) < P; b += 2) {
        const double carry = v[16 * (b + 1)];
        for (int j = 16 * b; j < 16 * b + 16; ++j) v[j] = v[j] + carry;
    }
    if (P == 64) {
        const double carry = v[32];
        for (int j = 0; j < 32; ++j) v[j] = v[j] + carry;
    }
}

/* PAIR form of the scans (horizons 20 < N <= 40, P = 64):  w_i = v[2i] + v[2i+1];  W = ks_prefix(w) over 32 entries;
 *   prefix[2i+1] = W_i,  prefix[2i] = (i ? W_{i-1} : 0.0) + v[2i];
 * suffix, the mirror image:  Z = ks_suffix(w);  suffix[2i] = Z_i,  suffix[2i+1] = (i < 31 ? Z_{i+1} : 0.0) + v[2i+1]. */
static void pair_prefix(double *v)
{
    double w[32], out[64];
    for (int i = 0; i < 32; ++i) w[i] = v[2 * i] + v[2 * i + 1];
    ks_prefix(w, 32);
    for (int i = 0; i < 32; ++i) {
        out[2 * i + 1] = w[i];
        out[2 * i] = (i ? w[i - 1] : 0.0) + v[2 * i];
    }
    memcpy(v, out, sizeof(out));
}

static void pair_suffix(double *v)
{
    double w[32], out[64];
    for (int i = 0; i < 32; ++i) w[i] = v[2 * i] + v[2 * i + 1];
    ks_suffix(w, 32);
    for (int i = 0; i < 32; ++i) {
        out[2 * i] = w[i];
        out[2 * i + 1] = (i < 31 ? w[i + 1] : 0.0) + v[2 * i + 1];
    }
    memcpy(v, out, sizeof(out));
}

/* the scans of a horizon of N stages padded to P entries: 20 < N <= 40 is served by the kernel that keeps two stages per lane */
static void scan_prefix(double *v, int P, int N) { if (N > 20 && N <= 40) pair_prefix(v); else ks_prefix(v, P); }
static void scan_suffix(double *v, int P, int N) { if (N > 20 && N <= 40) pair_suffix(v); else ks_suffix(v, P); }
/* padded horizon of a problem: 32 entries for N <= 20 (one stage per lane, three query points per wave), else 64 */
static int horizon_pad(int N) { return N <= 20 ? 32 : 64; }

/* max/min with the semantics of the IEEE maxNum/minNum the GPU's v_max_f64/v_min_f64 implement,
 * for the non-NaN operands this path produces (second operand is always a finite constant) */
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double clampd(double x, double lo, double hi) { return dmin(dmax(x, lo), hi); }

/* a horizon vector: lane t holds the (v_t, omega_t) pair; lanes >= N are zero */
typedef struct { double v[MAXP], w[MAXP]; } hvec;

static double hdot(const hvec *a, const hvec *b, int P)
{
    double t[MAXP];
    for (int j = 0; j < P; ++j) t[j] = fma(a->v[j], b->v[j], a->w[j] * b->w[j]);
    return tree_sum_p(t, P);
}

/* The inner product of the Gram-form L-BFGS: the nst (20 or 40, zero padded) stages in four quarters, each quarter one
 * sequential fma chain over (v_t, w_t) from +0.0, the quarters combined as (q0 + q1) + (q2 + q3).  What a wavefront computes when
 * lane (vector, quarter) runs its chain and the four wave rows are then added with two permlane swaps (nmpc_solve_hyb.h). */
static double qdot(const hvec *a, const hvec *b, int nst)
{
    double q[4];
    const int n4 = nst / 4;
    for (int i = 0; i < 4; ++i) {
        double acc = 0.0;
        for (int t = n4 * i; t < n4 * i + n4; ++t) { acc = fma(a->v[t], b->v[t], acc); acc = fma(a->w[t], b->w[t], acc); }
        q[i] = acc;
    }
    return (q[0] + q[1]) + (q[2] + q[3]);
}

/* ------------------------------------------------------------------------------------------ */
/* per-instance data derived from p once per solve                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int N, P, nobs, ndyn;
    double ts, inv_ts;
    double vmin, vmax, wmax, amin, amax, awmax;
    /* p[0:20]  (mpc_generator.py:73-75) */
    double x0, y0, th0, vinit, winit, xf, yf, thf;
    double q, qv, qth, rv, rw, qN, qthN, qcte, pa, pw;
    double vref[MAXP];                              /* p[nz+t]                   (:85)      */
    double xs[MAXOBS], ys[MAXOBS], r2[MAXOBS];      /* static circles            (:93-95)   */
    double ex[MAXDYN][MAXP], ey[MAXDYN][MAXP];      /* ellipse centres per stage (:100-101) */
    double ca[MAXDYN][MAXP], sa[MAXDYN][MAXP];      /* cos/sin of As             (:104,118) */
    double irx2[MAXDYN][MAXP], iry2[MAXDYN][MAXP];  /* 1/rx^2, 1/ry^2            (:102-103) */
    double s1x[MAXP], s1y[MAXP], sdx[MAXP], sdy[MAXP], sinv[MAXP]; /* segments    (:127-136) */
} inst_t;

int orc_n_u(const orc_problem *pb) { return 2 * pb->N; }
int orc_n1(const orc_problem *pb) { return 2 * pb->N; }
int orc_n2(const orc_problem *pb) { return pb->nobs + pb->ndyn; }
int orc_n_p(const orc_problem *pb)
{   /* mpc_generator.py:71 with nz=20, nobs=3, ndynobs=5, nx=3 */
    return NZ + pb->N + 3 * pb->nobs + 5 * pb->ndyn * pb->N + 3 * pb->N;
}

void orc_default_opts(orc_opts *o)
{
    o->tolerance = 1e-4;            /* mpc_generator.py:185 */
    o->initial_tolerance = 1e-4;
    o->delta_tolerance = 1e-4;
    o->initial_penalty = 1.0;
    o->penalty_update = 5.0;
    o->tolerance_update = 0.1;
    o->sufficient_decrease = 0.1;
    o->lbfgs_memory = 10;
    o->max_inner = 500;
    o->max_outer = 10;
    o->max_total_inner = 0;
    o->akkt_gradient = 1;
    o->ls_failure = 0;
    o->inner_status = 0;
    o->reserved = 0;
}

static int check_problem(const orc_problem *pb)
{
    if (pb->N < 2 || pb->N > MAXN) return -1;
    if (pb->nobs < 0 || pb->nobs > MAXOBS) return -2;
    if (pb->ndyn < 0 || pb->ndyn > MAXDYN) return -3;
    if (!(pb->ts > 0.0)) return -4;
    return 0;
}

static void prepare(const orc_problem *pb, const double *p, inst_t *I)
{
    const int N = pb->N;
    memset(I, 0, sizeof(*I));
    I->N = N;
    I->P = horizon_pad(N);
    I->nobs = pb->nobs;
    I->ndyn = pb->ndyn;
    I->ts = pb->ts;
    I->inv_ts = 1.0 / pb->ts;
    I->vmin = pb->vmin; I->vmax = pb->vmax; I->wmax = pb->wmax;
    I->amin = pb->amin; I->amax = pb->amax; I->awmax = pb->awmax;
    I->x0 = p[0]; I->y0 = p[1]; I->th0 = p[2]; I->vinit = p[3]; I->winit = p[4];
    I->xf = p[5]; I->yf = p[6]; I->thf = p[7];          /* p[8:10] unused by the cost (:74) */
    I->q = p[10]; I->qv = p[11]; I->qth = p[12]; I->rv = p[13]; I->rw = p[14];
    I->qN = p[15]; I->qthN = p[16]; I->qcte = p[17]; I->pa = p[18]; I->pw = p[19];
    for (int t = 0; t < N; ++t) I->vref[t] = p[NZ + t];
    const double *ps = p + NZ + N;
    for (int k = 0; k < pb->nobs; ++k) {
        I->xs[k] = ps[3 * k];
        I->ys[k] = ps[3 * k + 1];
        I->r2[k] = ps[3 * k + 2] * ps[3 * k + 2];
    }
    const double *pd = ps + 3 * pb->nobs;              /* obstacle-major, stage-minor (:97-104) */
    for (int k = 0; k < pb->ndyn; ++k)
        for (int t = 0; t < N; ++t) {
            const double *e = pd + (k * N + t) * 5;
            I->ex[k][t] = e[0];
            I->ey[k][t] = e[1];
            I->irx2[k][t] = 1.0 / (e[2] * e[2]);
            I->iry2[k][t] = 1.0 / (e[3] * e[3]);
            orc_sincos(e[4], &I->sa[k][t], &I->ca[k][t]);
        }
    const double *pr = pd + 5 * pb->ndyn * N;          /* base (:79) ; (x, y, theta) per sample */
    for (int i = 1; i < N; ++i) {                      /* segment i joins sample i-1 and i (:128-134) */
        double ax = pr[3 * (i - 1)], ay = pr[3 * (i - 1) + 1];
        double bx = pr[3 * i], by = pr[3 * i + 1];
        double dx = bx - ax, dy = by - ay;
        I->s1x[i - 1] = ax; I->s1y[i - 1] = ay;
        I->sdx[i - 1] = dx; I->sdy[i - 1] = dy;
        I->sinv[i - 1] = 1.0 / (fma(dx, dx, dy * dy) + 1e-16);     /* (:136) */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* psi, grad psi, F1, F2                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    double psi;
    hvec g;                 /* gradient                                  */
    double av[MAXP], aw[MAXP];   /* F1 = [acc ; omega_acc]   (:160-162)  */
    double F2[MAXOBS + MAXDYN];  /* (:119)                                */
} eval_out;

/* y may be NULL (zeros).  want_grad = 0 skips the adjoint sweep. */
static void eval_psi(const inst_t *I, const hvec *u, double c, const hvec *y, int want_grad,
                     eval_out *o)
{
    const int N = I->N, P = I->P, n2 = I->nobs + I->ndyn;
    const double ts = I->ts, inv_ts = I->inv_ts;
    double th[MAXP], thn[MAXP], sn[MAXP], cs[MAXP], xn[MAXP], yn[MAXP], xp[MAXP], yp[MAXP];
    double l[MAXP], sv[MAXP], sw[MAXP], dvv[MAXP];
    int amin_seg[MAXP];
    double tmp[MAXP], tmp2[MAXP];

    /* rollout (:88-90) as three prefix sums */
    for (int j = 0; j < P; ++j) tmp[j] = u->w[j];
    scan_prefix(tmp, P, N);
    for (int j = 0; j < P; ++j) thn[j] = fma(ts, tmp[j], I->th0);
    for (int j = 0; j < P; ++j) th[j] = j == 0 ? I->th0 : thn[j - 1];
    for (int j = 0; j < P; ++j) orc_sincos(th[j], &sn[j], &cs[j]);
    for (int j = 0; j < P; ++j) { tmp[j] = u->v[j] * cs[j]; tmp2[j] = u->v[j] * sn[j]; }
    scan_prefix(tmp, P, N);
    scan_prefix(tmp2, P, N);
    for (int j = 0; j < P; ++j) { xn[j] = fma(ts, tmp[j], I->x0); yn[j] = fma(ts, tmp2[j], I->y0); }
    for (int j = 0; j < P; ++j) { xp[j] = j == 0 ? I->x0 : xn[j - 1]; yp[j] = j == 0 ? I->y0 : yn[j - 1]; }

    const double cbar_inv = 1.0 / dmax(c, 1.0);
    const double half_c = 0.5 * c;
    for (int t = 0; t < P; ++t) {
        if (t >= N) { l[t] = 0.0; o->av[t] = o->aw[t] = sv[t] = sw[t] = 0.0; continue; }
        const double v = u->v[t], w = u->w[t];
        double acc = (I->rv * v) * v;                                    /* (:84) */
        acc = fma(I->rw * w, w, acc);
        const double dv = v - I->vref[t];                                /* (:85) */
        dvv[t] = dv;
        acc = fma(I->qv * dv, dv, acc);
        const double ddx = xp[t] - I->xf, ddy = yp[t] - I->yf, dth = th[t] - I->thf;   /* (:86,59-64) */
        acc = fma(I->q, fma(ddx, ddx, ddy * ddy), acc);
        acc = fma(I->qth * dth, dth, acc);
        /* cross-track error: min over the N-1 reference segments (:121-144) */
        double best = INFINITY;
        int bi = 0;
        for (int i = 0; i < N - 1; ++i) {
            const double px = xn[t] - I->s1x[i], py = yn[t] - I->s1y[i];
            const double dot = fma(px, I->sdx[i], py * I->sdy[i]);
            const double that = dot * I->sinv[i];
            const double tst = dmin(dmax(that, 0.0), 1.0);               /* (:138) */
            const double ex = fma(tst, I->sdx[i], -px), ey = fma(tst, I->sdy[i], -py);
            const double d2 = fma(ex, ex, ey * ey);                      /* (:142) */
            if (d2 < best) { best = d2; bi = i; }
        }
        amin_seg[t] = bi;
        acc = fma(I->qcte, best, acc);                                   /* (:144) */
        /* accelerations (:160-161) and their cost (:170-171) */
        const double vprev = t == 0 ? I->vinit : u->v[t - 1];
        const double wprev = t == 0 ? I->winit : u->w[t - 1];
        const double av = (v - vprev) * inv_ts, aw = (w - wprev) * inv_ts;
        o->av[t] = av; o->aw[t] = aw;
        acc = fma(I->pa * av, av, acc);
        acc = fma(I->pw * aw, aw, acc);
        /* ALM term  c/2 * dist^2_C(F1 + y/max(c,1))    */
        const double yv = y ? y->v[t] : 0.0, yw = y ? y->w[t] : 0.0;
        const double tv = fma(yv, cbar_inv, av), tw = fma(yw, cbar_inv, aw);
        sv[t] = tv - clampd(tv, I->amin, I->amax);
        sw[t] = tw - clampd(tw, -I->awmax, I->awmax);
        acc = fma(half_c, fma(sv[t], sv[t], sw[t] * sw[t]), acc);
        if (t == N - 1) {                                                /* terminal (:148) */
            const double tx = xn[t] - I->xf, ty = yn[t] - I->yf, tth = thn[t] - I->thf;
            acc = fma(I->qN, fma(tx, tx, ty * ty), acc);
            acc = fma(I->qthN * tth, tth, acc);
        }
        l[t] = acc;
    }
    const double fsum = tree_sum_p(l, P);

    /* obstacle penalties on the post-update state (:106-119): F2_k = sum_t max(0, h_kt) */
    for (int k = 0; k < I->nobs; ++k) {
        for (int t = 0; t < P; ++t) {
            if (t >= N) { tmp[t] = 0.0; continue; }
            const double dx = xn[t] - I->xs[k], dy = yn[t] - I->ys[k];
            const double h = fma(-dy, dy, fma(-dx, dx, I->r2[k]));       /* (:112) */
            tmp[t] = dmax(h, 0.0);
        }
        o->F2[k] = tree_sum_p(tmp, P);
    }
    for (int k = 0; k < I->ndyn; ++k) {
        for (int t = 0; t < P; ++t) {
            if (t >= N) { tmp[t] = 0.0; continue; }
            const double dx = xn[t] - I->ex[k][t], dy = yn[t] - I->ey[k][t];
            const double a = fma(dx, I->ca[k][t], dy * I->sa[k][t]);
            const double b = fma(dx, I->sa[k][t], -(dy * I->ca[k][t]));
            const double h = fma(-(b * b), I->iry2[k][t], fma(-(a * a), I->irx2[k][t], 1.0)); /* (:118) */
            tmp[t] = dmax(h, 0.0);
        }
        o->F2[I->nobs + k] = tree_sum_p(tmp, P);
    }
    double pen = 0.0;
    for (int k = 0; k < n2; ++k) pen = fma(o->F2[k], o->F2[k], pen);
    o->psi = fma(half_c, pen, fsum);
    if (!want_grad) return;

    /* ---- adjoint sweep: what CasADi reverse AD did for the reference ---- */
    double Gx[MAXP], Gy[MAXP], Gt[MAXP], qa[MAXP], qw[MAXP];
    const double two_qcte = 2.0 * I->qcte;
    for (int t = 0; t < P; ++t) {
        if (t >= N) { Gx[t] = Gy[t] = Gt[t] = qa[t] = qw[t] = 0.0; continue; }
        /* cross-track error through the arg-min segment */
        const int i = amin_seg[t];
        const double px = xn[t] - I->s1x[i], py = yn[t] - I->s1y[i];
        const double dot = fma(px, I->sdx[i], py * I->sdy[i]);
        const double that = dot * I->sinv[i];
        const double tst = dmin(dmax(that, 0.0), 1.0);
        const double ex = fma(tst, I->sdx[i], -px), ey = fma(tst, I->sdy[i], -py);
        const double ed = fma(ex, I->sdx[i], ey * I->sdy[i]);
        const double m = (that > 0.0 && that < 1.0) ? ed * I->sinv[i] : 0.0;
        double gx = two_qcte * fma(m, I->sdx[i], -ex);
        double gy = two_qcte * fma(m, I->sdy[i], -ey);
        /* obstacle penalties: c * F2_k * dh_kt/d(x,y) where h_kt > 0 */
        for (int k = 0; k < I->nobs; ++k) {
            const double wk = -2.0 * (c * o->F2[k]);
            const double dx = xn[t] - I->xs[k], dy = yn[t] - I->ys[k];
            const double h = fma(-dy, dy, fma(-dx, dx, I->r2[k]));
            if (h > 0.0) { gx = fma(wk, dx, gx); gy = fma(wk, dy, gy); }
        }
        for (int k = 0; k < I->ndyn; ++k) {
            const double wk = -2.0 * (c * o->F2[I->nobs + k]);
            const double dx = xn[t] - I->ex[k][t], dy = yn[t] - I->ey[k][t];
            const double a = fma(dx, I->ca[k][t], dy * I->sa[k][t]);
            const double b = fma(dx, I->sa[k][t], -(dy * I->ca[k][t]));
            const double h = fma(-(b * b), I->iry2[k][t], fma(-(a * a), I->irx2[k][t], 1.0));
            if (h > 0.0) {
                const double A = a * I->irx2[k][t], B = b * I->iry2[k][t];
                const double hx = fma(A, I->ca[k][t], B * I->sa[k][t]);
                const double hy = fma(A, I->sa[k][t], -(B * I->ca[k][t]));
                gx = fma(wk, hx, gx);
                gy = fma(wk, hy, gy);
            }
        }
        /* the post-update state of stage t is the tracked state of stage t+1 (:86), or the
         * terminal state (:148) */
        const double wq = t < N - 1 ? I->q : I->qN, wth = t < N - 1 ? I->qth : I->qthN;
        gx = fma(2.0 * wq, xn[t] - I->xf, gx);
        gy = fma(2.0 * wq, yn[t] - I->yf, gy);
        Gx[t] = gx; Gy[t] = gy;
        Gt[t] = (2.0 * wth) * (thn[t] - I->thf);
        qa[t] = fma(c, sv[t], (2.0 * I->pa) * o->av[t]);     /* d psi / d acc_t       */
        qw[t] = fma(c, sw[t], (2.0 * I->pw) * o->aw[t]);     /* d psi / d omega_acc_t */
    }
    scan_suffix(Gx, P, N);
    scan_suffix(Gy, P, N);
    double Dt[MAXP];
    for (int t = 0; t < P; ++t) {
        const double e = fma(Gy[t], cs[t], -(Gx[t] * sn[t]));
        Dt[t] = t < N ? (ts * u->v[t]) * e : 0.0;
    }
    for (int t = 0; t < P; ++t) tmp[t] = t < N ? Gt[t] + (t + 1 < P ? Dt[t + 1] : 0.0) : 0.0;
    scan_suffix(tmp, P, N);
    for (int t = 0; t < P; ++t) {
        if (t >= N) { o->g.v[t] = o->g.w[t] = 0.0; continue; }
        const double qan = t + 1 < P ? qa[t + 1] : 0.0, qwn = t + 1 < P ? qw[t + 1] : 0.0;
        const double dyn = fma(Gx[t], cs[t], Gy[t] * sn[t]);
        double gv = fma(2.0 * I->rv, u->v[t], (2.0 * I->qv) * dvv[t]);
        gv = fma(inv_ts, qa[t] - qan, gv);
        o->g.v[t] = fma(ts, dyn, gv);
        double gw = (2.0 * I->rw) * u->w[t];
        gw = fma(inv_ts, qw[t] - qwn, gw);
        o->g.w[t] = fma(ts, tmp[t], gw);
    }
}

static void load_hvec(hvec *h, const double *flat, int N, int interleaved)
{
    memset(h, 0, sizeof(*h));
    if (!flat) return;
    for (int t = 0; t < N; ++t) {
        if (interleaved) { h->v[t] = flat[2 * t]; h->w[t] = flat[2 * t + 1]; }  /* u: (v0,w0,v1,w1..) (:83,157-158) */
        else { h->v[t] = flat[t]; h->w[t] = flat[N + t]; }                    /* F1/y: [acc ; omega_acc] (:162) */
    }
}

static void store_hvec(const hvec *h, double *flat, int N, int interleaved)
{
    if (!flat) return;
    for (int t = 0; t < N; ++t) {
        if (interleaved) { flat[2 * t] = h->v[t]; flat[2 * t + 1] = h->w[t]; }
        else { flat[t] = h->v[t]; flat[N + t] = h->w[t]; }
    }
}

int orc_eval(const orc_problem *pb, const double *p, const double *u, double c, const double *y,
             double *psi, double *grad, double *F1, double *F2)
{
    int rc = check_problem(pb);
    if (rc) return rc;
    inst_t *I = (inst_t *)malloc(sizeof(inst_t));
    eval_out o;
    hvec hu, hy;
    prepare(pb, p, I);
    load_hvec(&hu, u, pb->N, 1);
    load_hvec(&hy, y, pb->N, 0);
    eval_psi(I, &hu, c, y ? &hy : NULL, grad != NULL, &o);
    if (psi) *psi = o.psi;
    store_hvec(&o.g, grad, pb->N, 1);
    if (F1) for (int t = 0; t < pb->N; ++t) { F1[t] = o.av[t]; F1[pb->N + t] = o.aw[t]; }
    if (F2) for (int k = 0; k < pb->nobs + pb->ndyn; ++k) F2[k] = o.F2[k];
    free(I);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* PANOC inner solver                                                    */
/* ------------------------------------------------------------------------------------------ */
#define GAMMA_L_COEFF 0.95
#define DELTA_LIPSCHITZ 1e-12
#define EPSILON_LIPSCHITZ 1e-6
#define LIPSCHITZ_UPDATE_EPSILON 1e-6
#define MAX_LIPSCHITZ_UPDATE_ITERATIONS 10
#define MAX_LIPSCHITZ_CONSTANT 1e9
#define MIN_LIPSCHITZ_CONSTANT 1e-10
#define MAX_LINESEARCH_ITERATIONS 10
#define LBFGS_SY_EPSILON 1e-10
#define LBFGS_CBFGS_EPSILON 1e-8     /* with cbfgs alpha = 1 */

#define GRAM_M 10      /* pairs the Gram form carries (the hybrid kernel's ring): ages >= m, and inactive ages, are exactly zero */
/* stages the quarter dot runs over (zero padded): 20 for N <= 20 (nmpc_solve_hyb.h), 40 for 20 < N <= 40 (nmpc_solve_hyb2.h) */
#define GRAM_NST(N) ((N) <= 20 ? 20 : 40)
typedef struct {
    int m, active, first_old;
    int gram;                      /* stages the quarter dots run over (GRAM_NST): what nmpc_solve_hyb.h / nmpc_solve_hyb2.h compute */
    hvec S[MAXMEM], Y[MAXMEM];     /* index 0 = newest */
    double rho[MAXMEM];
    double H0;
    hvec old_s, old_g;
    /* Gram form, by age: SY[a][b] = <s_a, y_b> for a OLDER than b (a > b), zero otherwise; YY[a][b] = <y_a, y_b> */
    double SY[GRAM_M][GRAM_M], YY[GRAM_M][GRAM_M];
} lbfgs_t;

typedef struct {
    hvec g, gs, uh, r, d, up, gprev;   /* gradient, gradient step, half step, gamma*fpr, direction, u_plus */
    hvec gk;                           /* gradient at the current iterate, kept through the line search (ls_failure = 1) */
    double cost, L, gamma, hig, sigma, nr2, norm_r, tau;      /* hig = 0.5 / gamma, formed when gamma changes */
    int iteration;
    lbfgs_t lb;
    uint32_t n_cost, n_grad;
    uint32_t passes3, passes6;         /* diagnostic: evaluation passes a 3- / 6-points-per-pass schedule would need */
    eval_out scratch;
} panoc_t;

static void project_U(const inst_t *I, hvec *x)
{
    for (int t = 0; t < I->N; ++t) {
        x->v[t] = clampd(x->v[t], I->vmin, I->vmax);
        x->w[t] = clampd(x->w[t], -I->wmax, I->wmax);
    }
}

/* gs = x - gamma*g ; uh = Pi_U(gs) */
static void grad_and_half_step(const inst_t *I, panoc_t *c, const hvec *x)
{
    for (int t = 0; t < I->P; ++t) {
        c->gs.v[t] = fma(-c->gamma, c->g.v[t], x->v[t]);
        c->gs.w[t] = fma(-c->gamma, c->g.w[t], x->w[t]);
    }
    c->uh = c->gs;
    project_U(I, &c->uh);
}

static void compute_fpr(const inst_t *I, panoc_t *c, const hvec *u)
{
    for (int t = 0; t < I->P; ++t) { c->r.v[t] = u->v[t] - c->uh.v[t]; c->r.w[t] = u->w[t] - c->uh.w[t]; }
    c->nr2 = qdot(&c->r, &c->r, c->lb.gram);
    c->norm_r = sqrt(c->nr2);
}

static void lbfgs_reset(lbfgs_t *lb)
{
    lb->active = 0; lb->first_old = 1;
    /* the Gram form runs over all GRAM_M ages every time: what is not active is exactly zero */
    memset(lb->S, 0, sizeof(hvec) * GRAM_M); memset(lb->Y, 0, sizeof(hvec) * GRAM_M);
    memset(lb->rho, 0, sizeof(double) * GRAM_M);
    memset(lb->SY, 0, sizeof(lb->SY)); memset(lb->YY, 0, sizeof(lb->YY));
}

/* the pair (s, y), <y, s> = ys, enters the buffer as its newest (age 0); every other pair ages by one, the oldest leaves */
static void lbfgs_push(lbfgs_t *lb, const hvec *s, const hvec *y, double ys)
{
    for (int k = lb->m - 1; k > 0; --k) { lb->S[k] = lb->S[k - 1]; lb->Y[k] = lb->Y[k - 1]; lb->rho[k] = lb->rho[k - 1]; }
    lb->S[0] = *s;
    lb->Y[0] = *y;
    lb->rho[0] = 1.0 / ys;
    /* the new pair's column of SY and row / column of YY are measured against the pairs that stay.
     * (Each entry is a function of two stored vectors only, so keeping it equals recomputing it.) */
    for (int a = lb->m - 1; a > 0; --a)
        for (int b = lb->m - 1; b > 0; --b) { lb->SY[a][b] = lb->SY[a - 1][b - 1]; lb->YY[a][b] = lb->YY[a - 1][b - 1]; }
    for (int a = 1; a < lb->m; ++a) {
        lb->SY[a][0] = qdot(&lb->S[a], y, lb->gram);
        lb->SY[0][a] = 0.0;
        lb->YY[a][0] = lb->YY[0][a] = qdot(&lb->Y[a], y, lb->gram);
    }
    lb->SY[0][0] = 0.0;
    lb->YY[0][0] = qdot(y, y, lb->gram);
    lb->H0 = ys / lb->YY[0][0];
    if (lb->active < lb->m) lb->active++;
}

/* lbfgs crate: update_hessian(g := gamma_fpr, s := u) with sy-epsilon and C-BFGS safeguards */
static void lbfgs_update(const inst_t *I, lbfgs_t *lb, const hvec *r, const hvec *u, double norm_r)
{
    const int P = I->P;
    if (lb->first_old) { lb->first_old = 0; lb->old_s = *u; lb->old_g = *r; return; }
    hvec s, y;
    for (int t = 0; t < P; ++t) {
        s.v[t] = u->v[t] - lb->old_s.v[t]; s.w[t] = u->w[t] - lb->old_s.w[t];
        y.v[t] = r->v[t] - lb->old_g.v[t]; y.w[t] = r->w[t] - lb->old_g.w[t];
    }
    const double ys = qdot(&s, &y, lb->gram), ss = qdot(&s, &s, lb->gram);
    if (ss <= DBL_MIN || ys <= LBFGS_SY_EPSILON) return;
    if (!(ys > (LBFGS_CBFGS_EPSILON * norm_r) * ss)) return;      /* C-BFGS: <y, s> / ||s||^2 > eps ||r||^alpha, alpha = 1 */
    lb->old_s = *u;
    lb->old_g = *r;
    lbfgs_push(lb, &s, &y, ys);
}

/* d := H d in the Gram form (nmpc_solve_hyb.h, nmpc_solve_hyb2.h): the coefficients of the two-loop recursion,
 *   alpha_j = rho_j <s_j, q_j>,  beta_j = rho_j <y_j, z_j>,
 * come out of two recurrences over the inner products <s_k, r>, <y_k, r>, SY, YY instead of twenty dependent reductions over the
 * horizon; the vector updates are those of the two-loop recursion, in its order.  All GRAM_M ages take part every time (what is
 * not active is zero).  SY is strictly lower triangular by age, so entry k of a1 is final once step k has used it, and entry k
 * of a2 once step k of the second recurrence has. */
static void lbfgs_apply_gram(const inst_t *I, const lbfgs_t *lb, hvec *d)
{
    const int P = I->P;
    double a1[GRAM_M], a2[GRAM_M], alv[GRAM_M];
    if (lb->active == 0) return;
    const hvec r = *d;
    for (int k = 0; k < GRAM_M; ++k) { a1[k] = qdot(&lb->S[k], &r, lb->gram); a2[k] = qdot(&lb->Y[k], &r, lb->gram); }
    for (int j = 0; j < GRAM_M; ++j) {
        const double al = lb->rho[j] * a1[j];
        for (int k = 0; k < GRAM_M; ++k) { a1[k] = fma(-al, lb->SY[k][j], a1[k]); a2[k] = fma(-al, lb->YY[k][j], a2[k]); }
        for (int t = 0; t < P; ++t) { d->v[t] = fma(-al, lb->Y[j].v[t], d->v[t]); d->w[t] = fma(-al, lb->Y[j].w[t], d->w[t]); }
    }
    for (int k = 0; k < GRAM_M; ++k) { alv[k] = lb->rho[k] * a1[k]; a2[k] = lb->H0 * a2[k]; }
    for (int t = 0; t < P; ++t) { d->v[t] = lb->H0 * d->v[t]; d->w[t] = lb->H0 * d->w[t]; }
    for (int j = GRAM_M - 1; j >= 0; --j) {
        const double be = lb->rho[j] * a2[j];
        const double ab = alv[j] - be;
        for (int k = 0; k < GRAM_M; ++k) a2[k] = fma(ab, lb->SY[j][k], a2[k]);
        for (int t = 0; t < P; ++t) { d->v[t] = fma(ab, lb->S[j].v[t], d->v[t]); d->w[t] = fma(ab, lb->S[j].w[t], d->w[t]); }
    }
}

/* test hook (tests/test_oracle_solver.py compares with a two-loop recursion written in numpy): `npush` pairs (s, y) -- [npush][2 N],
 * (v, w) interleaved by stage, oldest first -- enter an empty buffer of memory m without the safeguards; d_io: in r, out H r */
int orc_test_lbfgs_gram(int N, int m, int npush, const double *s_list, const double *y_list, double *d_io)
{
    if (N < 2 || N > MAXN || m < 1 || m > GRAM_M || npush < 0) return -1;
    inst_t *I = (inst_t *)calloc(1, sizeof(inst_t));
    lbfgs_t *lb = (lbfgs_t *)calloc(1, sizeof(lbfgs_t));
    I->N = N; I->P = horizon_pad(N);
    lb->m = m; lb->gram = GRAM_NST(N);
    lbfgs_reset(lb);
    for (int k = 0; k < npush; ++k) {
        hvec s, y;
        memset(&s, 0, sizeof s); memset(&y, 0, sizeof y);
        for (int t = 0; t < N; ++t) {
            s.v[t] = s_list[(size_t)k * 2 * N + 2 * t]; s.w[t] = s_list[(size_t)k * 2 * N + 2 * t + 1];
            y.v[t] = y_list[(size_t)k * 2 * N + 2 * t]; y.w[t] = y_list[(size_t)k * 2 * N + 2 * t + 1];
        }
        lbfgs_push(lb, &s, &y, qdot(&s, &y, lb->gram));
    }
    hvec d;
    memset(&d, 0, sizeof d);
    for (int t = 0; t < N; ++t) { d.v[t] = d_io[2 * t]; d.w[t] = d_io[2 * t + 1]; }
    lbfgs_apply_gram(I, lb, &d);
    for (int t = 0; t < N; ++t) { d_io[2 * t] = d.v[t]; d_io[2 * t + 1] = d.w[t]; }
    free(lb); free(I);
    return 0;
}

/* forward-backward envelope at the point whose cost/gradient/gs/uh are in the cache */
static double fbe(const inst_t *I, const panoc_t *c)
{
    double t[MAXP] = {0.0};
    for (int j = 0; j < I->P; ++j) {
        const double a = c->gs.v[j] - c->uh.v[j], b = c->gs.w[j] - c->uh.w[j];
        t[j] = fma(a, a, b * b);
    }
    const double dist2 = tree_sum_p(t, I->P);
    const double gg = hdot(&c->g, &c->g, I->P);
    return c->cost - (0.5 * c->gamma) * gg + dist2 * c->hig;
}

static void do_eval(const inst_t *I, panoc_t *c, const hvec *x, double pen, const hvec *y, int want_grad,
                    eval_out *o)
{
    eval_psi(I, x, pen, y, want_grad, o);
    if (want_grad) c->n_grad++; else c->n_cost++;
}

/* returns exit status (0 converged / 1 iterations / 2 out of budget); u in/out; iters and norm_fpr reported.
 * budget_left: PANOC iterations this inner solve may still spend (0 = unlimited), opts->max_total_inner. */
static int panoc_solve(const inst_t *I, const orc_opts *opts, panoc_t *c, hvec *u, double pen, const hvec *y,
                       double tol, double akkt_tol, int max_iter, uint32_t budget_left, uint32_t *iters,
                       double *cost_out)
{
    const int P = I->P, N = I->N;
    eval_out *o = &c->scratch;
    int timed_out = 0;
    /* ---- init ---- */
    lbfgs_reset(&c->lb);
    c->iteration = 0;
    c->tau = 1.0;
    do_eval(I, c, u, pen, y, 1, o);
    c->cost = o->psi;
    c->g = o->g;
    {   /* local Lipschitz estimate of grad psi at u: h_i = max(1e-6 u_i, 1e-12) */
        hvec uh2 = *u, dg;
        double t[MAXP];
        for (int j = 0; j < P; ++j) {
            if (j >= N) { t[j] = 0.0; continue; }
            const double hv = EPSILON_LIPSCHITZ * u->v[j] > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * u->v[j] : DELTA_LIPSCHITZ;
            const double hw = EPSILON_LIPSCHITZ * u->w[j] > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * u->w[j] : DELTA_LIPSCHITZ;
            uh2.v[j] = u->v[j] + hv;
            uh2.w[j] = u->w[j] + hw;
            t[j] = fma(hv, hv, hw * hw);
        }
        const double norm_h = sqrt(tree_sum_p(t, P));
        do_eval(I, c, &uh2, pen, y, 1, o);
        for (int j = 0; j < P; ++j) { dg.v[j] = o->g.v[j] - c->g.v[j]; dg.w[j] = o->g.w[j] - c->g.w[j]; }
        c->L = sqrt(hdot(&dg, &dg, P)) / norm_h;
    }
    c->gamma = GAMMA_L_COEFF / dmax(c->L, MIN_LIPSCHITZ_CONSTANT);
    c->hig = 0.5 / c->gamma;
    c->sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * c->gamma);
    grad_and_half_step(I, c, u);
    c->passes3++; c->passes6++;       /* u and u + h in one pass */

    /* ---- iterations ---- */
    uint32_t num_iter = 0;
    for (;;) {
        /* step(): returns "continue" */
        compute_fpr(I, c, u);
        if (c->norm_r < tol) {           /* fpr test, then the AKKT test (short-circuit) */
            if (opts->akkt_gradient == 2) break;                 /* no AKKT test */
            if (opts->akkt_gradient == 1 && c->iteration >= 1) {
                /* grad_prev was copied from grad at the top of this step: the difference is exactly zero and the residual is ||r|| / gamma */
                if (c->norm_r < akkt_tol * c->gamma) break;
            } else {
            double t[MAXP];
            for (int j = 0; j < P; ++j) {
                double a, b;
                if (opts->akkt_gradient == 1) {
                    /* iteration 0: grad_prev is still the zero vector */
                    a = c->r.v[j] / c->gamma + c->g.v[j];
                    b = c->r.w[j] / c->gamma + c->g.w[j];
                } else {
                    a = c->r.v[j] / c->gamma + (c->g.v[j] - c->gprev.v[j]);
                    b = c->r.w[j] / c->gamma + (c->g.w[j] - c->gprev.w[j]);
                }
                t[j] = fma(a, a, b * b);
            }
            if (sqrt(tree_sum_p(t, P)) < akkt_tol) break;
            }
        }
        /* Lipschitz / gamma backtracking */
        do_eval(I, c, &c->uh, pen, y, 0, o);
        double cost_uh = o->psi;
        int n_back = 0, n_trials = 0;
        for (int it = 0; it < MAX_LIPSCHITZ_UPDATE_ITERATIONS && c->L < MAX_LIPSCHITZ_CONSTANT; ++it) {
            const double rhs = c->cost + LIPSCHITZ_UPDATE_EPSILON * fabs(c->cost) - qdot(&c->g, &c->r, c->lb.gram)
                             + (GAMMA_L_COEFF / (2.0 * c->gamma)) * c->nr2;
            if (!(cost_uh > rhs)) break;
            lbfgs_reset(&c->lb);
            c->L *= 2.0;
            c->gamma /= 2.0;
            c->hig = 0.5 / c->gamma;
            grad_and_half_step(I, c, u);
            do_eval(I, c, &c->uh, pen, y, 0, o);
            cost_uh = o->psi;
            compute_fpr(I, c, u);
            n_back++;
        }
        c->sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * c->gamma);
        /* L-BFGS buffer update and direction */
        lbfgs_update(I, &c->lb, &c->r, u, c->norm_r);
        if (c->iteration > 0) { c->d = c->r; lbfgs_apply_gram(I, &c->lb, &c->d); }
        if (c->iteration == 0) {
            /* first iteration: plain forward-backward step */
            *u = c->uh;
            do_eval(I, c, u, pen, y, 1, o);
            c->cost = o->psi;
            c->g = o->g;
            grad_and_half_step(I, c, u);
        } else {
            const double rhs_ls = fbe(I, c) - c->sigma * c->nr2;
            int exhausted = 0;
            c->gk = c->g;
            c->tau = 1.0;
            for (int n = 0;; ++n) {
                const double omt = 1.0 - c->tau;
                for (int j = 0; j < P; ++j) {
                    c->up.v[j] = fma(-c->tau, c->d.v[j], fma(-omt, c->r.v[j], u->v[j]));
                    c->up.w[j] = fma(-c->tau, c->d.w[j], fma(-omt, c->r.w[j], u->w[j]));
                }
                c->gprev = c->g;                 /* akkt_gradient = 0: cached before every overwrite */
                n_trials++;
                do_eval(I, c, &c->up, pen, y, 1, o);
                c->cost = o->psi;
                c->g = o->g;
                grad_and_half_step(I, c, &c->up);
                if (!(fbe(I, c) > rhs_ls)) break;
                if (n >= MAX_LINESEARCH_ITERATIONS) { exhausted = opts->ls_failure == 1; break; }
                c->tau /= 2.0;
            }
            if (exhausted) {
                /* tau = 0: u+ = u_bar, the forward-backward step from the current iterate */
                c->tau = 0.0;
                c->g = c->gk;
                grad_and_half_step(I, c, u);
                *u = c->uh;
                do_eval(I, c, u, pen, y, 1, o);
                c->cost = o->psi;
                c->g = o->g;
                grad_and_half_step(I, c, u);
            } else {
                *u = c->up;
            }
        }
        {   /* diagnostic pass model: the first pass of an iteration holds u_bar and the first k - 1 trials; every
             * Lipschitz back-off costs a pass of its own (and a pass for the trials that were thrown away) */
            const int first3 = n_back ? 0 : 2, first6 = n_back ? 0 : 5;
            c->passes3 += 1 + n_back + (n_trials > first3 ? (n_trials - first3 + 2) / 3 : 0);
            c->passes6 += 1 + n_back + (n_trials > first6 ? (n_trials - first6 + 5) / 6 : 0);
        }
        c->iteration++;
        /* OpEn: while step() && num_iter < max_iter { num_iter++ } */
        if (!(num_iter < (uint32_t)max_iter)) break;
        num_iter++;
        if (budget_left > 0 && num_iter >= budget_left) { timed_out = 1; break; }
    }
    const int status = timed_out ? 2 : (num_iter < (uint32_t)max_iter ? 0 : 1);
    *iters = num_iter;
    *cost_out = c->cost;
    *u = c->uh;      /* the feasible half step is what PANOC returns */
    return status;
}

/* ------------------------------------------------------------------------------------------ */
/* ALM / penalty outer loop                                              */
/* ------------------------------------------------------------------------------------------ */
static int vec_finite(const hvec *h, int N)
{
    for (int t = 0; t < N; ++t) if (!isfinite(h->v[t]) || !isfinite(h->w[t])) return 0;
    return 1;
}

int orc_solve(const orc_problem *pb, const orc_opts *opts, const double *p, double *u_io,
              const double *y0, double c0, double *y_out, orc_status *st)
{
    int rc = check_problem(pb);
    if (rc) return rc;
    if (opts->lbfgs_memory < 1 || opts->lbfgs_memory > GRAM_M) return -5;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int N = pb->N;
    inst_t *I = (inst_t *)malloc(sizeof(inst_t));
    panoc_t *pc = (panoc_t *)calloc(1, sizeof(panoc_t));
    eval_out *o = &pc->scratch;
    prepare(pb, p, I);
    const int P = I->P, n2 = pb->nobs + pb->ndyn;
    pc->lb.m = opts->lbfgs_memory;
    pc->lb.gram = GRAM_NST(N);
    hvec u, y, yplus;
    load_hvec(&u, u_io, N, 1);
    load_hvec(&y, y0, N, 0);
    yplus = y;
    double c = c0 > 0.0 ? c0 : opts->initial_penalty;
    double eps_nu = opts->initial_tolerance;
    double dy_norm = 0.0, f2_norm = 0.0, dy_norm_plus = DBL_MAX, f2_norm_plus = 0.0;
    double last_fpr = 0.0, last_cost = 0.0;
    uint32_t inner_total = 0, outer = 0;
    int exit_status = 0;
    const double SMALL = DBL_EPSILON;

    for (int nu = 0; nu < opts->max_outer; ++nu) {
        outer++;
        for (int t = 0; t < N; ++t) {                    /* y <- Pi_Y(y), Y = [-1e12, 1e12]^n1 */
            y.v[t] = clampd(y.v[t], -1e12, 1e12);
            y.w[t] = clampd(y.w[t], -1e12, 1e12);
        }
        uint32_t it = 0;
        const uint32_t budget = opts->max_total_inner > 0 ? (uint32_t)opts->max_total_inner : 0u;
        if (opts->akkt_gradient == 1) memset(&pc->gprev, 0, sizeof(pc->gprev));
        int inner_status = panoc_solve(I, opts, pc, &u, c, &y, opts->tolerance, eps_nu, opts->max_inner,
                                       budget ? budget - inner_total : 0u, &it, &last_cost);
        inner_total += it;
        last_fpr = pc->norm_r;
        if (!vec_finite(&u, N) || !isfinite(last_cost) || !isfinite(last_fpr)) { exit_status = 4; break; }
        /* F1, F2 at the inner solution; y+ = y + c (F1 - Pi_C(F1 + y/max(c,1))) */
        eval_psi(I, &u, c, &y, 0, o);
        pc->n_cost++;
        pc->passes3++; pc->passes6++;
        {
            double t[MAXP];
            const double cbar_inv = 1.0 / dmax(c, 1.0);
            for (int j = 0; j < P; ++j) {
                if (j >= N) { t[j] = 0.0; yplus.v[j] = yplus.w[j] = 0.0; continue; }
                const double tv = fma(y.v[j], cbar_inv, o->av[j]), tw = fma(y.w[j], cbar_inv, o->aw[j]);
                yplus.v[j] = fma(c, o->av[j] - clampd(tv, I->amin, I->amax), y.v[j]);
                yplus.w[j] = fma(c, o->aw[j] - clampd(tw, -I->awmax, I->awmax), y.w[j]);
                const double a = yplus.v[j] - y.v[j], b = yplus.w[j] - y.w[j];
                t[j] = fma(a, a, b * b);
            }
            dy_norm_plus = sqrt(tree_sum_p(t, P));
            double pen = 0.0;
            for (int k = 0; k < n2; ++k) pen = fma(o->F2[k], o->F2[k], pen);
            f2_norm_plus = sqrt(pen);
        }
        if (getenv("ORC_TRACE"))
            fprintf(stderr, "  outer %d: c=%g inner=%u status=%d fpr=%.3e cost=%.6g dy/c=%.3e f2=%.3e L=%.3e evals=%u/%u\n", nu, c, it,
                    inner_status, pc->norm_r, last_cost, dy_norm_plus / c, f2_norm_plus, pc->L, pc->n_cost, pc->n_grad);
        const int crit1 = nu > 0 && dy_norm_plus <= c * opts->delta_tolerance + SMALL;
        const int crit2 = n2 == 0 || f2_norm_plus <= opts->delta_tolerance + SMALL;
        const int crit3 = eps_nu <= opts->tolerance + SMALL;
        if (crit1 && crit2 && crit3) { exit_status = opts->inner_status == 1 ? 0 : inner_status; break; }
        const int stall = nu == 0 || (dy_norm_plus <= opts->sufficient_decrease * dy_norm + SMALL &&
                                      f2_norm_plus <= opts->sufficient_decrease * f2_norm + SMALL);
        if (!stall) c *= opts->penalty_update;
        eps_nu = dmax(opts->tolerance_update * eps_nu, opts->tolerance);
        y = yplus;
        dy_norm = dy_norm_plus;
        f2_norm = f2_norm_plus;
        if (nu == opts->max_outer - 1) exit_status = 1;
        else if (inner_status == 2) { exit_status = 2; break; }     /* the budget is spent: no further outer iteration */
    }
    store_hvec(&u, u_io, N, 1);
    store_hvec(&yplus, y_out, N, 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (st) {
        st->exit_status = exit_status;
        st->num_outer_iterations = outer;
        st->num_inner_iterations = inner_total;
        st->num_cost_evals = pc->n_cost;
        st->num_grad_evals = pc->n_grad;
        st->reserved = pc->passes3;
        if (getenv("ORC_PASSES")) fprintf(stderr, "ORC_PASSES %p %u %u %u\n", (const void *)p, inner_total, pc->passes3, pc->passes6);
        st->last_problem_norm_fpr = last_fpr;
        st->delta_y_norm_over_c = dy_norm_plus / c;
        st->f2_norm = f2_norm_plus;
        st->penalty = c;
        st->cost = last_cost;
        st->solve_time_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }
    free(pc); free(I);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const orc_problem *pb; const orc_opts *opts; int B;
    int *next;                   /* shared work queue: the batch is heavy-tailed, threads pull the next instance */
    const double *p; double *u; const double *y0; const double *c0; double *y_out; orc_status *st;
    int rc;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const int nu = orc_n_u(j->pb), np = orc_n_p(j->pb), n1 = orc_n1(j->pb);
    for (;;) {
        const int b = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (b >= j->B) break;
        int rc = orc_solve(j->pb, j->opts, j->p + (size_t)b * np, j->u + (size_t)b * nu,
                           j->y0 ? j->y0 + (size_t)b * n1 : NULL, j->c0 ? j->c0[b] : 0.0,
                           j->y_out ? j->y_out + (size_t)b * n1 : NULL, j->st ? j->st + b : NULL);
        if (rc) j->rc = rc;
    }
    return NULL;
}

int orc_solve_batch(const orc_problem *pb, const orc_opts *opts, int B, const double *p, double *u,
                    const double *y0, const double *c0, double *y_out, orc_status *st, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    int next = 0;
    for (int i = 0; i < threads; ++i) {
        jobs[i] = (job_t){pb, opts, B, &next, p, u, y0, c0, y_out, st, 0};
        if (threads == 1) worker(&jobs[i]);
        else pthread_create(&th[i], NULL, worker, &jobs[i]);
    }
    int rc = 0;
    for (int i = 0; i < threads; ++i) {
        if (threads > 1) pthread_join(th[i], NULL);
        if (jobs[i].rc) rc = jobs[i].rc;
    }
    return rc;
}
