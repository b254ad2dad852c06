// nmpc_kernels.hip -- batched NMPC solve on MI355X (gfx950): PANOC inner iteration + L-BFGS +
// ALM/penalty outer loop, with the diff-drive rollout, stage/terminal costs, cross-track error and
// circle/ellipse soft-constraint penalties evaluated per step, entirely on device in f64.
//
// What is restated (paths relative to the reference repo):
//   cost / constraints   src/mpc/mpc_generator.py:66-171           (eval_psi)
//   solver               OpEn's PANOC + ALM that src/mpc/mpc_generator.py:173-193 generates and
//                        :206 calls; algorithm per SURVEY.md Appendix C  (solve kernel state machine)
// This file is original CDNA4 code; nothing here is translated from OpEn's Rust or CasADi's C.
#include "nmpc_device.h"
#include "../../include/nmpc_solver.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef NMPC_WIN
#define NMPC_WIN 1      // half width of the cross-track window of the one-stage-per-lane kernels (eval_psi); 0 = always the full scan
#endif

namespace nmpc {

constexpr int NZ = 20;         // reference configs/default.yaml:35
constexpr int MAXMEM = 10;     // L-BFGS memory the kernel is built for
constexpr int NDYN_MAX = 3;    // Ndynobs the kernel is built for
constexpr int GRAM_LD = 11;    // row stride of the kept inner products gsy / gyy (doubles): with 10 the ten lanes of a column read hit 8 bank pairs, with 11 ten
constexpr int GRAM_NST = 20;   // stages the Gram-form L-BFGS of the hybrid kernel runs over (N_hor <= 20, zero padded)
constexpr int OBS_STRIDE = 4;  // doubles per static circle in LDS: xs ys r^2 r
constexpr int SEG_STRIDE = 5;  // doubles per reference segment in LDS (odd: the per-lane window gathers of eval_psi spread over all banks)
// team mode of the hybrid kernel (nmpc_solve_hyb.h): four waves per workgroup; a wave without work of its own evaluates
// line-search trials for its siblings.  Request = u, r, d by stage (3 x 24 pairs); one result area = three trials'
// gradients by stage (3 x 24 pairs) + their psi values
constexpr int TEAM_WAVES = 4;
constexpr int TEAM_REQ_DOUBLES = 3 * 24 * 2;
constexpr int TEAM_AREA_DOUBLES = 3 * 24 * 2 + 8;     // + psi[3], envelope[3]
constexpr int TEAM_CTL_INTS = 64;
// instances waiting for a wave (nmpc_solve_hyb.h): long = an outer criterion is still open after the outer iteration just finished, cold = all hold: the next outer iteration is the last (and short)
constexpr int NPOOLS = 2;
enum { POOL_LONG = 0, POOL_COLD = 1 };

// PANOC constants
constexpr double GAMMA_L_COEFF = 0.95;
constexpr double DELTA_LIPSCHITZ = 1e-12;
constexpr double EPSILON_LIPSCHITZ = 1e-6;
constexpr double LIPSCHITZ_UPDATE_EPSILON = 1e-6;
constexpr int MAX_LIPSCHITZ_UPDATE_ITERATIONS = 10;
constexpr double MAX_LIPSCHITZ_CONSTANT = 1e9;
constexpr double MIN_LIPSCHITZ_CONSTANT = 1e-10;
constexpr int MAX_LINESEARCH_ITERATIONS = 10;
constexpr double LBFGS_SY_EPSILON = 1e-10;
constexpr double LBFGS_CBFGS_EPSILON = 1e-8;

// LDS slice of one group (offsets in doubles)
struct LdsMap {
    int sc;      // 18 instance scalars: x0 y0 th0 vinit winit xf yf thf | q qv qth rv rw qN qthN qcte pa pw | vinit winit again, as an aligned pair
    int cw;      // CW_NCOEF sin/cos polynomial coefficients (nmpc_device.h)
    int par;     // up to 24 parked solver scalars (hybrid kernel)
    int seg;     // SEG_STRIDE = 5 per reference segment (40 B): s1x s1y dx dy 1/(|d|^2 + 1e-16)
    int obs;     // OBS_STRIDE per static circle: xs ys r^2 r
    int f2;      // n2 penalty values
    int dyn;     // NDYN_MAX x 6 x dyn_stride per-stage ellipse data
    int dyn_stride;  // columns per (ellipse, field): 24 / 32 for the three- / two-point layouts, N rounded up to even for one point
    int req;     // hybrid kernel, team mode: the line-search request of this wave's instance -- u, r, d as 3 x 24 (v, w) pairs by stage
    int vec;     // 7 x P parked (v, w) pairs: L-BFGS old u / old r, previous gradient, y+, y, reference speed, grad at u_k
    int rho;     // m
    int S, Y;    // m slots x N lanes x (v, w)
    int nv;      // hybrid kernel, Gram-form L-BFGS: the four vectors of an iteration -- s | y | r | g -- as 4 x GRAM_NST (v, w) pairs by stage
    int gsy, gyy; // ... and the inner products it keeps, [slot][slot]: <s_a, y_b> (a older than b; zero otherwise), <y_a, y_b>
    int total;
};

struct KArgs {
    nmpc_problem pb;
    nmpc_opts op;
    LdsMap map;
    int B;
    int n_p, n_u, n1, n2;
    double inv_ts;
    const double *p;
    double *u;
    const double *y0;
    const double *c0;
    double *y_out;
    nmpc_status *st;
    unsigned int *queue;
    const int *order;          // queue position -> instance (longest-expected-first), or NULL = index order
    // migration of long-running instances to the SIMD's favoured wave slot (nmpc_solve_hyb.h), 0 = off
    int park_min;              // passes after which an instance on an unfavoured wave is parked at an outer-iteration boundary
    int park_depth;            // ... unless this many parked instances are already waiting for a favoured wave
    double *park;              // [B][park_stride]: parked solver state
    int *pool;                 // [B]: parked instance ids in arrival order (-1: not yet published)
    unsigned int *pool_ctr;    // per pool: head, tail, count, pad (nmpc_solve_hyb.h: POOL_CTRS); after the pools: instances alive that are known to be long
    int pool_cap;              // slots per pool ring (>= B)
    int sched_mode;            // 0: an instance stays on its wave (but for the slot migration); 1: step-aside scheduling at outer-iteration boundaries (nmpc_solve_hyb.h)
    int sched_long_cap;        // long instances alive beyond this many time-share the waves
    int sched_cold_cap;        // cold instances step aside once this many long instances are alive (a batch without long instances has nobody to make room for)
    int dbg;                   // experiments (NMPC_DEBUG_PRIO): static wave priorities + per-instance cycle counts
    int team_owners;           // hybrid kernel: waves per workgroup that take instances from the queue (1..4); the others only help
    int team_help;             // 0: nobody asks for help (experiments, NMPC_TEAM_HELP=0: the single-wave baseline)
    double cull_radius;        // eval_psi's CULL path: circles whose edge is farther than this from the start position are left out of the scan
    // eval kernel only
    const double *ev_c;
    const double *ev_y;
    double *ev_psi, *ev_grad, *ev_F1, *ev_F2;
};

enum { SC_X0 = 0, SC_Y0, SC_TH0, SC_VINIT, SC_WINIT, SC_XF, SC_YF, SC_THF,
       SC_Q, SC_QV, SC_QTH, SC_RV, SC_RW, SC_QN, SC_QTHN, SC_QCTE, SC_PA, SC_PW };

// LDS pointers carry their address space: no generic-pointer casts, always ds_* instructions
typedef __attribute__((address_space(3))) double lds_double;
typedef double dbl2 __attribute__((ext_vector_type(2)));     // (v, w) pair, 16-byte aligned
typedef __attribute__((address_space(3))) dbl2 lds_double2;

#ifdef NMPC_NO_SCHED_BARRIER
#define NMPC_SCHED_BARRIER() do { } while (0)
#else
#define NMPC_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
#define NMPC_WAVE_SYNC()                                           \
    do {                                                           \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     \
        __builtin_amdgcn_wave_barrier();                           \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     \
    } while (0)

// per-stage data of the dynamic ellipses: six values per (ellipse, stage), kept in the LDS slice as
// [ellipse][field][stage] so the stage's lane reads its column conflict-free
enum { DY_EX = 0, DY_EY, DY_CA, DY_SA, DY_IRX2, DY_IRY2, DY_FIELDS };
struct DynStage {
    const lds_double *col;     // this lane's column
    int stride;                // lanes per group (P)
    __device__ __forceinline__ double get(int k, int f) const { return col[(k * DY_FIELDS + f) * stride]; }
};

// Problem shape known at compile time (0 / -1: taken from the arguments at run time).  The reference
// generates one solver per configuration (mpc_generator.py:173-193); ShapeDefault is the shape of
// configs/default.yaml (N_hor 20, Nobs 10, Ndynobs 3), for which loops unroll and LDS offsets fold.
struct ShapeAny { static constexpr int N = 0, NOBS = -1, NDYN = -1; };
struct ShapeDefault { static constexpr int N = 20, NOBS = 10, NDYN = 3; };
struct ShapeNobs50 { static constexpr int N = 20, NOBS = 50, NDYN = 3; };     // BASELINE config 3
struct ShapeN40 { static constexpr int N = 40, NOBS = 10, NDYN = 3; };        // BASELINE config 2
template <class SH> __device__ __forceinline__ int shape_N(const KArgs &a) { if constexpr (SH::N > 0) return SH::N; else return a.pb.N; }
template <class SH> __device__ __forceinline__ int shape_nobs(const KArgs &a) { if constexpr (SH::NOBS >= 0) return SH::NOBS; else return a.pb.nobs; }
template <class SH> __device__ __forceinline__ int shape_ndyn(const KArgs &a) { if constexpr (SH::NDYN >= 0) return SH::NDYN; else return a.pb.ndyn; }

// The LDS slice layout (offsets in doubles) as a function of the problem shape and the lane layout P (20: three query points
// per wave, 32: two, 64: one).  constexpr: the shape-specialised kernels fold every offset into the ds_* instructions'
// immediate fields instead of carrying a dozen kernel arguments in (spilled) SGPRs; the host computes the same map for the
// run-time-shape kernels and for sizing the launch.  The L-BFGS ring is sized for MAXMEM slots whatever opts.lbfgs_memory is.
__host__ __device__ constexpr LdsMap lds_layout(int N, int nobs, int ndyn, int P)
{
    LdsMap mp{};
    int o = 0;
    mp.sc = o;  o += 20;
    mp.cw = o;  o += CW_NCOEF;
    mp.par = o; o += 24;
    mp.seg = o; o += SEG_STRIDE * (N + 5);
    mp.obs = o; o += OBS_STRIDE * (nobs + 4);
    const int points = P == 64 ? 1 : 3;               // F2 arrays: one per query point of a pass (eval kernel: per group slice)
    // (three-point layout: only the cost-layer kernel writes F2, and it has no parked vectors -- the array shares their place)
    mp.f2 = o;  o += P == 20 ? 0 : points * (nobs + ndyn + 1);
    mp.rho = o; o += MAXMEM;
    const int cols = P == 20 ? 24 : P;                // >= lay_cols (hybrid kernel: state lanes 24..31 share column 23 -- all zeros)
    // one point per wave keeps its solver vectors in registers and needs ellipse columns for the real stages only: without
    // the 64-column tables a 40-stage slice is 21.6 KB instead of 32.9 KB -- 7 resident waves per CU instead of 4
    mp.dyn_stride = P == 64 ? ((N + 1) & ~1) : (P == 20 ? 24 : P);
    mp.dyn = o; o += NDYN_MAX * 6 * mp.dyn_stride;
    o = (o + 1) & ~1;
    mp.req = o; o += P == 20 ? TEAM_REQ_DOUBLES : 0;
    mp.vec = o; o += P == 64 ? 0 : 7 * 2 * cols;
    if (P == 20) { mp.f2 = mp.vec; if (7 * 2 * cols < points * (nobs + ndyn + 1)) o = mp.vec + points * (nobs + ndyn + 1); }
    o = (o + 1) & ~1;                                 // 16-byte alignment for the double2 arrays
    // hybrid kernel: GRAM_NST + 1 columns per slot whatever N is -- the Gram batch reads a slot as GRAM_NST pairs, the last column is
    // all zeros (lanes beyond the horizon read it); gsy | gyy | S | Y are contiguous (zeroed together when the buffer is reset)
    mp.gsy = o; o += P == 20 ? MAXMEM * GRAM_LD : 0;
    mp.gyy = o; o += P == 20 ? MAXMEM * GRAM_LD : 0;
    const int ring = P == 20 ? GRAM_NST + 1 : N;
    mp.S = o;   o += 2 * ring * MAXMEM;
    mp.Y = o;   o += 2 * ring * MAXMEM;
    mp.nv = o;  o += P == 20 ? 4 * 2 * GRAM_NST : 0;
    mp.total = (o + 1) & ~1;
    // team mode: a helper wave's slice holds one result area per (owner, task) from offset 0 -- twelve of them; short horizons make slices
    // smaller than that (N_hor <= 14), and an area past the slice would land in the next wave's tables
    if (P == 20 && mp.total < 3 * TEAM_WAVES * TEAM_AREA_DOUBLES) mp.total = 3 * TEAM_WAVES * TEAM_AREA_DOUBLES;
    return mp;
}
// the map a kernel instantiation works with: compile-time for a fixed shape, the launch argument otherwise
template <class SH, int P> __device__ __forceinline__ LdsMap the_map(const KArgs &a)
{
    if constexpr (SH::N > 0 && SH::NOBS >= 0 && SH::NDYN >= 0) return lds_layout(SH::N, SH::NOBS, SH::NDYN, P);
    else return a.map;
}

// ---------------------------------------------------------------------------------------------
// instance set-up: p -> LDS slice + per-lane registers     (reference mpc_generator.py:73-79,93-104,127-136)
// ---------------------------------------------------------------------------------------------
template <int P, class SH = ShapeAny>
__device__ __forceinline__ void prepare_instance(const KArgs &a, lds_double *L, const double *p, int t,
                                                 double &vref, DynStage &dyn)
{
    const int N = shape_N<SH>(a), nobs = shape_nobs<SH>(a), ndyn = shape_ndyn<SH>(a);
    const LdsMap mp = the_map<SH, P>(a);
    if (t < 8) L[mp.sc + t] = p[t];                      // state, last input, target (p[8:10] unused)
    if (t >= 8 && t < 18) L[mp.sc + t] = p[t + 2];       // ten weights p[10:20]
    if (t == 18 || t == 19) L[mp.sc + t] = p[t - 15];    // the last input once more, as a (v, w) pair: "the stage before stage 0" of the hybrid kernel's transport
    if (t < CW_NCOEF) L[mp.cw + t] = CW_COEF_DEV[t];
    NMPC_WAVE_SYNC();
    vref = t < N ? p[NZ + t] : 0.0;
    const double *ps = p + NZ + N;
    for (int k = t; k < ((nobs + 4) & ~3); k += P) {       // padded to a multiple of 4 with inert zero circles (slot `nobs` always is one)
        const bool real = k < nobs;
        const double r = real ? ps[3 * k + 2] : 0.0;
        L[mp.obs + OBS_STRIDE * k] = real ? ps[3 * k] : 0.0;
        L[mp.obs + OBS_STRIDE * k + 1] = real ? ps[3 * k + 1] : 0.0;
        L[mp.obs + OBS_STRIDE * k + 2] = r * r;
        L[mp.obs + OBS_STRIDE * k + 3] = r > 0.0 ? r : -1e30;      // (obstacle certificate: an empty slot is infinitely far away)
    }
    const double *pd = ps + 3 * nobs;
    {
        lds_double *col = L + mp.dyn + t;
        dyn.col = col;
        // one point per wave (P = 64): only the N real stages have a column (the slice then fits 7 waves per CU, not 4)
        const int ds = P == 64 ? mp.dyn_stride : lay_cols<P>();
        dyn.stride = ds;
#pragma unroll
        for (int k = 0; k < NDYN_MAX; ++k) {
            double ex = 0.0, ey = 0.0, ca = 0.0, sa = 0.0, irx2 = 1.0, iry2 = 1.0;
            if (k < ndyn && t < N) {
                const double *e = pd + (k * N + t) * 5;
                ex = e[0];
                ey = e[1];
                irx2 = 1.0 / (e[2] * e[2]);
                iry2 = 1.0 / (e[3] * e[3]);
                sincos_cw_t(e[4], (const lds_double *)(L + mp.cw), sa, ca);
            }
            if (P != 64 || t < ds) {
                col[(k * DY_FIELDS + DY_EX) * ds] = ex;
                col[(k * DY_FIELDS + DY_EY) * ds] = ey;
                col[(k * DY_FIELDS + DY_CA) * ds] = ca;
                col[(k * DY_FIELDS + DY_SA) * ds] = sa;
                col[(k * DY_FIELDS + DY_IRX2) * ds] = irx2;
                col[(k * DY_FIELDS + DY_IRY2) * ds] = iry2;
            }
        }
    }
    const double *pr = pd + 5 * ndyn * N;
    const int nseg4 = (N - 1 + 3) & ~3;                    // the CTE loop runs 4 segments per trip; the padding
    if (t < nseg4) {                                       // repeats the last segment (cannot change a strict min)
        const int i = t < N - 1 ? t : N - 2;
        const double ax = pr[3 * i], ay = pr[3 * i + 1];
        const double bx = pr[3 * i + 3], by = pr[3 * i + 4];
        const double dx = bx - ax, dy = by - ay;
        lds_double *sg = L + mp.seg + SEG_STRIDE * t;
        sg[0] = ax;
        sg[1] = ay;
        sg[2] = dx;
        sg[3] = dy;
        sg[4] = 1.0 / (fma(dx, dx, dy * dy) + 1e-16);
    }
    NMPC_WAVE_SYNC();
}

// Windowed cross-track search (eval_psi / eval_psi2, WIN > 0).  What a lane remembers from its last FULL scan of the reference segments:
// the centre of its window, where the stage was then, and the squared distance from there to the nearest segment OUTSIDE the window
// (0 = nothing known: the next evaluation scans everything).
struct WinState {
    int ctr;
    double xr, yr, mo2;
};
// -DNMPC_TL (scripts/timeline.py): s_memtime of fifteen events of a helped iteration -- owner 0..10, the helper of its first task 11..14 --
// for 64 consecutive iterations of the instance that runs them (one instance solved alone)
#ifdef NMPC_TL
__device__ long long nmpc_tl[64 * 16];
#ifdef NMPC_MARKS      // (with -DNMPC_MARKS: the events as markers in the ISA dump instead)
#define NMPC_TL_EV(it, ev) do { (void)(it); __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK TL_" #ev); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NMPC_TL_EV(it, ev) do { if ((it) >= 200 && (it) < 264 && lane == 0) nmpc_tl[((it) - 200) * 16 + (ev)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#define NMPC_TL_KEEP(x) do { double keep_ = (x); asm volatile("" : "+v"(keep_)); } while (0)
#else
#define NMPC_TL_EV(it, ev) do { } while (0)
#define NMPC_TL_KEEP(x) do { } while (0)
#endif
#ifdef NMPC_BBCOUNT
// scripts/bbcount.py: one counter per basic block of ONE solve kernel.  The increments are not in this source: bbcount.py rewrites the
// compiler's assembly (four instructions at the head of every block, registers the kernel does not use) and links the result against this array.
__device__ __attribute__((used)) unsigned int nmpc_bbcnt[4096];
#endif
#ifdef NMPC_WIN_STATS
__device__ unsigned long long nmpc_win_stats[4];       // evaluations that tried the window | of which fell back to the full scan | that tried the obstacle certificate | of which scanned
#endif
// Obstacle certificate (eval_psi, oc != nullptr).  The activity scan of an evaluation only decides WHICH circles / ellipses have a stage of
// the wave inside them (the touched ones are then summed exactly); from one evaluation to the next that set rarely changes.  So a lane
// remembers where its stage was at the wave's last scan and how far that was -- at least -- from every obstacle the scan found
// untouched (distance to the circle's edge; for an ellipse to the disc of its larger half axis around its centre; for the culled scan
// also to the culling radius), and the wave remembers the scan's verdict.  While every stage has moved by less than its clearance no
// untouched obstacle can have been entered: the old verdict is a superset of the true one, and a superfluous member contributes exactly
// zero (its sum is +0.0, no lane is inside it) -- the scan is skipped and the result is bit for bit the scanning evaluation's.  The
// clearances come from v_sqrt_f64 / v_rsq_f64 (approximate) with 1 % + 1e-6 taken off: they only decide whether the scan runs.
struct ObsCert {
    double xo, yo, m2;             // this lane: the stage's position at the wave's last scan, squared clearance there (0: scan next time).  (A reference
                                   // point shared with the cross-track window was measured: four registers less, but either certificate's failure then
                                   // runs both scans -- 10 % of the evaluations instead of 1 %, headline + 6 %.)
    // the wave: circles and ellipses the last scan found touched.  Kept in VECTOR registers (every lane the same value; read back with
    // v_readfirstlane): as scalar-register values in the select chains of the caller they crash ROCm 7.2's greedy register allocator
    // (VirtRegAuxInfo::isRematerializable, iterative-ilp, the Nobs = 50 instantiation)
    int act_lo, act_hi, act_dyn;
};
__device__ __forceinline__ int opaque_i(int x) { asm("" : "+v"(x)); return x; }
// Is the windowed minimum `best` (squared) the global one?  With a2 = |p - p_ref|^2 and mo2 = the squared clearance of the window at
// p_ref, every segment outside the window is at least sqrt(mo2) - |p - p_ref| away from p (distances are 1-Lipschitz), so it is if
// sqrt(best) + |p - p_ref| < sqrt(mo2)  <=>  t = mo2 - a2 - best > 0 and t^2 > 4 a2 best.  The margins (1e-5 relative on squared
// distances) dwarf the rounding of the distance formula (<= 2e-10 relative wherever it matters; mo2 <= 1e-8 is stored as 0).
__device__ __forceinline__ bool window_is_global(double a2, double best, double mo2)
{
    const double t = mo2 - (a2 + best);
    return t > 1e-5 * mo2 && t * t > 4.0001 * (a2 * best);
}

// the circles of an instance whose edge lies within `radius` of the start position (bit k = circle k); padding slots (r = 0) never are
__device__ __forceinline__ unsigned long long circle_near_mask(const double *p, int N, int nobs, int lane, double radius)
{
    const double *ps = p + NZ + N;
    bool keep = false;
    if (lane < nobs) {
        const double dx = ps[3 * lane] - p[0], dy = ps[3 * lane + 1] - p[1], r = ps[3 * lane + 2], lim = radius + r;
        keep = r > 0.0 && fma(dx, dx, dy * dy) <= lim * lim;
    }
    return __ballot(keep);
}

// ---------------------------------------------------------------------------------------------
// psi(z; c, y), grad psi, F1 (av, aw), sum_k F2_k^2 (pen); WRITE_F2: F2_k also left in the LDS slice
// ---------------------------------------------------------------------------------------------
#if defined(NMPC_PROF2) && NMPC_PROF2 == 2      // scripts/sections.py: cycles of the evaluation by section, accumulated in registers of the caller
#define NMPC_EVTICK(i) do { if (nmpc_pe) { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); nmpc_pe[i] += t_ - nmpc_pe[7]; nmpc_pe[7] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#elif defined(NMPC_MARKS)       // scripts/isa_stats.py: section markers in the ISA dump
#define NMPC_EVTICK(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " #i); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NMPC_EVTICK(i) do { } while (0)
#endif
// CULL: `near` is the set of static circles that can be touched at all while every stage stays within KArgs.cull_radius of the start
// position (circle_near_mask below); the activity scan visits those only, and falls back to all of them for an evaluation in
// which some stage is farther away -- so the result is exactly that of the full scan.
// The handful of launch-uniform scalars an evaluation reads, as values of their own.  Read from the argument block (a.pb.*) they belong to a
// sixteen-dword scalar load whose registers the allocator spills and reloads AS ONE (sixteen v_readlane per use of one bound); a kernel that
// hands them over in this struct -- each passed through scalar_own() once -- pays two.
struct EvK { double ts, inv_ts, amin, amax, awmax; };
__device__ __forceinline__ double scalar_own(double x)
{
    // through a vector register and back (v_readfirstlane): a definition of its own that the coalescer cannot fold back into the loaded tuple
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return __hiloint2double(__builtin_amdgcn_readfirstlane(hi), __builtin_amdgcn_readfirstlane(lo));
}
__device__ __forceinline__ int scalar_own(int x)
{
    asm volatile("" : "+v"(x));
    return __builtin_amdgcn_readfirstlane(x);
}
// What the hybrid kernel's owner path hands over because its query points travel through LDS (nmpc_solve_hyb.h, "transport"): the control
// pair of the stage before (the last input for stage 0) -- read from the transport area one slot down instead of fetched from the neighbour
// lane --, this lane's slot of the area for handing (qa, qw) to the stage before, and the slot of the stage after (a zero pad behind the
// last stage).  Two pointers that the compiler cannot tell apart: the write stays in front of the read, and LDS serves a wave in order.
// Lanes 60..63 of such an evaluation hold zeros in zv, zw (Z60: group_prefix_ex_z60).
struct EvX {
    double vprev, wprev;
    lds_double2 *mine;
    const lds_double2 *next;
};
template <int P, class SH = ShapeAny, bool WRITE_F2 = false, bool CULL = false, int WIN = 0, bool Z60 = false>
__device__ __forceinline__ void eval_psi(const KArgs &a, lds_double *L, int f2off, int lane, int t, double zv, double zw,
                                         double c, double cbar_inv, double yv, double yw, double vref, const DynStage &dyn,
                                         bool want_grad, double &psi, double &pen_out, double &gv,
                                         double &gw, double &av_out, double &aw_out, unsigned long long near = ~0ull, WinState *ws = nullptr,
                                         ObsCert *oc = nullptr, long long *nmpc_pe = nullptr, const EvK *ek = nullptr, const EvX *evx = nullptr)
{
    static_assert(!Z60 || P == 20, "zero pads in lanes 60..63: the tri layout only");
    const int N = shape_N<SH>(a), nobs = shape_nobs<SH>(a), ndyn = shape_ndyn<SH>(a);
    const LdsMap mp = the_map<SH, P>(a);
    const double ts = ek ? ek->ts : a.pb.ts, inv_ts = ek ? ek->inv_ts : a.inv_ts;
    const double k_amin = ek ? ek->amin : a.pb.amin, k_amax = ek ? ek->amax : a.pb.amax, k_awmax = ek ? ek->awmax : a.pb.awmax;
    (void)nmpc_pe;
    // every stage lane of the tri layout is inside a 20-stage horizon; lanes 60..63 then hold
    // don't-care values that no cross-lane operation lets into the other lanes (nmpc_device.h)
    constexpr bool FULL = P == 20 && SH::N == 20;
    const bool in_r = t < N;                    // a real stage
    const bool in = FULL ? true : in_r;         // arithmetic masks: compile-time true when FULL
    const lds_double *sc = L + mp.sc;
    const double x0 = sc[SC_X0], y0 = sc[SC_Y0], th0 = sc[SC_TH0];
    const double xf = sc[SC_XF], yf = sc[SC_YF], thf = sc[SC_THF];

    // rollout (:88-90) as three prefix sums
    // (the pre-update state of a stage is the post-update state of the stage before: the same fma on the prefix sum of the stage before,
    // which the scan hands over with its own carry exchange -- group_prefix_ex)
    double ew_, ex_, ey_;
    auto prefix_ex = [lane](double v, double &excl) {
        if constexpr (Z60) return group_prefix_ex_z60(v, lane, excl);
        else return group_prefix_ex<P>(v, lane, excl);
    };
    const double thn = fma(ts, prefix_ex(zw, ew_), th0);
    const double th = t == 0 ? th0 : fma(ts, ew_, th0);
    double sn, cs;
    sincos_cw_t(th, (const lds_double *)(L + mp.cw), sn, cs);
    const double xn = fma(ts, prefix_ex(zv * cs, ex_), x0);
    const double yn = fma(ts, prefix_ex(zv * sn, ey_), y0);
    const double xp = t == 0 ? x0 : fma(ts, ex_, x0);
    const double yp = t == 0 ? y0 : fma(ts, ey_, y0);

    const double half_c = 0.5 * c;
    NMPC_EVTICK(0);     // rollout

    double acc = (sc[SC_RV] * zv) * zv;                                           // (:84)
    acc = fma(sc[SC_RW] * zw, zw, acc);
    const double dv = zv - vref;                                                  // (:85)
    acc = fma(sc[SC_QV] * dv, dv, acc);
    {
        const double ddx = xp - xf, ddy = yp - yf, dth = th - thf;                // (:86, 59-64)
        acc = fma(sc[SC_Q], fma(ddx, ddx, ddy * ddy), acc);
        acc = fma(sc[SC_QTH] * dth, dth, acc);
    }
    // cross-track error: min over the N-1 reference segments (:121-144)
    double best = __builtin_inf();
    int bi = 0;
    bool full_scan = true;
    int i0c = 0;                        // first segment of the window the full scan measures the clearance of
    if constexpr (WIN > 0) {
        // WINDOWED SEARCH (exact).  From one evaluation to the next a stage's nearest segment rarely moves, so only the 2 WIN + 1
        // segments around the lane's window centre are measured -- per-lane LDS gathers instead of broadcasts -- and the result is
        // accepted if it is PROVABLY the full scan's (window_is_global above).  If any stage of the wave fails the test, or holds no
        // clearance yet, the full scan below runs instead and renews every lane's clearance; either way `best`, `bi` are the full scan's.
        const int nseg = N - 1;
        if (nseg >= 2 * WIN + 1) {
            int cc = ws->ctr;
            cc = cc < 0 ? 0 : (cc > nseg - 1 ? nseg - 1 : cc);
            i0c = cc - WIN;
            i0c = i0c < 0 ? 0 : (i0c > nseg - (2 * WIN + 1) ? nseg - (2 * WIN + 1) : i0c);
            if (!__any(in_r & !(ws->mo2 > 0.0))) {
                const lds_double *sg = L + mp.seg + SEG_STRIDE * i0c;
                double wv[2 * WIN + 1][5];
#pragma unroll
                for (int j = 0; j <= 2 * WIN; ++j)
#pragma unroll
                    for (int f = 0; f < 5; ++f) wv[j][f] = sg[j * SEG_STRIDE + f];
#pragma unroll
                for (int j = 0; j <= 2 * WIN; ++j) {
                    const double px = xn - wv[j][0], py = yn - wv[j][1];
                    const double dot = fma(px, wv[j][2], py * wv[j][3]);
                    const double that = dot * wv[j][4];
                    const double tst = fmin(fmax(that, 0.0), 1.0);
                    const double ex = fma(tst, wv[j][2], -px), ey = fma(tst, wv[j][3], -py);
                    const double d2 = fma(ex, ex, ey * ey);
                    bi = d2 < best ? i0c + j : bi;
                    best = fmin(best, d2);
                }
                const double ax = xn - ws->xr, ay = yn - ws->yr;
                const bool sure = window_is_global(fma(ax, ax, ay * ay), best, ws->mo2);
                full_scan = __any(in_r & !sure);
#ifdef NMPC_WIN_STATS
                if (lane == 0) { atomicAdd(&nmpc_win_stats[0], 1ull); if (full_scan) atomicAdd(&nmpc_win_stats[1], 1ull); }
#endif
                if (full_scan) {
                    // the full scan measures the clearance of the window around what the old window found nearest
                    i0c = bi - WIN;
                    i0c = i0c < 0 ? 0 : (i0c > nseg - (2 * WIN + 1) ? nseg - (2 * WIN + 1) : i0c);
                    best = __builtin_inf(); bi = 0;
                }
            }
        }
    }
    if (full_scan) {
        const lds_double *sg = L + mp.seg;
        const int nseg4 = (N - 1 + 3) & ~3;
        // software pipeline: the ten LDS reads of the NEXT pair of segments are issued before the current
        // pair is reduced (the scheduling barriers keep the compiler from sinking them to their uses)
        double cur[2][5], nxt[2][5];
        double mout = __builtin_inf();                      // (WIN) nearest segment outside the window [i0c, i0c + 2 WIN]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int f = 0; f < 5; ++f) cur[j][f] = sg[j * SEG_STRIDE + f];
#pragma unroll SH::N > 0 ? (SH::N <= 20 ? 32 : 2) : 1
        for (int i = 0; i < nseg4; i += 2) {
            sg += 2 * SEG_STRIDE;                           // table is padded: reading one pair past the end is safe
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < 5; ++f) nxt[j][f] = sg[j * SEG_STRIDE + f];
            NMPC_SCHED_BARRIER();
            double d2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double px = xn - cur[j][0], py = yn - cur[j][1];
                const double dot = fma(px, cur[j][2], py * cur[j][3]);
                const double that = dot * cur[j][4];
                const double tst = fmin(fmax(that, 0.0), 1.0);
                const double ex = fma(tst, cur[j][2], -px), ey = fma(tst, cur[j][3], -py);
                d2[j] = fma(ex, ex, ey * ey);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {                   // strict <: the first minimum keeps its index
                bi = d2[j] < best ? i + j : bi;
                best = fmin(best, d2[j]);
                if constexpr (WIN > 0) {                    // (a padding entry repeats the last segment)
                    const int ie = i + j < N - 1 ? i + j : N - 2;
                    mout = (unsigned)(ie - i0c) <= 2u * WIN ? mout : fmin(mout, d2[j]);
                }
            }
            NMPC_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < 5; ++f) cur[j][f] = nxt[j][f];
        }
        if constexpr (WIN > 0) {
            // this lane's certificate for the evaluations to come: if the nearest segment lies in the window that was measured, the
            // window stays and its clearance is known; if not, the window moves there and the next evaluation measures it
            const bool inw = (unsigned)(bi - i0c) <= 2u * WIN;
            ws->ctr = inw ? i0c + WIN : bi;
            ws->xr = xn; ws->yr = yn;
            ws->mo2 = inw && mout > 1e-8 ? mout : 0.0;
        }
    }
    NMPC_EVTICK(1);     // stage cost + CTE loop
    acc = fma(sc[SC_QCTE], best, acc);                                            // (:144)
    // accelerations (:160-161), their cost (:170-171) and the ALM term
    const double vprev = evx ? evx->vprev : from_prev<P>(zv, lane, sc[SC_VINIT]);
    const double wprev = evx ? evx->wprev : from_prev<P>(zw, lane, sc[SC_WINIT]);
    double av = (zv - vprev) * inv_ts, aw = (zw - wprev) * inv_ts;
    acc = fma(sc[SC_PA] * av, av, acc);
    acc = fma(sc[SC_PW] * aw, aw, acc);
    const double tv = fma(yv, cbar_inv, av), tw = fma(yw, cbar_inv, aw);
    double sv = tv - clampd(tv, k_amin, k_amax);
    double sw = tw - clampd(tw, -k_awmax, k_awmax);
    acc = fma(half_c, fma(sv, sv, sw * sw), acc);
    if (t == N - 1) {                                                             // terminal (:148)
        const double tx = xn - xf, ty = yn - yf, tth = thn - thf;
        acc = fma(sc[SC_QN], fma(tx, tx, ty * ty), acc);
        acc = fma(sc[SC_QTHN] * tth, tth, acc);
    }
    if (!in) { acc = 0.0; av = aw = sv = sw = 0.0; }
    av_out = av;
    aw_out = aw;
    const double fsum = group_sum<P>(acc, lane);
    NMPC_EVTICK(2);     // accelerations, ALM term, cost sum

    // obstacle penalties on the post-update state (:106-119).  F2_k = sum_t max(0, h_kt); an obstacle that no stage
    // of any query point in this wave is inside of contributes exactly 0 to psi and to grad psi and is skipped
    // (wave-uniform branch).  The adjoint terms of a touched obstacle, c F2_k dh_kt/d(x, y), are added right where
    // its F2_k has just been summed -- same operations in the same order as a separate sweep would do them (cross-
    // track term first, circles in ascending order, then ellipses), without the round trip of F2 through LDS.
    double pen = 0.0;
    unsigned long long act = 0ull;      // wave-uniform: circles some stage is inside of
    unsigned act_dyn = 0u;              // wave-uniform: ellipses some stage is inside of
    bool scan = true;
    if (oc) {
        const double ox = xn - oc->xo, oy = yn - oc->yo;
        const bool sure = fma(ox, ox, oy * oy) < oc->m2;
        if (!__any(in_r & !sure)) {
            act = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(oc->act_hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(oc->act_lo);
            act_dyn = (unsigned)__builtin_amdgcn_readfirstlane(oc->act_dyn);
            scan = false;
        }
#ifdef NMPC_WIN_STATS
        if (lane == 0) { atomicAdd(&nmpc_win_stats[2], 1ull); if (scan) atomicAdd(&nmpc_win_stats[3], 1ull); }
#endif
    }
    if (scan) {
        double mg = __builtin_inf();        // (oc) this lane's clearance from the obstacles the scan finds untouched
        const lds_double *ob = L + mp.obs;
        const int nobs4 = (nobs + 3) & ~3;
        if constexpr (CULL) {
            // only the circles of `near` -- unless a stage of this evaluation has left the radius the set was made for
            const unsigned long long all = nobs >= 64 ? ~0ull : (1ull << nobs) - 1ull;
            unsigned long long todo = near & all;
            if (todo != all) {
                const double rx = xn - x0, ry = yn - y0;
                const double rg = 0.999 * a.cull_radius;
                const double ro2 = fma(rx, rx, ry * ry);
                if (__any(in_r & !(ro2 <= rg * rg))) todo = all;
                else if (oc) mg = rg - __builtin_amdgcn_sqrt(ro2);      // the set holds while the stage stays inside the radius
            }
            while (todo) {                                  // four circles per trip; slot `nobs` holds an inert zero circle
                int kk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kk[j] = todo ? __builtin_ctzll(todo) : nobs;
                    todo &= todo - (todo ? 1ull : 0ull);
                }
                double od[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const lds_double *oj = ob + OBS_STRIDE * kk[j];
                    od[4 * j] = oj[0]; od[4 * j + 1] = oj[1]; od[4 * j + 2] = oj[2]; od[4 * j + 3] = oc ? oj[3] : 0.0;
                }
                NMPC_SCHED_BARRIER();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double dx = xn - od[4 * j], dy = yn - od[4 * j + 1];
                    const double h = fma(-dy, dy, fma(-dx, dx, od[4 * j + 2]));       // (:112)
                    if (__any(in_r & (h > 0.0))) act |= 1ull << (kk[j] & 63);          // (the inert circle never is)
                    else if (oc) mg = fmin(mg, __builtin_amdgcn_sqrt(od[4 * j + 2] - h) - od[4 * j + 3]);
                }
            }
        } else {
#pragma unroll SH::NOBS >= 0 && SH::NOBS <= 16 ? 16 : 1
        for (int k = 0; k < nobs4; k += 4, ob += 4 * OBS_STRIDE) {      // activity scan: four circles per trip, one ballot each
            double od[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                od[4 * j] = ob[OBS_STRIDE * j]; od[4 * j + 1] = ob[OBS_STRIDE * j + 1]; od[4 * j + 2] = ob[OBS_STRIDE * j + 2];
                od[4 * j + 3] = oc ? ob[OBS_STRIDE * j + 3] : 0.0;
            }
            NMPC_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double dx = xn - od[4 * j], dy = yn - od[4 * j + 1];
                const double h = fma(-dy, dy, fma(-dx, dx, od[4 * j + 2]));       // (:112)
                if (__any(in_r & (h > 0.0))) act |= 1ull << (k + j);
                else if (oc) mg = fmin(mg, __builtin_amdgcn_sqrt(od[4 * j + 2] - h) - od[4 * j + 3]);
            }
        }
        }
        NMPC_EVTICK(5);     // static circle scan
        {
            double dv_[NDYN_MAX][DY_FIELDS];
#pragma unroll
            for (int k = 0; k < NDYN_MAX; ++k)
#pragma unroll
                for (int f = 0; f < DY_FIELDS; ++f) dv_[k][f] = k < ndyn ? dyn.get(k, f) : 0.0;
            NMPC_SCHED_BARRIER();
#pragma unroll
            for (int k = 0; k < NDYN_MAX; ++k) {
                if (k < ndyn) {
                    const double ca = dv_[k][DY_CA], sa = dv_[k][DY_SA];
                    const double dx = xn - dv_[k][DY_EX], dy = yn - dv_[k][DY_EY];
                    const double ea = fma(dx, ca, dy * sa);
                    const double eb = fma(dx, sa, -(dy * ca));
                    const double h = fma(-(eb * eb), dv_[k][DY_IRY2], fma(-(ea * ea), dv_[k][DY_IRX2], 1.0));   // (:118)
                    if (__any(in_r & (h > 0.0))) act_dyn |= 1u << k;
                    else if (oc)      // the ellipse lies inside the disc of its larger half axis
                        mg = fmin(mg, __builtin_amdgcn_sqrt(fma(dx, dx, dy * dy)) - __builtin_amdgcn_rsq(fmin(dv_[k][DY_IRX2], dv_[k][DY_IRY2])));
                }
            }
        }
        if (oc) {
            const double m = fma(0.99, mg, -1e-6);
            oc->xo = xn; oc->yo = yn;
            oc->m2 = m > 0.0 ? (m < 1e100 ? m * m : 1e200) : 0.0;
            oc->act_lo = opaque_i((int)(unsigned)act); oc->act_hi = opaque_i((int)(unsigned)(act >> 32)); oc->act_dyn = opaque_i((int)act_dyn);
        }
        NMPC_EVTICK(6);     // ellipse scan
    }
    // ---- adjoint, first term: the cross-track error through the arg-min segment of this stage ----
    double gx = 0.0, gy = 0.0;
    if (want_grad) {
        const lds_double *sg = L + mp.seg + SEG_STRIDE * bi;
        const double px = xn - sg[0], py = yn - sg[1];
        const double dot = fma(px, sg[2], py * sg[3]);
        const double that = dot * sg[4];
        const double tst = fmin(fmax(that, 0.0), 1.0);
        const double ex = fma(tst, sg[2], -px), ey = fma(tst, sg[3], -py);
        const double ed = fma(ex, sg[2], ey * sg[3]);
        const double m = (that > 0.0 && that < 1.0) ? ed * sg[4] : 0.0;
        const double two_q = 2.0 * sc[SC_QCTE];
        gx = two_q * fma(m, sg[2], -ex);
        gy = two_q * fma(m, sg[3], -ey);
    }
    // ---- touched obstacles: F2_k, its square into the penalty, its adjoint terms ----
    if ((act | act_dyn) != 0ull) {
        for (unsigned long long rem = act; rem;) {          // two touched circles per trip: their tree sums interleave
            const int k0 = __builtin_ctzll(rem);
            rem &= rem - 1;
            if (rem == 0ull) {
                // a single circle left (the usual case of an instance that grazes an obstacle): one sum, not a pair with a dummy twin
                const lds_double *o0 = L + mp.obs + OBS_STRIDE * k0;
                const double ax = o0[0], ay = o0[1], ar = o0[2];
                const double dx0 = xn - ax, dy0 = yn - ay;
                const double h0 = fma(-dy0, dy0, fma(-dx0, dx0, ar));
                const double f20 = group_sum<P>(in ? fmax(h0, 0.0) : 0.0, lane);
                if (WRITE_F2 && t == 0) L[f2off + k0] = f20;
                pen = fma(f20, f20, pen);
                if (want_grad) {
                    const double w0 = -2.0 * (c * f20);
                    if (h0 > 0.0) { gx = fma(w0, dx0, gx); gy = fma(w0, dy0, gy); }
                }
                break;
            }
            const int k1 = __builtin_ctzll(rem);
            rem &= rem - 1;
            const lds_double *o0 = L + mp.obs + OBS_STRIDE * k0, *o1 = L + mp.obs + OBS_STRIDE * k1;
            const double ax = o0[0], ay = o0[1], ar = o0[2], bx = o1[0], by = o1[1], br = o1[2];
            const double dx0 = xn - ax, dy0 = yn - ay, dx1 = xn - bx, dy1 = yn - by;
            const double h0 = fma(-dy0, dy0, fma(-dx0, dx0, ar)), h1 = fma(-dy1, dy1, fma(-dx1, dx1, br));
            const double f20 = group_sum<P>(in ? fmax(h0, 0.0) : 0.0, lane);
            const double f21 = group_sum<P>(in ? fmax(h1, 0.0) : 0.0, lane);
            if (WRITE_F2 && t == 0) { L[f2off + k0] = f20; L[f2off + k1] = f21; }
            pen = fma(f20, f20, pen);
            pen = fma(f21, f21, pen);
            if (want_grad) {
                const double w0 = -2.0 * (c * f20), w1 = -2.0 * (c * f21);
                if (h0 > 0.0) { gx = fma(w0, dx0, gx); gy = fma(w0, dy0, gy); }
                if (h1 > 0.0) { gx = fma(w1, dx1, gx); gy = fma(w1, dy1, gy); }
            }
        }
#pragma unroll
        for (int k = 0; k < NDYN_MAX; ++k) {
            if (act_dyn & (1u << k)) {
                const double ca = dyn.get(k, DY_CA), sa = dyn.get(k, DY_SA);
                const double irx2 = dyn.get(k, DY_IRX2), iry2 = dyn.get(k, DY_IRY2);
                const double dx = xn - dyn.get(k, DY_EX), dy = yn - dyn.get(k, DY_EY);
                const double ea = fma(dx, ca, dy * sa);
                const double eb = fma(dx, sa, -(dy * ca));
                const double h = fma(-(eb * eb), iry2, fma(-(ea * ea), irx2, 1.0));      // (:118)
                const double f2 = group_sum<P>(in ? fmax(h, 0.0) : 0.0, lane);
                if (WRITE_F2 && t == 0) L[f2off + nobs + k] = f2;
                pen = fma(f2, f2, pen);
                if (want_grad) {
                    const double wk = -2.0 * (c * f2);
                    if (h > 0.0) {
                        const double A = ea * irx2, Bq = eb * iry2;
                        const double hx = fma(A, ca, Bq * sa);
                        const double hy = fma(A, sa, -(Bq * ca));
                        gx = fma(wk, hx, gx);
                        gy = fma(wk, hy, gy);
                    }
                }
            }
        }
    }
    psi = fma(half_c, pen, fsum);
    pen_out = pen;
    NMPC_EVTICK(3);     // obstacles
    if (!want_grad) return;

    // ---- adjoint sweep, continued (what CasADi reverse AD generated for the reference) ----
    // the post-update state of stage t is the tracked state of stage t+1 (:86) or the terminal state (:148)
    const double wq = t < N - 1 ? sc[SC_Q] : sc[SC_QN];
    const double wth = t < N - 1 ? sc[SC_QTH] : sc[SC_QTHN];
    gx = fma(2.0 * wq, xn - xf, gx);
    gy = fma(2.0 * wq, yn - yf, gy);
    double gt = (2.0 * wth) * (thn - thf);
    double qa = fma(c, sv, (2.0 * sc[SC_PA]) * av);
    double qw = fma(c, sw, (2.0 * sc[SC_PW]) * aw);
    if (!in) { gx = gy = gt = qa = qw = 0.0; }
    const double Sx = group_suffix<P>(gx, lane);
    const double Sy = group_suffix<P>(gy, lane);
    const double e = fma(Sy, cs, -(Sx * sn));
    const double Dt = in ? (ts * zv) * e : 0.0;
    const double St = group_suffix<P>(in ? gt + from_next<P>(Dt, lane) : 0.0, lane);
    double qan, qwn;
    if (evx) { *evx->mine = dbl2{qa, qw}; const dbl2 n_ = *evx->next; qan = n_.x; qwn = n_.y; }
    else { qan = from_next<P>(qa, lane); qwn = from_next<P>(qw, lane); }
    const double dynv = fma(Sx, cs, Sy * sn);
    double g1 = fma(2.0 * sc[SC_RV], zv, (2.0 * sc[SC_QV]) * dv);
    g1 = fma(inv_ts, qa - qan, g1);
    g1 = fma(ts, dynv, g1);
    double g2 = (2.0 * sc[SC_RW]) * zw;
    g2 = fma(inv_ts, qw - qwn, g2);
    g2 = fma(ts, St, g2);
    gv = in ? g1 : 0.0;
    gw = in ? g2 : 0.0;
    NMPC_EVTICK(4);     // adjoint sweep
}

// dot product of two horizon vectors (lane t holds the (v_t, w_t) pair)
template <int P>
__device__ __forceinline__ double hdot(double av, double aw, double bv, double bw, int lane)
{
    return group_sum<P>(fma(av, bv, aw * bw), lane);
}

// ---------------------------------------------------------------------------------------------
// cost-layer kernel: one evaluation per instance (parity tests, F1/F2 mapping API)
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(64) void nmpc_eval_kernel(KArgs a)
{
    extern __shared__ double lds[];
    constexpr int K = 64 / P;
    const int lane = threadIdx.x, g = lay_group<P>(lane), t = lay_stage<P>(lane);
    lds_double *L = (lds_double *)lds + g * a.map.total;
    const int N = a.pb.N;
    const bool in = t < N;
    const int inst = blockIdx.x * K + g;
    const int b = inst < a.B ? inst : a.B - 1;          // surplus groups redo the last instance, write nothing
    double vref;
    DynStage dyn;
    prepare_instance<P>(a, L, a.p + (size_t)b * a.n_p, t, vref, dyn);
    const double *u = a.u + (size_t)b * a.n_u;
    const double zv = in ? u[2 * t] : 0.0, zw = in ? u[2 * t + 1] : 0.0;
    const double c = a.ev_c ? a.ev_c[b] : 0.0;
    const double yv = (a.ev_y && in) ? a.ev_y[(size_t)b * a.n1 + t] : 0.0;
    const double yw = (a.ev_y && in) ? a.ev_y[(size_t)b * a.n1 + N + t] : 0.0;
    for (int k = t; k < a.n2; k += P) L[a.map.f2 + k] = 0.0;
    NMPC_WAVE_SYNC();
    double psi, pen, gv, gw, av, aw;
    eval_psi<P, ShapeAny, true>(a, L, a.map.f2, lane, t, zv, zw, c, 1.0 / fmax(c, 1.0), yv, yw, vref, dyn, true, psi, pen, gv, gw, av, aw);
    NMPC_WAVE_SYNC();          // F2_k written by lane 0 of the group are read by all its lanes below
    if (inst >= a.B) return;
    if (t == 0 && a.ev_psi) a.ev_psi[b] = psi;
    if (in) {
        if (a.ev_grad) { a.ev_grad[(size_t)b * a.n_u + 2 * t] = gv; a.ev_grad[(size_t)b * a.n_u + 2 * t + 1] = gw; }
        if (a.ev_F1) { a.ev_F1[(size_t)b * a.n1 + t] = av; a.ev_F1[(size_t)b * a.n1 + N + t] = aw; }
    }
    if (a.ev_F2) for (int k = t; k < a.n2; k += P) a.ev_F2[(size_t)b * a.n2 + k] = L[a.map.f2 + k];
}

}  // namespace nmpc

namespace nmpc {
// doubles per parked instance: u, y, previous gradient (2N each) + 16 scalars
__host__ __device__ inline int park_stride(int N) { return 6 * N + 16; }
}
#include "nmpc_solve_common.h"
#include "nmpc_solve_hyb.h"
#include "nmpc_solve_hyb2.h"
#include "nmpc_loop.h"

// ---------------------------------------------------------------------------------------------
// launch-order heuristic.  Iteration counts are heavy-tailed and a batch ends when its slowest
// instance does, so instances that LOOK hard are handed out first (list scheduling, longest expected
// first).  "Looks hard" uses the inputs only: the reference samples of the horizon pass within
// SCHED_CLEARANCE of a circle / ellipse, or the reference bends by more than SCHED_BEND inside
// the horizon.  Only the order of processing changes; every instance's result is independent of it.
// ---------------------------------------------------------------------------------------------
namespace nmpc {
constexpr double SCHED_CLEARANCE = 0.6;    // m
constexpr double SCHED_GRAZE = 0.05;       // m: the reference itself touches an obstacle's edge -- its penalty will be active
constexpr double SCHED_BEND = 0.05;        // rad, summed |heading change| of the reference samples
constexpr double SCHED_SPEED_GAP = 1.0;    // m/s between the last applied and the first reference speed: the
                                           // acceleration bounds stay active for several stages (many outer iterations)
constexpr int SCHED_LEVELS = 13;           // hardness level = 4 x (grazes) + 4 x (grazes within the first half of the horizon) + other criteria met

__global__ void nmpc_classify_kernel(KArgs a, unsigned char *cls)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int N = a.pb.N, nobs = a.pb.nobs, ndyn = a.pb.ndyn;
    const double *p = a.p + (size_t)b * a.n_p;
    const double *ps = p + NZ + N, *pd = ps + 3 * nobs, *pr = pd + 5 * ndyn * N;
    bool hard = false, graze = false, early = false;
    double bend = 0.0;
    for (int t = 0; t < N; ++t) {
        const double rx = pr[3 * t], ry = pr[3 * t + 1];
        if (t > 0) {
            double d = pr[3 * t + 2] - pr[3 * t - 1];
            d = d - 6.283185307179586 * rint(d * 0.15915494309189535);
            bend += fabs(d);
        }
        for (int k = 0; k < nobs; ++k) {
            const double r = ps[3 * k + 2];
            if (r > 0.0) {
                const double dx = rx - ps[3 * k], dy = ry - ps[3 * k + 1], lim = r + SCHED_CLEARANCE, lim0 = r + SCHED_GRAZE;
                hard |= dx * dx + dy * dy < lim * lim;
                graze |= dx * dx + dy * dy < lim0 * lim0;
                early |= 2 * t < N && dx * dx + dy * dy < lim0 * lim0;      // the sooner the robot meets the obstacle, the longer the solve
            }
        }
        for (int k = 0; k < ndyn; ++k) {
            const double *e = pd + (k * N + t) * 5;
            const double dx = rx - e[0], dy = ry - e[1], lim = fmax(e[2], e[3]) + SCHED_CLEARANCE, lim0 = fmin(e[2], e[3]) + SCHED_GRAZE;
            hard |= dx * dx + dy * dy < lim * lim;
            graze |= dx * dx + dy * dy < lim0 * lim0;
        }
    }
    const bool gap = fabs(p[NZ] - p[3]) > SCHED_SPEED_GAP;
    // the horizon reaches the goal: the reference is padded with the end pose (degenerate segments, braking profile)
    const bool goal = pr[3 * (N - 1)] == pr[3 * (N - 2)] && pr[3 * (N - 1) + 1] == pr[3 * (N - 2) + 1];
    cls[b] = (unsigned char)((graze ? 4 : 0) + (early ? 4 : 0) + (hard ? 1 : 0) + (bend > SCHED_BEND ? 1 : 0) + (gap ? 1 : 0) + (goal ? 1 : 0));
}

// The same levels from what a receding-horizon loop already knows: the evaluation passes each instance's solve took one step earlier
// (nmpc_status.reserved).  Consecutive solves of one robot are alike -- the previous count is a far better predictor of the next than anything the
// inputs show -- so the closed loop hands out its instances longest-last-time first (nmpc_loop_step); level = position of the count's top bit.
__global__ void nmpc_classify_prev_kernel(int B, const nmpc_status *prev, unsigned char *cls)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned n = prev[b].reserved;
    const int lvl = n < 32u ? 0 : (31 - __clz((int)n)) - 4;      // 32..63 passes -> 1, 64..127 -> 2, ...
    cls[b] = (unsigned char)(lvl > SCHED_LEVELS - 1 ? SCHED_LEVELS - 1 : lvl);
}

// stable partition of 0..B-1 by level (highest first); one block, deterministic
__global__ void nmpc_order_kernel(int B, const unsigned char *cls, int *order)
{
    __shared__ int cnt[SCHED_LEVELS][1024];
    const int t = threadIdx.x, nt = blockDim.x;
    const int chunk = (B + nt - 1) / nt;
    const int lo = t * chunk < B ? t * chunk : B, hi = lo + chunk < B ? lo + chunk : B;
    int c[SCHED_LEVELS];
#pragma unroll
    for (int k = 0; k < SCHED_LEVELS; ++k) c[k] = 0;
    for (int i = lo; i < hi; ++i) {
#pragma unroll
        for (int k = 0; k < SCHED_LEVELS; ++k) c[k] += cls[i] == k;
    }
#pragma unroll
    for (int k = 0; k < SCHED_LEVELS; ++k) cnt[k][t] = c[k];
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {            // inclusive scans, one per level
        int v[SCHED_LEVELS];
#pragma unroll
        for (int k = 0; k < SCHED_LEVELS; ++k) v[k] = t >= off ? cnt[k][t - off] : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCHED_LEVELS; ++k) cnt[k][t] += v[k];
        __syncthreads();
    }
    int pos[SCHED_LEVELS], base = 0;                    // write cursor of this chunk inside each level's segment
#pragma unroll
    for (int k = SCHED_LEVELS - 1; k >= 0; --k) {
        pos[k] = base + cnt[k][t] - c[k];
        base += cnt[k][nt - 1];
    }
    for (int i = lo; i < hi; ++i) {
        const int k = cls[i];
#pragma unroll
        for (int j = 0; j < SCHED_LEVELS; ++j) if (k == j) order[pos[j]++] = i;
    }
}
}  // namespace nmpc

// =================================================================================================
// C ABI (include/nmpc_solver.h)
// =================================================================================================
using nmpc::KArgs;
using nmpc::LdsMap;

struct nmpc_handle {
    nmpc_problem pb;
    nmpc_opts op;
    int device;
    int max_batch;
    bool alive;
    LdsMap map;
    int P;                 // 20: three query points per wave, one stage per lane (N_hor <= 20); 40: three points, two stages per lane (20 < N_hor <= 40)
    bool shape_default;    // (N, Nobs, Ndynobs) == ShapeDefault: the shape-specialised kernel runs
    bool shape_nobs50;     // ... == ShapeNobs50
    bool shape_n40;        // ... == ShapeN40
    int grid_cap;          // resident waves the launch is sized for
    double last_ms;        // kernel time of the last host-path batch
    size_t team_lds;       // hybrid kernel: dynamic LDS bytes of one workgroup (four slices + control block)
    unsigned int *d_queue;
    int park_min, park_depth;  // hybrid kernel: migrate instances after this many passes (0 = never) / pool depth limit
    int sched_mode;            // step-aside scheduling (NMPC_SCHED=0 switches it off); long instances time-share beyond sched_theta x resident waves (NMPC_SCHED_THETA)
    double sched_theta, sched_cold;
    int team_owners_forced;    // experiments (NMPC_TEAM_OWNERS): waves per workgroup that take instances, 0 = automatic
    int team_help;             // experiments (NMPC_TEAM_HELP=0): helpers never asked
    double cull_radius;        // eval_psi CULL (NMPC_CULL_RADIUS)
    double *d_park;            // parked solver states, allocated on first use
    int *d_pool;
    unsigned int *d_pool_ctr;
    bool loop_order_prev;      // nmpc_loop_step: launch order from the previous step's pass counts (experiments: NMPC_LOOP_ORDER_PREV=0 switches it off)
    int *d_order;              // launch order (hard-looking instances first)
    bool use_order;
    const nmpc_status *order_hint;   // set by nmpc_loop_step for the duration of its solve: the previous step's statuses (launch order by their pass counts)
    unsigned char *d_cls;
    // staging buffers of the host path
    double *d_p, *d_u, *d_y0, *d_c0, *d_yout, *d_psi, *d_grad, *d_F1, *d_F2;
    char *h_pin[2];            // pinned bounce buffers of the host entry points (pageable user memory <-> HBM at DMA speed)
    hipEvent_t pin_ev[2];
    bool staging_ready;        // every staging resource above exists
    // small batches through the host entry point (the reference's own call is B = 1, src/path_generator.py:385): ONE device arena
    // [p | c0 | y0 | u | y_out | status] per instance block, one pinned mirror, one copy in (p .. u), one copy out (u .. status), two events kept
    char *d_small, *h_small;
    hipEvent_t small_ev[2];
    bool small_ready;
    nmpc_status *d_st;
    std::string err;
};

extern "C" {

void nmpc_default_opts(nmpc_opts *o)
{
    o->tolerance = 1e-4;
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
    o->akkt_gradient = 1;      // step_top: OpEn caches the previous gradient at the top of step()
    o->ls_failure = 0;
    o->inner_status = 0;
    o->reserved = 0;
}

int nmpc_n_u(const nmpc_problem *pb) { return 2 * pb->N; }
int nmpc_n1(const nmpc_problem *pb) { return 2 * pb->N; }
int nmpc_n2(const nmpc_problem *pb) { return pb->nobs + pb->ndyn; }
int nmpc_n_p(const nmpc_problem *pb) { return nmpc::NZ + pb->N + 3 * pb->nobs + 5 * pb->ndyn * pb->N + 3 * pb->N; }
int nmpc_abi_version(void) { return NMPC_ABI_VERSION; }
int nmpc_experiments_build(void)
{
#ifdef NMPC_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

static int fail(nmpc_handle *h, int code, const char *what, hipError_t e = hipSuccess)
{
    if (h) {
        h->err = what;
        if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); }
    }
    return code;
}

#define HIP_TRY(h, call)                                                       \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return fail((h), NMPC_ERR_HIP, #call, e_);       \
    } while (0)

static LdsMap make_map(const nmpc_problem &pb, int P) { return nmpc::lds_layout(pb.N, pb.nobs, pb.ndyn, P); }

int nmpc_new(const nmpc_problem *pb, const nmpc_opts *opts, int device_id, int max_batch, nmpc_handle **out)
{
    if (!pb || !out || max_batch < 1) return NMPC_ERR_BAD_ARG;
    if (pb->N < 2 || pb->N > NMPC_MAX_HORIZON || pb->nobs < 0 || pb->nobs > 64 || pb->ndyn < 0 || pb->ndyn > nmpc::NDYN_MAX ||
        !(pb->ts > 0.0))
        return NMPC_ERR_BAD_PROBLEM;
    nmpc_opts op;
    if (opts) op = *opts; else nmpc_default_opts(&op);
    if (op.lbfgs_memory < 1 || op.lbfgs_memory > nmpc::MAXMEM || op.max_inner < 1 || op.max_outer < 1 ||
        op.max_total_inner < 0 || op.akkt_gradient < 0 || op.akkt_gradient > 2 || op.ls_failure < 0 || op.ls_failure > 1 ||
        op.inner_status < 0 || op.inner_status > 1)
        return NMPC_ERR_BAD_OPTS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
        return NMPC_ERR_NO_DEVICE;
    nmpc_handle *h = new nmpc_handle();
    h->pb = *pb; h->op = op; h->device = device_id; h->max_batch = max_batch; h->alive = true; h->last_ms = 0.0;
    h->P = pb->N <= 20 ? 20 : 40;      // one stage per lane (nmpc_solve_hyb.h) / two stages per lane (nmpc_solve_hyb2.h); longer horizons are not served
    h->shape_default = pb->N == nmpc::ShapeDefault::N && pb->nobs == nmpc::ShapeDefault::NOBS &&
                       pb->ndyn == nmpc::ShapeDefault::NDYN;
    h->shape_nobs50 = pb->N == nmpc::ShapeNobs50::N && pb->nobs == nmpc::ShapeNobs50::NOBS &&
                      pb->ndyn == nmpc::ShapeNobs50::NDYN;
    h->shape_n40 = pb->N == nmpc::ShapeN40::N && pb->nobs == nmpc::ShapeN40::NOBS && pb->ndyn == nmpc::ShapeN40::NDYN;
#ifdef NMPC_EXPERIMENTS      // (the experiments build, csrc/variants/libnmpc_experiments.so: tests and scripts only -- the shipped library reads no environment)
    if (const char *env = getenv("NMPC_SHAPE")) {              // force the run-time-shape kernel
        if (!strcmp(env, "any")) h->shape_default = h->shape_nobs50 = h->shape_n40 = false;
    }
#endif
    h->map = make_map(*pb, h->P == 40 ? 64 : h->P);      // (P = 40: the kernels compute their own map, nmpc_solve_hyb2.h)
    h->d_queue = nullptr;
    h->d_park = nullptr; h->d_pool = nullptr; h->d_pool_ctr = nullptr;
    h->park_min = 500; h->park_depth = 8;
    h->loop_order_prev = true;
    // long instances time-share beyond this fraction of the resident waves: the favoured half of them for the one-stage kernel (two waves per SIMD),
    // 0.8 for the two-stage kernel (one wave per SIMD); measured flat between 0.4 and 0.7 / 0.5 and 1.0 (profiles/r04/sched_sweep*.txt)
    h->sched_mode = 1; h->sched_theta = h->P == 20 ? 0.5 : 0.8;
    h->sched_cold = 0.4;
    h->team_owners_forced = 0;
    h->team_help = 1;
    // culling radius: what the input bounds let the robot travel in a horizon, plus a margin (any value is exact: an evaluation
    // with a stage beyond it scans every circle); NMPC_CULL_RADIUS overrides it (tests use 0.5 m: the fall-back runs all the time)
    h->cull_radius = 1.1 * pb->N * pb->ts * fmax(fabs(pb->vmin), fabs(pb->vmax));
    h->use_order = true;
    h->order_hint = nullptr;
#ifdef NMPC_EXPERIMENTS      // knobs of the experiments build: tests use them to check that every setting gives the same bits, scripts to measure
    if (const char *env = getenv("NMPC_PARK_MIN")) h->park_min = atoi(env);       // 0 switches the slot migration off
    if (const char *env = getenv("NMPC_PARK_DEPTH")) h->park_depth = atoi(env);
    if (const char *env = getenv("NMPC_LOOP_ORDER_PREV")) h->loop_order_prev = atoi(env) != 0;
    if (const char *env = getenv("NMPC_SCHED")) h->sched_mode = atoi(env);
    if (const char *env = getenv("NMPC_SCHED_THETA")) { const double v = atof(env); if (v > 0.0) h->sched_theta = v; }
    if (const char *env = getenv("NMPC_SCHED_COLD")) { const double v = atof(env); if (v > 0.0) h->sched_cold = v; }
    if (const char *env = getenv("NMPC_CULL_RADIUS")) { const double v = atof(env); if (v > 0.0) h->cull_radius = v; }
    if (const char *env = getenv("NMPC_TEAM_HELP")) h->team_help = atoi(env) != 0;
    if (const char *env = getenv("NMPC_ORDER")) h->use_order = atoi(env) != 0;      // 0 = instances in index order
    if (const char *env = getenv("NMPC_TEAM_OWNERS")) { const int v = atoi(env); if (v >= 1 && v <= nmpc::TEAM_WAVES) h->team_owners_forced = v; }
#endif
    h->d_order = nullptr;
    h->d_cls = nullptr;
    h->d_p = h->d_u = h->d_y0 = h->d_c0 = h->d_yout = h->d_psi = h->d_grad = h->d_F1 = h->d_F2 = nullptr;
    h->h_pin[0] = h->h_pin[1] = nullptr; h->pin_ev[0] = h->pin_ev[1] = nullptr; h->staging_ready = false;
    h->d_small = h->h_small = nullptr; h->small_ev[0] = h->small_ev[1] = nullptr; h->small_ready = false;
    h->d_st = nullptr;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_queue, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_order, sizeof(int) * (size_t)max_batch);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_cls, (size_t)max_batch);
    if (e != hipSuccess) { nmpc_free(h); return NMPC_ERR_HIP; }
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, device_id);
    const size_t lds_bytes = h->P == 40 ? (size_t)nmpc::lds_layout2(pb->N, pb->nobs, pb->ndyn).total * sizeof(double) * 3
                                        : (size_t)h->map.total * sizeof(double) * (64 / h->P);   // eval kernel: one slice per group
    if (lds_bytes > 160 * 1024) { nmpc_free(h); return NMPC_ERR_BAD_PROBLEM; }
    // the solve kernels use one LDS slice per wave; resident waves per CU are bounded by LDS and by
    // the register budget (2 waves per SIMD).  The hybrid kernel runs workgroups of four waves (teams).
    int per_cu;
    if (h->P == 20) {
        const size_t wg_bytes = nmpc::TEAM_WAVES * (size_t)h->map.total * sizeof(double) + nmpc::TEAM_CTL_INTS * sizeof(int);
        int wgs = (int)((160 * 1024) / wg_bytes);
        if (wgs > 2) wgs = 2;
        if (wgs < 1) { nmpc_free(h); return NMPC_ERR_BAD_PROBLEM; }
        per_cu = wgs * nmpc::TEAM_WAVES;
        h->team_lds = wg_bytes;
        // more than 64 KB of dynamic LDS per workgroup has to be asked for
        const int bytes = (int)wg_bytes;
        e = hipFuncSetAttribute((const void *)nmpc::nmpc_solve_hyb_kernel<nmpc::ShapeDefault>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nmpc::nmpc_solve_hyb_kernel<nmpc::ShapeNobs50>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nmpc::nmpc_solve_hyb_kernel<nmpc::ShapeAny>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { nmpc_free(h); return NMPC_ERR_HIP; }
    } else {
        // two stages per lane: one wave per SIMD (512 registers), the four waves of a CU are one team
        const size_t wg_bytes = nmpc::TEAM_WAVES * (size_t)nmpc::lds_layout2(pb->N, pb->nobs, pb->ndyn).total * sizeof(double) +
                                nmpc::TEAM_CTL_INTS * sizeof(int);
        if (wg_bytes > 160 * 1024) { nmpc_free(h); return NMPC_ERR_BAD_PROBLEM; }
        per_cu = nmpc::TEAM_WAVES;
        h->team_lds = wg_bytes;
        e = hipFuncSetAttribute((const void *)nmpc::nmpc_solve_hyb2_kernel<nmpc::ShapeN40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nmpc::nmpc_solve_hyb2_kernel<nmpc::ShapeAny>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nmpc::nmpc_eval2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { nmpc_free(h); return NMPC_ERR_HIP; }
    }
#ifdef NMPC_EXPERIMENTS
    if (const char *env = getenv("NMPC_WAVES_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v <= per_cu && v % nmpc::TEAM_WAVES == 0) per_cu = v;
    }
#endif
    if (per_cu < 1) per_cu = 1;
    h->grid_cap = prop.multiProcessorCount * per_cu;
    *out = h;
    return NMPC_OK;
}

void nmpc_free(nmpc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_queue); (void)hipFree(h->d_order); (void)hipFree(h->d_cls);
    (void)hipFree(h->d_park); (void)hipFree(h->d_pool); (void)hipFree(h->d_pool_ctr);
    for (int k = 0; k < 2; ++k) { if (h->h_pin[k]) (void)hipHostFree(h->h_pin[k]); if (h->pin_ev[k]) (void)hipEventDestroy(h->pin_ev[k]); }
    (void)hipFree(h->d_small); if (h->h_small) (void)hipHostFree(h->h_small);
    for (int k = 0; k < 2; ++k) if (h->small_ev[k]) (void)hipEventDestroy(h->small_ev[k]);
    (void)hipFree(h->d_p); (void)hipFree(h->d_u); (void)hipFree(h->d_y0); (void)hipFree(h->d_c0); (void)hipFree(h->d_yout);
    (void)hipFree(h->d_psi); (void)hipFree(h->d_grad); (void)hipFree(h->d_F1); (void)hipFree(h->d_F2); (void)hipFree(h->d_st);
    delete h;
}

int nmpc_ping(const nmpc_handle *h) { return (h && h->alive) ? NMPC_OK : NMPC_ERR_DEAD_HANDLE; }
const char *nmpc_last_error(const nmpc_handle *h) { return h ? h->err.c_str() : "null handle"; }
double nmpc_last_batch_ms(const nmpc_handle *h) { return h ? h->last_ms : 0.0; }
const char *nmpc_kernel_name(const nmpc_handle *h)
{
    if (!h) return "";
    if (h->P == 20)
        return h->shape_default ? "nmpc_solve_hyb_kernel<ShapeDefault>"
                                : (h->shape_nobs50 ? "nmpc_solve_hyb_kernel<ShapeNobs50>" : "nmpc_solve_hyb_kernel<ShapeAny>");
    return h->shape_n40 ? "nmpc_solve_hyb2_kernel<ShapeN40>" : "nmpc_solve_hyb2_kernel<ShapeAny>";
}

static void fill_args(const nmpc_handle *h, KArgs &a, int B)
{
    std::memset(&a, 0, sizeof(a));
    a.pb = h->pb; a.op = h->op; a.map = h->map; a.B = B;
    a.n_p = nmpc_n_p(&h->pb); a.n_u = nmpc_n_u(&h->pb); a.n1 = nmpc_n1(&h->pb); a.n2 = nmpc_n2(&h->pb);
    a.queue = h->d_queue;
    a.inv_ts = 1.0 / h->pb.ts;
#ifdef NMPC_EXPERIMENTS
    if (const char *env = getenv("NMPC_DEBUG_PRIO")) a.dbg = atoi(env);
#endif
}

int nmpc_solve_batch_device(nmpc_handle *h, int B, const double *d_p, double *d_u, const double *d_y0,
                            const double *d_c0, double *d_y_out, nmpc_status *d_status, void *stream)
{
    if (!h) return NMPC_ERR_BAD_ARG;
    if (!h->alive) return NMPC_ERR_DEAD_HANDLE;
    if (B < 0 || B > h->max_batch || (B > 0 && (!d_p || !d_u))) return fail(h, NMPC_ERR_BAD_ARG, "bad batch arguments");
    if (B == 0) return NMPC_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->device));
    KArgs a;
    fill_args(h, a, B);
    a.p = d_p; a.u = d_u; a.y0 = d_y0; a.c0 = d_c0; a.y_out = d_y_out; a.st = d_status;
    HIP_TRY(h, hipMemsetAsync(h->d_queue, 0, sizeof(unsigned int), s));
    // one instance per wave, three query points per pass: N_hor <= 20 with one stage per lane (hybrid / tri layouts), 20 < N_hor <= 40 with two
    const int grid = B < h->grid_cap ? B : h->grid_cap;          // waves that take instances
    if (B > grid) {        // more instances than resident waves: hand the hard-looking ones out first
        if (h->use_order) {
            if (h->order_hint) hipLaunchKernelGGL(nmpc::nmpc_classify_prev_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, h->order_hint, h->d_cls);
            else hipLaunchKernelGGL(nmpc::nmpc_classify_kernel, dim3((B + 255) / 256), dim3(256), 0, s, a, h->d_cls);
            hipLaunchKernelGGL(nmpc::nmpc_order_kernel, dim3(1), dim3(1024), 0, s, B, h->d_cls, h->d_order);
            a.order = h->d_order;
        }
        if ((h->P == 20 && (h->park_min > 0 || h->sched_mode > 0)) || (h->P == 40 && h->sched_mode > 0)) {      // instances may leave their wave at outer-iteration boundaries
            const size_t cap = (size_t)B;                 // ring buffers of this launch: an instance waits in at most one slot at a time
            const size_t cap_max = (size_t)h->max_batch;
            if (!h->d_park || !h->d_pool || !h->d_pool_ctr) {      // (all three or none: a half-made set is freed and made again)
                (void)hipFree(h->d_park); (void)hipFree(h->d_pool); (void)hipFree(h->d_pool_ctr);
                h->d_park = nullptr; h->d_pool = nullptr; h->d_pool_ctr = nullptr;
                HIP_TRY(h, hipMalloc((void **)&h->d_park, (size_t)h->max_batch * nmpc::park_stride(h->pb.N) * 8));
                HIP_TRY(h, hipMalloc((void **)&h->d_pool, nmpc::NPOOLS * cap_max * sizeof(int)));
                HIP_TRY(h, hipMalloc((void **)&h->d_pool_ctr, (4 * nmpc::NPOOLS + 2) * sizeof(unsigned int)));
            }
            HIP_TRY(h, hipMemsetAsync(h->d_pool, 0xFF, nmpc::NPOOLS * cap * sizeof(int), s));
            HIP_TRY(h, hipMemsetAsync(h->d_pool_ctr, 0, (4 * nmpc::NPOOLS + 2) * sizeof(unsigned int), s));
            a.park_min = h->P == 20 ? h->park_min : 0; a.park_depth = h->park_depth;      // (the slot migration is the one-stage kernel's: two waves per SIMD)
            a.park = h->d_park; a.pool = h->d_pool; a.pool_ctr = h->d_pool_ctr; a.pool_cap = (int)cap;
            a.sched_mode = h->sched_mode;
        }
    }
    {
        // teams of four waves.  With fewer instances than workgroups fit on the chip every instance gets a workgroup of its
        // own (one wave solves, three help from the first iteration on: the small-batch / latency mode); otherwise as many
        // waves per workgroup take instances as it needs for all of them to start at once, up to all four.
        const int max_wgs = h->grid_cap / nmpc::TEAM_WAVES;
        int owners = (B + max_wgs - 1) / max_wgs;
        if (owners > nmpc::TEAM_WAVES) owners = nmpc::TEAM_WAVES;
        if (h->team_owners_forced > 0) owners = h->team_owners_forced;
        int wgs = (grid + owners - 1) / owners;
        if (wgs > max_wgs) wgs = max_wgs;
        a.team_owners = owners;
        a.sched_long_cap = (int)(h->sched_theta * (double)(wgs * owners));
        a.sched_cold_cap = (int)(h->sched_cold * (double)(wgs * owners));
        a.team_help = h->team_help;
        a.cull_radius = h->cull_radius;
        const size_t tlds = h->team_lds;
        if (h->P == 40) {
            if (h->shape_n40) hipLaunchKernelGGL(nmpc::nmpc_solve_hyb2_kernel<nmpc::ShapeN40>, dim3(wgs), dim3(64 * nmpc::TEAM_WAVES), tlds, s, a);
            else hipLaunchKernelGGL(nmpc::nmpc_solve_hyb2_kernel<nmpc::ShapeAny>, dim3(wgs), dim3(64 * nmpc::TEAM_WAVES), tlds, s, a);
        }
        else if (h->shape_default) hipLaunchKernelGGL(nmpc::nmpc_solve_hyb_kernel<nmpc::ShapeDefault>, dim3(wgs), dim3(64 * nmpc::TEAM_WAVES), tlds, s, a);
        else if (h->shape_nobs50) hipLaunchKernelGGL(nmpc::nmpc_solve_hyb_kernel<nmpc::ShapeNobs50>, dim3(wgs), dim3(64 * nmpc::TEAM_WAVES), tlds, s, a);
        else hipLaunchKernelGGL(nmpc::nmpc_solve_hyb_kernel<nmpc::ShapeAny>, dim3(wgs), dim3(64 * nmpc::TEAM_WAVES), tlds, s, a);
    }
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

int nmpc_eval_batch_device(nmpc_handle *h, int B, const double *d_p, const double *d_u, const double *d_c,
                           const double *d_y, double *d_psi, double *d_grad, double *d_F1, double *d_F2,
                           void *stream)
{
    if (!h) return NMPC_ERR_BAD_ARG;
    if (!h->alive) return NMPC_ERR_DEAD_HANDLE;
    if (B < 0 || B > h->max_batch || (B > 0 && (!d_p || !d_u))) return fail(h, NMPC_ERR_BAD_ARG, "bad batch arguments");
    if (B == 0) return NMPC_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->device));
    KArgs a;
    fill_args(h, a, B);
    a.p = d_p; a.u = const_cast<double *>(d_u);
    a.ev_c = d_c; a.ev_y = d_y; a.ev_psi = d_psi; a.ev_grad = d_grad; a.ev_F1 = d_F1; a.ev_F2 = d_F2;
    if (h->P == 40) {          // two stages per lane: three instances per wave
        const size_t lds2 = (size_t)nmpc::lds_layout2(h->pb.N, h->pb.nobs, h->pb.ndyn).total * sizeof(double) * 3;
        hipLaunchKernelGGL(nmpc::nmpc_eval2_kernel, dim3((B + 2) / 3), dim3(64), lds2, s, a);
        HIP_TRY(h, hipGetLastError());
        return NMPC_OK;
    }
    const int K = 3;                       // instances per wave: the tri layout
    const int grid = (B + K - 1) / K;
    const size_t lds = (size_t)h->map.total * sizeof(double) * K;
    hipLaunchKernelGGL(nmpc::nmpc_eval_kernel<20>, dim3(grid), dim3(64), lds, s, a);
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

// ---- host path: staging buffers sized for max_batch, allocated on first use ----
static constexpr size_t PIN_CHUNK = 4u << 20;
static int ensure_staging(nmpc_handle *h)
{
    if (h->staging_ready) return NMPC_OK;
    if (h->d_p) return fail(h, NMPC_ERR_HIP, "staging buffers: an earlier allocation failed half way");
    const size_t B = (size_t)h->max_batch;
    const size_t np = nmpc_n_p(&h->pb), nu = nmpc_n_u(&h->pb), n1 = nmpc_n1(&h->pb), n2 = nmpc_n2(&h->pb) + 1;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMalloc((void **)&h->d_p, B * np * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_u, B * nu * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_y0, B * n1 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_c0, B * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_yout, B * n1 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_psi, B * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_grad, B * nu * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_F1, B * n1 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_F2, B * n2 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_st, B * sizeof(nmpc_status)));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(h, hipHostMalloc((void **)&h->h_pin[k], PIN_CHUNK, hipHostMallocDefault));
        HIP_TRY(h, hipEventCreateWithFlags(&h->pin_ev[k], hipEventDisableTiming));
    }
    h->staging_ready = true;
    return NMPC_OK;
}

// Host buffers of the caller are pageable: a plain hipMemcpy moves them at ~3 GB/s.  They go through two pinned 4 MB bounce
// buffers instead -- the CPU fills one while the DMA engine drains the other -- in stream order with the kernels (null stream).
static hipError_t h2d_staged(nmpc_handle *h, void *dst, const void *src, size_t bytes)
{
    hipError_t e = hipSuccess;
    int k = 0;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += PIN_CHUNK, k ^= 1) {
        const size_t n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
        e = hipEventSynchronize(h->pin_ev[k]);                       // (the copy that last used this buffer; a fresh event is complete)
        if (e != hipSuccess) break;
        std::memcpy(h->h_pin[k], (const char *)src + off, n);
        e = hipMemcpyAsync((char *)dst + off, h->h_pin[k], n, hipMemcpyHostToDevice, nullptr);
        if (e == hipSuccess) e = hipEventRecord(h->pin_ev[k], nullptr);
    }
    return e;
}
static hipError_t d2h_staged(nmpc_handle *h, void *dst, const void *src, size_t bytes)
{
    hipError_t e = hipSuccess;
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
    int k = 0;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += PIN_CHUNK, k ^= 1) {
        const size_t n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
        if (pend_n[k]) {                                             // drain what this buffer still holds
            e = hipEventSynchronize(h->pin_ev[k]);
            if (e != hipSuccess) break;
            std::memcpy((char *)dst + pend_off[k], h->h_pin[k], pend_n[k]);
        }
        e = hipMemcpyAsync(h->h_pin[k], (const char *)src + off, n, hipMemcpyDeviceToHost, nullptr);
        if (e == hipSuccess) e = hipEventRecord(h->pin_ev[k], nullptr);
        pend_off[k] = off; pend_n[k] = n;
    }
    for (int j = 0; j < 2 && e == hipSuccess; ++j, k ^= 1)           // the last one or two chunks, oldest first
        if (pend_n[k]) {
            e = hipEventSynchronize(h->pin_ev[k]);
            if (e == hipSuccess) std::memcpy((char *)dst + pend_off[k], h->h_pin[k], pend_n[k]);
            pend_n[k] = 0;
        }
    return e;
}

static constexpr int SMALL_BATCH = 16;      // instances the small-batch arena holds
static int solve_small_host(nmpc_handle *h, int B, const double *p, double *u, const double *y0, const double *c0,
                            double *y_out, nmpc_status *status)
{
    const size_t np = nmpc_n_p(&h->pb), nu = nmpc_n_u(&h->pb), n1 = nmpc_n1(&h->pb);
    const size_t cap = SMALL_BATCH;
    const size_t o_p = 0, o_c = o_p + cap * np * 8, o_y = o_c + cap * 8, o_u = o_y + cap * n1 * 8, o_yo = o_u + cap * nu * 8,
                 o_st = o_yo + cap * n1 * 8, total = o_st + cap * sizeof(nmpc_status);
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->small_ready) {
        // all or none: what an earlier, failed attempt left behind is released first, so a transient failure costs one call, not the handle's small-batch path
        (void)hipFree(h->d_small); h->d_small = nullptr;
        if (h->h_small) { (void)hipHostFree(h->h_small); h->h_small = nullptr; }
        for (int k = 0; k < 2; ++k) if (h->small_ev[k]) { (void)hipEventDestroy(h->small_ev[k]); h->small_ev[k] = nullptr; }
        HIP_TRY(h, hipMalloc((void **)&h->d_small, total));
        HIP_TRY(h, hipHostMalloc((void **)&h->h_small, total, hipHostMallocDefault));
        for (int k = 0; k < 2; ++k) HIP_TRY(h, hipEventCreate(&h->small_ev[k]));
        h->small_ready = true;
    }
    char *hs = h->h_small, *ds = h->d_small;
    // in: p .. u of the B instances (each array at its arena offset; only what is used travels, as one copy from the first to the last byte used)
    std::memcpy(hs + o_p, p, B * np * 8);
    if (c0) std::memcpy(hs + o_c, c0, B * 8);
    if (y0) std::memcpy(hs + o_y, y0, B * n1 * 8);
    std::memcpy(hs + o_u, u, B * nu * 8);
    HIP_TRY(h, hipMemcpyAsync(ds, hs, o_u + B * nu * 8, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(h, hipEventRecord(h->small_ev[0], nullptr));
    const int rc = nmpc_solve_batch_device(h, B, (const double *)(ds + o_p), (double *)(ds + o_u), y0 ? (const double *)(ds + o_y) : nullptr,
                                           c0 ? (const double *)(ds + o_c) : nullptr, (double *)(ds + o_yo), (nmpc_status *)(ds + o_st), nullptr);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->small_ev[1], nullptr));
    HIP_TRY(h, hipMemcpyAsync(hs + o_u, ds + o_u, total - o_u, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->small_ev[0], h->small_ev[1]));
    h->last_ms = (double)ms;
    std::memcpy(u, hs + o_u, B * nu * 8);
    if (y_out) std::memcpy(y_out, hs + o_yo, B * n1 * 8);
    if (status) std::memcpy(status, hs + o_st, B * sizeof(nmpc_status));
    return NMPC_OK;
}

int nmpc_solve_batch_host(nmpc_handle *h, int B, const double *p, double *u, const double *y0, const double *c0,
                          double *y_out, nmpc_status *status)
{
    if (!h) return NMPC_ERR_BAD_ARG;
    if (!h->alive) return NMPC_ERR_DEAD_HANDLE;
    if (B < 0 || B > h->max_batch || (B > 0 && (!p || !u))) return fail(h, NMPC_ERR_BAD_ARG, "bad batch arguments");
    if (B == 0) return NMPC_OK;
    if (B <= SMALL_BATCH) return solve_small_host(h, B, p, u, y0, c0, y_out, status);
    int rc = ensure_staging(h);
    if (rc) return rc;
    const size_t np = nmpc_n_p(&h->pb), nu = nmpc_n_u(&h->pb), n1 = nmpc_n1(&h->pb);
    HIP_TRY(h, h2d_staged(h, h->d_p, p, B * np * 8));
    HIP_TRY(h, h2d_staged(h, h->d_u, u, B * nu * 8));
    if (y0) HIP_TRY(h, h2d_staged(h, h->d_y0, y0, B * n1 * 8));
    if (c0) HIP_TRY(h, h2d_staged(h, h->d_c0, c0, B * 8));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = hipEventRecord(e0, nullptr);
    if (he == hipSuccess) {
        rc = nmpc_solve_batch_device(h, B, h->d_p, h->d_u, y0 ? h->d_y0 : nullptr, c0 ? h->d_c0 : nullptr,
                                     h->d_yout, h->d_st, nullptr);
        if (rc == NMPC_OK) {
            he = hipEventRecord(e1, nullptr);
            if (he == hipSuccess) he = hipDeviceSynchronize();
            if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (he != hipSuccess) return fail(h, NMPC_ERR_HIP, "solve_batch_host", he);
    h->last_ms = (double)ms;
    HIP_TRY(h, d2h_staged(h, u, h->d_u, B * nu * 8));
    if (y_out) HIP_TRY(h, d2h_staged(h, y_out, h->d_yout, B * n1 * 8));
    if (status) HIP_TRY(h, d2h_staged(h, status, h->d_st, B * sizeof(nmpc_status)));
    return NMPC_OK;
}

int nmpc_eval_batch_host(nmpc_handle *h, int B, const double *p, const double *u, const double *c, const double *y,
                         double *psi, double *grad, double *F1, double *F2)
{
    if (!h) return NMPC_ERR_BAD_ARG;
    if (!h->alive) return NMPC_ERR_DEAD_HANDLE;
    if (B < 0 || B > h->max_batch || (B > 0 && (!p || !u))) return fail(h, NMPC_ERR_BAD_ARG, "bad batch arguments");
    if (B == 0) return NMPC_OK;
    int rc = ensure_staging(h);
    if (rc) return rc;
    const size_t np = nmpc_n_p(&h->pb), nu = nmpc_n_u(&h->pb), n1 = nmpc_n1(&h->pb), n2 = nmpc_n2(&h->pb);
    HIP_TRY(h, h2d_staged(h, h->d_p, p, B * np * 8));
    HIP_TRY(h, h2d_staged(h, h->d_u, u, B * nu * 8));
    if (y) HIP_TRY(h, h2d_staged(h, h->d_y0, y, B * n1 * 8));
    if (c) HIP_TRY(h, h2d_staged(h, h->d_c0, c, B * 8));
    rc = nmpc_eval_batch_device(h, B, h->d_p, h->d_u, c ? h->d_c0 : nullptr, y ? h->d_y0 : nullptr, h->d_psi,
                                h->d_grad, h->d_F1, h->d_F2, nullptr);
    if (rc) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    if (psi) HIP_TRY(h, hipMemcpy(psi, h->d_psi, B * 8, hipMemcpyDeviceToHost));
    if (grad) HIP_TRY(h, hipMemcpy(grad, h->d_grad, B * nu * 8, hipMemcpyDeviceToHost));
    if (F1) HIP_TRY(h, hipMemcpy(F1, h->d_F1, B * n1 * 8, hipMemcpyDeviceToHost));
    if (F2 && n2) HIP_TRY(h, hipMemcpy(F2, h->d_F2, B * n2 * 8, hipMemcpyDeviceToHost));
    return NMPC_OK;
}

// ---- receding-horizon loop on device (nmpc_loop.h) ----
struct nmpc_loop {
    nmpc_handle *h;
    nmpc::LoopArgs a;            // device pointers and constants; t / dyn_in / dyn_out / traj_row change per step
    int steps, max_steps;
    double *d_tab;               // route tables, one allocation
    double *d_dynpar, *d_state, *d_last_u, *d_dyn[2], *d_P, *d_U, *d_Y, *d_traj;
    int *d_idx;
    unsigned char *d_done;
    nmpc_status *d_st;
};

int nmpc_loop_new(nmpc_handle *h, const nmpc_route *r, int B, const double *starts, const int32_t *idx0, int K,
                  const double *dyn, int max_steps, nmpc_loop **out)
{
    if (!h || !r || !out || !starts) return NMPC_ERR_BAD_ARG;
    if (!h->alive) return NMPC_ERR_DEAD_HANDLE;
    if (B < 1 || B > h->max_batch || K < 0 || K > h->pb.ndyn || (K > 0 && !dyn) || max_steps < 0)
        return fail(h, NMPC_ERR_BAD_ARG, "bad loop arguments");
    if (idx0) for (int b = 0; b < B; ++b) if (idx0[b] < 0 || idx0[b] >= r->n_ref) return fail(h, NMPC_ERR_BAD_ARG, "idx0 out of range");
    if (r->n_ref < 1 || r->n_vert < 0 || r->n_brake < 1 || r->num_steps_taken < 1 || r->num_steps_taken > h->pb.N ||
        !r->x_ref || !r->y_ref || !r->theta_ref || !r->brake_vel || !r->brake_dist || (r->n_vert > 0 && !r->vert_xy))
        return fail(h, NMPC_ERR_BAD_ARG, "bad route");
    HIP_TRY(h, hipSetDevice(h->device));
    nmpc_loop *l = new nmpc_loop();
    std::memset(l, 0, sizeof(*l));
    l->h = h;
    l->max_steps = max_steps;
    nmpc::LoopArgs &a = l->a;
    a.B = B; a.N = h->pb.N; a.nobs = h->pb.nobs; a.ndyn = h->pb.ndyn; a.K = K;
    a.n_p = nmpc_n_p(&h->pb); a.n_u = nmpc_n_u(&h->pb);
    a.n_ref = r->n_ref; a.n_vert = r->n_vert; a.n_brake = r->n_brake; a.s = r->num_steps_taken; a.t = 0;
    a.ts = h->pb.ts; a.base = r->base_speed; a.radius = r->radius; a.pad = r->dyn_pad;
    for (int i = 0; i < 3; ++i) a.end[i] = r->end[i];
    for (int i = 0; i < 10; ++i) a.w[i] = r->weights[i];
    const size_t n1 = nmpc_n1(&h->pb), ntab = 3 * (size_t)r->n_ref + 2 * (size_t)r->n_vert + 2 * (size_t)r->n_brake;
    const size_t ndynrow = (size_t)a.ndyn * a.N * 5;
    hipError_t e = hipMalloc((void **)&l->d_tab, (ntab ? ntab : 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_dynpar, ((size_t)B * (K ? K : 1)) * 10 * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_state, (size_t)B * 3 * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_last_u, (size_t)B * 2 * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_dyn[0], (size_t)B * (ndynrow ? ndynrow : 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_dyn[1], (size_t)B * (ndynrow ? ndynrow : 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_P, (size_t)B * a.n_p * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_U, (size_t)B * a.n_u * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_Y, (size_t)B * n1 * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_idx, (size_t)B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_done, (size_t)B);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_st, (size_t)B * sizeof(nmpc_status));
    if (e == hipSuccess && max_steps > 0)
        e = hipMalloc((void **)&l->d_traj, ((size_t)max_steps * a.s + 1) * B * 3 * 8);
    if (e != hipSuccess) { nmpc_loop_free(l); return fail(h, NMPC_ERR_HIP, "nmpc_loop_new: hipMalloc", e); }
    // route tables: x_ref | y_ref | theta_ref | vertices | brake velocities | brake distances
    std::vector<double> tab(ntab);
    double *q = tab.data();
    std::memcpy(q, r->x_ref, 8 * (size_t)r->n_ref); q += r->n_ref;
    std::memcpy(q, r->y_ref, 8 * (size_t)r->n_ref); q += r->n_ref;
    std::memcpy(q, r->theta_ref, 8 * (size_t)r->n_ref); q += r->n_ref;
    if (r->n_vert) std::memcpy(q, r->vert_xy, 16 * (size_t)r->n_vert);
    q += 2 * r->n_vert;
    std::memcpy(q, r->brake_vel, 8 * (size_t)r->n_brake); q += r->n_brake;
    std::memcpy(q, r->brake_dist, 8 * (size_t)r->n_brake);
    e = hipMemcpy(l->d_tab, tab.data(), ntab * 8, hipMemcpyHostToDevice);
    a.xr = l->d_tab; a.yr = a.xr + r->n_ref; a.thr = a.yr + r->n_ref; a.vert = a.thr + r->n_ref;
    a.bv = a.vert + 2 * r->n_vert; a.bd = a.bv + r->n_brake;
    if (e == hipSuccess && K) e = hipMemcpy(l->d_dynpar, dyn, (size_t)B * K * 10 * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(l->d_state, starts, (size_t)B * 3 * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && l->d_traj) e = hipMemcpy(l->d_traj, starts, (size_t)B * 3 * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(l->d_last_u, 0, (size_t)B * 2 * 8);
    if (e == hipSuccess) e = hipMemset(l->d_U, 0, (size_t)B * a.n_u * 8);
    if (e == hipSuccess) e = hipMemset(l->d_Y, 0, (size_t)B * n1 * 8);
    if (e == hipSuccess) e = idx0 ? hipMemcpy(l->d_idx, idx0, (size_t)B * sizeof(int), hipMemcpyHostToDevice)
                                  : hipMemset(l->d_idx, 0, (size_t)B * sizeof(int));
    if (e == hipSuccess) e = hipMemset(l->d_done, 0, (size_t)B);
    if (e == hipSuccess) e = hipMemset(l->d_st, 0, (size_t)B * sizeof(nmpc_status));
    if (e == hipSuccess && ndynrow) {          // padding block: zeros with unit radii (path_generator.py:274-280)
        std::vector<double> pad((size_t)B * ndynrow, 0.0);
        for (size_t i = 0; i < pad.size(); i += 5) { pad[i + 2] = 1.0; pad[i + 3] = 1.0; }
        e = hipMemcpy(l->d_dyn[0], pad.data(), pad.size() * 8, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { nmpc_loop_free(l); return fail(h, NMPC_ERR_HIP, "nmpc_loop_new: initialisation", e); }
    a.dynpar = l->d_dynpar; a.state = l->d_state; a.last_u = l->d_last_u; a.idx = l->d_idx;
    a.P = l->d_P; a.U = l->d_U; a.done = l->d_done; a.traj = l->d_traj; a.traj_row = 1;
    *out = l;
    return NMPC_OK;
}

void nmpc_loop_free(nmpc_loop *l)
{
    if (!l) return;
    (void)hipSetDevice(l->h->device);
    (void)hipFree(l->d_tab); (void)hipFree(l->d_dynpar); (void)hipFree(l->d_state); (void)hipFree(l->d_last_u);
    (void)hipFree(l->d_dyn[0]); (void)hipFree(l->d_dyn[1]); (void)hipFree(l->d_P); (void)hipFree(l->d_U);
    (void)hipFree(l->d_Y); (void)hipFree(l->d_idx); (void)hipFree(l->d_done); (void)hipFree(l->d_st);
    (void)hipFree(l->d_traj);
    delete l;
}

int nmpc_loop_step(nmpc_loop *l, void *stream)
{
    if (!l) return NMPC_ERR_BAD_ARG;
    nmpc_handle *h = l->h;
    if (!h->alive) return NMPC_ERR_DEAD_HANDLE;
    if (l->max_steps > 0 && l->steps >= l->max_steps) return fail(h, NMPC_ERR_BAD_ARG, "trajectory buffer is full");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->device));
    nmpc::LoopArgs &a = l->a;
    const int cur = l->steps & 1;
    a.dyn_in = l->d_dyn[cur];
    a.dyn_out = l->d_dyn[cur ^ 1];
    hipLaunchKernelGGL(nmpc::nmpc_loop_assemble_kernel, dim3(a.B), dim3(64), 0, s, a);
    HIP_TRY(h, hipGetLastError());
    // warm start: previous controls and multipliers, penalty back to its initial value (the server's behaviour)
    // launch order: from the second step on, by the pass counts of the step before (read by the classification kernel ahead of the solve, which
    // then overwrites them); the first step has only the inputs to go by
    h->order_hint = (l->steps > 0 && h->loop_order_prev) ? l->d_st : nullptr;
    const int rc = nmpc_solve_batch_device(h, a.B, l->d_P, l->d_U, l->d_Y, nullptr, l->d_Y, l->d_st, stream);
    h->order_hint = nullptr;
    if (rc) return rc;
    hipLaunchKernelGGL(nmpc::nmpc_loop_advance_kernel, dim3((a.B + 255) / 256), dim3(256), 0, s, a);
    HIP_TRY(h, hipGetLastError());
    a.t += a.s;
    a.traj_row += a.s;
    l->steps++;
    return NMPC_OK;
}

int nmpc_loop_read(nmpc_loop *l, double *state, double *last_u, int32_t *idx, uint8_t *done, nmpc_status *status)
{
    if (!l) return NMPC_ERR_BAD_ARG;
    nmpc_handle *h = l->h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const size_t B = (size_t)l->a.B;
    if (state) HIP_TRY(h, hipMemcpy(state, l->d_state, B * 3 * 8, hipMemcpyDeviceToHost));
    if (last_u) HIP_TRY(h, hipMemcpy(last_u, l->d_last_u, B * 2 * 8, hipMemcpyDeviceToHost));
    if (idx) HIP_TRY(h, hipMemcpy(idx, l->d_idx, B * sizeof(int), hipMemcpyDeviceToHost));
    if (done) HIP_TRY(h, hipMemcpy(done, l->d_done, B, hipMemcpyDeviceToHost));
    if (status) HIP_TRY(h, hipMemcpy(status, l->d_st, B * sizeof(nmpc_status), hipMemcpyDeviceToHost));
    return NMPC_OK;
}

int nmpc_loop_params(nmpc_loop *l, double *p, double *u, double *y)
{
    if (!l) return NMPC_ERR_BAD_ARG;
    nmpc_handle *h = l->h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const size_t B = (size_t)l->a.B;
    if (p) HIP_TRY(h, hipMemcpy(p, l->d_P, B * l->a.n_p * 8, hipMemcpyDeviceToHost));
    if (u) HIP_TRY(h, hipMemcpy(u, l->d_U, B * l->a.n_u * 8, hipMemcpyDeviceToHost));
    if (y) HIP_TRY(h, hipMemcpy(y, l->d_Y, B * (size_t)nmpc_n1(&h->pb) * 8, hipMemcpyDeviceToHost));
    return NMPC_OK;
}

int nmpc_loop_trajectory(nmpc_loop *l, double *rows, int max_rows)
{
    if (!l || !rows) return NMPC_ERR_BAD_ARG;
    nmpc_handle *h = l->h;
    if (!l->d_traj) return fail(h, NMPC_ERR_BAD_ARG, "the loop was created without a trajectory buffer");
    const int nrows = l->steps * l->a.s + 1;
    if (max_rows < nrows) return fail(h, NMPC_ERR_BAD_ARG, "trajectory does not fit");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(rows, l->d_traj, (size_t)nrows * l->a.B * 3 * 8, hipMemcpyDeviceToHost));
    return nrows;
}

// ---- arithmetic primitives, for bit-level checks against the oracle ----
__global__ void nmpc_test_sincos_kernel(int n, const double *x, double *s, double *c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) nmpc::sincos_cw(x[i], s[i], c[i]);
}
__global__ void nmpc_test_divsqrt_kernel(int n, const double *a, const double *b, double *q, double *r)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { q[i] = a[i] / b[i]; r[i] = sqrt(a[i]); }
}

static int run_unary_test(nmpc_handle *h, int n, const double *x0, const double *x1, double *o0, double *o1, int which)
{
    if (!h || n < 0 || !x0 || !o0 || !o1) return NMPC_ERR_BAD_ARG;
    if (n == 0) return NMPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    double *d[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipMalloc((void **)&d[i], (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpy(d[0], x0, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && x1) e = hipMemcpy(d[1], x1, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int blocks = (n + 255) / 256;
        if (which == 0) hipLaunchKernelGGL(nmpc_test_sincos_kernel, dim3(blocks), dim3(256), 0, nullptr, n, d[0], d[2], d[3]);
        else hipLaunchKernelGGL(nmpc_test_divsqrt_kernel, dim3(blocks), dim3(256), 0, nullptr, n, d[0], d[1], d[2], d[3]);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(o0, d[2], (size_t)n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(o1, d[3], (size_t)n * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) (void)hipFree(d[i]);
    return e == hipSuccess ? NMPC_OK : fail(h, NMPC_ERR_HIP, "arithmetic primitive test", e);
}

int nmpc_test_sincos_host(nmpc_handle *h, int n, const double *x, double *out_s, double *out_c)
{
    return run_unary_test(h, n, x, nullptr, out_s, out_c, 0);
}
int nmpc_test_divsqrt_host(nmpc_handle *h, int n, const double *a, const double *b, double *out_div, double *out_sqrt)
{
    if (!b) return NMPC_ERR_BAD_ARG;
    return run_unary_test(h, n, a, b, out_div, out_sqrt, 1);
}

#ifdef NMPC_TL
// experiments only (scripts/timeline.py)
int nmpc_debug_timeline(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(nmpc::nmpc_tl), 64 * 16 * sizeof(long long)) == hipSuccess ? NMPC_OK : NMPC_ERR_HIP; }
#endif
#ifdef NMPC_BBCOUNT
// experiments only (scripts/bbcount.py): executions of every basic block of the instrumented kernel since the last reset
int nmpc_debug_bbcount(unsigned int *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(nmpc::nmpc_bbcnt), 4096 * sizeof(unsigned int));
    if (e == hipSuccess && reset) { static const unsigned int z[4096] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(nmpc::nmpc_bbcnt), z, sizeof z); }
    return e == hipSuccess ? NMPC_OK : NMPC_ERR_HIP;
}
#endif
#ifdef NMPC_WIN_STATS
// experiments only (scripts/win_stats.py): windowed cross-track searches and how many of them fell back to the full scan
int nmpc_debug_win_stats(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(nmpc::nmpc_win_stats), 4 * sizeof(unsigned long long));      // windows: tried | fell back; obstacle certificates: tried | scanned
    if (e == hipSuccess && reset) { const unsigned long long z[4] = {0, 0, 0, 0}; e = hipMemcpyToSymbol(HIP_SYMBOL(nmpc::nmpc_win_stats), z, sizeof z); }
    return e == hipSuccess ? NMPC_OK : NMPC_ERR_HIP;
}
#endif

}  // extern "C"
