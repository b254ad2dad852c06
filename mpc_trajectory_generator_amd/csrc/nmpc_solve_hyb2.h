// nmpc_solve_hyb2.h -- the 20 < N_hor <= 40 solver (BASELINE config 2: N_hor = 40): the design of nmpc_solve_hyb.h with
// TWO STAGES PER LANE.
//
// One problem instance per wavefront, THREE query points per pass: the evaluation runs in the "tri" lane layout of
// nmpc_device.h -- three groups of 20 lanes -- and lane j of a group holds stages 2j and 2j + 1 of the horizon (their
// controls, post-update states and adjoints), so a 40-stage horizon fits where nmpc_solve_hyb.h keeps 20.  The solver
// state lives, as there, in both 32-lane halves of the wave (lane j < 20: stages 2j, 2j + 1).  Per-lane work doubles,
// cross-lane work does not: a tree sum adds the lane's two stages first and then runs the 20-lane tree of the one-stage
// layout (that IS the canonical adjacent-pair tree over 64 zero-padded entries, oracle/nmpc_oracle.c tree_sum_p); a
// stage shift crosses lanes for one of the two stages only; the cross-track loop reads a segment once for two stages.
// Prefix / suffix sums use the PAIR form of the canonical scans (oracle: pair_prefix / pair_suffix): pair sums, the
// 20-lane Kogge-Stone scan over them, then the first stage of a pair adds its own value to the exclusive result.
//
// Query points travel from the state layout to the evaluation layout, and gradients back, through LDS (a lane holds
// four doubles per vector here; as ds_bpermute traffic that would be sixteen instructions per vector).
// The kernel needs more than 256 VGPRs: one wave per SIMD, four per CU -- the four waves of a CU are one TEAM
// (nmpc_solve_hyb.h: a wave without work of its own evaluates line-search trials for its siblings).  With one wave per
// SIMD there is no favoured wave slot, hence no migration.
//
// Same sequential semantics, same counters and -- against the oracle -- the same bits as every other solve kernel.
#pragma once

namespace nmpc {

// the two stages (2j, 2j + 1) a lane holds
struct D2 { double a, b; };
__device__ __forceinline__ D2 d2s(double s) { return D2{s, s}; }
__device__ __forceinline__ D2 operator+(D2 x, D2 y) { return D2{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ D2 operator-(D2 x, D2 y) { return D2{x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ D2 operator*(D2 x, D2 y) { return D2{x.a * y.a, x.b * y.b}; }
__device__ __forceinline__ D2 operator*(double s, D2 y) { return D2{s * y.a, s * y.b}; }
__device__ __forceinline__ D2 operator-(D2 x) { return D2{-x.a, -x.b}; }
__device__ __forceinline__ D2 fma2(D2 x, D2 y, D2 z) { return D2{fma(x.a, y.a, z.a), fma(x.b, y.b, z.b)}; }
__device__ __forceinline__ D2 fma2(double s, D2 y, D2 z) { return D2{fma(s, y.a, z.a), fma(s, y.b, z.b)}; }
__device__ __forceinline__ D2 fma2(double s, D2 y, double z) { return D2{fma(s, y.a, z), fma(s, y.b, z)}; }
__device__ __forceinline__ D2 sel2(bool ca, bool cb, D2 x, D2 y) { return D2{ca ? x.a : y.a, cb ? x.b : y.b}; }
__device__ __forceinline__ D2 clamp2(D2 x, double lo, double hi) { return D2{clampd(x.a, lo, hi), clampd(x.b, lo, hi)}; }

// ---- evaluation layout (three groups of 20 lanes, nmpc_device.h "tri"), two stages per lane
__device__ __forceinline__ double gsum2(D2 v, int lane) { return group_sum<20>(v.a + v.b, lane); }
__device__ __forceinline__ D2 gprefix2(D2 v, int lane)
{
    double E;
    const double W = group_prefix_ex<20>(v.a + v.b, lane, E);      // (the exclusive sum comes with the scan's own carry exchange)
    return D2{E + v.a, W};
}
__device__ __forceinline__ D2 gsuffix2(D2 v, int lane)
{
    const double Z = group_suffix<20>(v.a + v.b, lane);
    return D2{Z, from_next<20>(Z, lane) + v.b};
}
// value of the stage before / after each of the lane's stages
__device__ __forceinline__ D2 prev2(D2 v, int lane, double fill) { return D2{from_prev<20>(v.b, lane, fill), v.a}; }
__device__ __forceinline__ D2 next2(D2 v, int lane) { return D2{v.b, from_next<20>(v.a, lane)}; }

// ---- state layout (stage pair j at lane j of both 32-lane halves)
__device__ __forceinline__ double hdot2(D2 av, D2 aw, D2 bv, D2 bw) { return fma(av.a, bv.a, aw.a * bw.a) + fma(av.b, bv.b, aw.b * bw.b); }

// A (v, w) x two-stage vector in LDS: CNT entries of four doubles, kept as TWO planes of 16-byte pairs -- plane 0: {v_a, w_a} of every entry,
// plane 1: {v_b, w_b} -- so that the lanes of a ds_read_b128 / ds_write_b128 are 16 bytes apart (as one 32-byte record per entry they were 32 bytes
// apart: lanes i and i + 4 of every eight hit the same banks, a two-way conflict on every access of the ring, the parked columns and the point /
// gradient / request areas; round 3's "address-bit swizzle" against it cost more in address arithmetic than it saved -- the planes cost nothing:
// the second access is the first one's address with another immediate offset).  `blk` is the vector's base, `e` the entry.
template <int CNT> __device__ __forceinline__ void ld4(const lds_double2 *blk, int e, D2 &v, D2 &w)
{
    const dbl2 c0 = blk[e], c1 = blk[CNT + e];
    v = D2{c0.x, c1.x};
    w = D2{c0.y, c1.y};
}
template <int CNT> __device__ __forceinline__ void st4(lds_double2 *blk, int e, D2 v, D2 w)
{
    blk[e] = dbl2{v.a, w.a};
    blk[CNT + e] = dbl2{v.b, w.b};
}

// LDS slice of one wave (offsets in doubles); every per-stage table has room for 24 lane pairs = 48 stages
struct LdsMap2 {
    int sc, cw, par, seg, obs, f2, rho;
    int dyn;      // NDYN_MAX x 6 x 48: [ellipse][field][stage]
    int vr;       // reference speed by stage (48)
    int pts;      // query points of a pass: X of half 0 | X of half 1 | Y, 24 entries of 4 doubles each
    int grd;      // gradients of the pass's three points, the same shape
    int req;      // team request: u | r | d, the same shape
    int vec;      // 7 parked state-layout vectors: 24 entries of 4 doubles each (state lanes 24..31 share entry 23: zeros)
    int gsy, gyy; // Gram-form L-BFGS (nmpc_solve_hyb.h): the kept inner products [slot][slot]
    int S, Y;     // L-BFGS ring: MAXMEM slots x 21 entries (20 lane pairs + a zero column) of 4 doubles
    int nv;       // ... the four vectors of an iteration -- s | y | r | g -- in the ring's shape (shares the place of pts | grd)
    int total;
};
#ifndef NMPC_WIN2
// Half width of the cross-track window of the two-stage kernel.  Measured on config 2: 1 -> 174-179 ms, 2 -> 176-180 ms, 0 (no windows) -> 196 ms.
// (Round 3 saw a W = 2 build whose results changed from run to run and blamed the window.  The cause was the compiler: register-allocator copies
// in front of an EXEC restore, codegen_check.py; W = 2 is exact and repeatable under all four scheduler strategies, only not faster.)
#define NMPC_WIN2 1
#endif
constexpr int H2_COLS = 24, H2_NS = 21, H2_ENT = 24;
constexpr int TEAM2_AREA_DOUBLES = 3 * H2_ENT * 4 + 8;
__host__ __device__ constexpr LdsMap2 lds_layout2(int N, int nobs, int ndyn)
{
    LdsMap2 mp{};
    int o = 0;
    mp.sc = o;  o += 20;
    mp.cw = o;  o += CW_NCOEF;
    mp.par = o; o += 20;
    mp.seg = o; o += SEG_STRIDE * (N + 5);
    mp.obs = o; o += OBS_STRIDE * (nobs + 4);
    mp.f2 = o;  o += 0;                 // (only the cost-layer kernel writes F2, and it has no parked vectors: the array shares their place)
    mp.rho = o; o += MAXMEM;
    o = (o + 1) & ~1;
    mp.dyn = o; o += NDYN_MAX * 6 * 48;
    mp.vr = o;  o += 48;
    mp.pts = o; o += 3 * H2_ENT * 4;
    mp.grd = o; o += 3 * H2_ENT * 4;
    mp.req = o; o += 3 * H2_ENT * 4;
    mp.vec = o; o += 7 * H2_COLS * 4;
    mp.f2 = mp.vec;
    if (7 * H2_COLS * 4 < 3 * (nobs + ndyn + 1)) o = mp.vec + 3 * (nobs + ndyn + 1);
    o = (o + 1) & ~1;
    mp.gsy = o; o += MAXMEM * GRAM_LD;   // gsy | gyy | S | Y are contiguous (zeroed together when the buffer is reset)
    mp.gyy = o; o += MAXMEM * GRAM_LD;
    mp.S = o;   o += MAXMEM * H2_NS * 4;
    mp.Y = o;   o += MAXMEM * H2_NS * 4;
    // (the four vectors live from the top of a pass to the end of its L-BFGS phase; the query points and gradients of the pass, which are
    // written after that and dead before the next one, share their place: 336 of 576 doubles)
    mp.nv = mp.pts;
    mp.total = (o + 1) & ~1;
    return mp;
}
template <class SH> __device__ __forceinline__ LdsMap2 the_map2(const KArgs &a)
{
    if constexpr (SH::N > 0 && SH::NOBS >= 0 && SH::NDYN >= 0) return lds_layout2(SH::N, SH::NOBS, SH::NDYN);
    else return lds_layout2(a.pb.N, a.pb.nobs, a.pb.ndyn);
}

// ---------------------------------------------------------------------------------------------
// instance set-up: p -> LDS slice     (reference mpc_generator.py:73-79,93-104,127-136)
// ---------------------------------------------------------------------------------------------
template <class SH>
__device__ __forceinline__ void prepare_instance2(const KArgs &a, lds_double *L, const LdsMap2 &mp, const double *p, int lane)
{
    const int N = shape_N<SH>(a), nobs = shape_nobs<SH>(a), ndyn = shape_ndyn<SH>(a);
    if (lane < 8) L[mp.sc + lane] = p[lane];                          // state, last input, target (p[8:10] unused)
    if (lane >= 8 && lane < 18) L[mp.sc + lane] = p[lane + 2];        // ten weights p[10:20]
    if (lane < CW_NCOEF) L[mp.cw + lane] = CW_COEF_DEV[lane];
    NMPC_WAVE_SYNC();
    if (lane < 48) L[mp.vr + lane] = lane < N ? p[NZ + lane] : 0.0;
    const double *ps = p + NZ + N;
    for (int k = lane; k < ((nobs + 3) & ~3); k += 64) {               // padded to a multiple of 4 with inert zero circles
        const bool real = k < nobs;
        const double r = real ? ps[3 * k + 2] : 0.0;
        L[mp.obs + OBS_STRIDE * k] = real ? ps[3 * k] : 0.0;
        L[mp.obs + OBS_STRIDE * k + 1] = real ? ps[3 * k + 1] : 0.0;
        L[mp.obs + OBS_STRIDE * k + 2] = r * r;
        L[mp.obs + OBS_STRIDE * k + 3] = r > 0.0 ? r : -1e30;      // (obstacle certificate: an empty slot is infinitely far away)
    }
    const double *pd = ps + 3 * nobs;
    if (lane < 48) {                                                   // stage `lane`: one column of the ellipse tables
        lds_double *col = L + mp.dyn + lane;
#pragma unroll
        for (int k = 0; k < NDYN_MAX; ++k) {
            double ex = 0.0, ey = 0.0, ca = 0.0, sa = 0.0, irx2 = 1.0, iry2 = 1.0;
            if (k < ndyn && lane < N) {
                const double *e = pd + (k * N + lane) * 5;
                ex = e[0];
                ey = e[1];
                irx2 = 1.0 / (e[2] * e[2]);
                iry2 = 1.0 / (e[3] * e[3]);
                sincos_cw_t(e[4], (const lds_double *)(L + mp.cw), sa, ca);
            }
            col[(k * DY_FIELDS + DY_EX) * 48] = ex;
            col[(k * DY_FIELDS + DY_EY) * 48] = ey;
            col[(k * DY_FIELDS + DY_CA) * 48] = ca;
            col[(k * DY_FIELDS + DY_SA) * 48] = sa;
            col[(k * DY_FIELDS + DY_IRX2) * 48] = irx2;
            col[(k * DY_FIELDS + DY_IRY2) * 48] = iry2;
        }
    }
    const double *pr = pd + 5 * ndyn * N;
    const int nseg4 = (N - 1 + 3) & ~3;                    // the CTE loop runs 4 segments per trip; the padding
    if (lane < nseg4) {                                    // repeats the last segment (cannot change a strict min)
        const int i = lane < N - 1 ? lane : N - 2;
        const double ax = pr[3 * i], ay = pr[3 * i + 1];
        const double bx = pr[3 * i + 3], by = pr[3 * i + 4];
        const double dx = bx - ax, dy = by - ay;
        lds_double *sg = L + mp.seg + SEG_STRIDE * lane;
        sg[0] = ax;
        sg[1] = ay;
        sg[2] = dx;
        sg[3] = dy;
        sg[4] = 1.0 / (fma(dx, dx, dy * dy) + 1e-16);
    }
    NMPC_WAVE_SYNC();
}

// this lane's two columns of a per-stage table of the ellipses
__device__ __forceinline__ D2 dyn2(const lds_double *L, const LdsMap2 &mp, int te, int k, int f)
{
    const dbl2 v = ((const lds_double2 *)(L + mp.dyn + (k * DY_FIELDS + f) * 48))[te];
    return D2{v.x, v.y};
}

// ---------------------------------------------------------------------------------------------
// psi(z; c, y), grad psi, F1 (av, aw), sum_k F2_k^2 (pen) for the query point whose stages 2 te, 2 te + 1 this lane holds.
// The arithmetic per stage is that of eval_psi (nmpc_kernels.hip); what differs is which lane holds which stage.
// ---------------------------------------------------------------------------------------------
// the obstacle certificate of eval_psi (nmpc_kernels.hip: ObsCert), one per stage of the lane
struct ObsCert2 {
    D2 xo, yo, m2;
    int act_lo, act_hi, act_dyn;
};
template <class SH, bool WRITE_F2 = false, int WIN = 0>
__device__ __forceinline__ void eval_psi2(const KArgs &a, lds_double *L, const LdsMap2 &mp, int f2off, int lane, int te, D2 zv, D2 zw,
                                          double c, double cbar_inv, D2 yv, D2 yw, bool want_grad, double &psi, double &pen_out,
                                          D2 &gv, D2 &gw, D2 &av_out, D2 &aw_out, WinState *ws = nullptr, ObsCert2 *oc = nullptr)
{
    const int N = shape_N<SH>(a), nobs = shape_nobs<SH>(a), ndyn = shape_ndyn<SH>(a);
    const double ts = a.pb.ts, inv_ts = a.inv_ts;
    constexpr bool FULL = SH::N == 40;              // every stage of a stage lane is inside the horizon
    const int sa_ = 2 * te, sb_ = 2 * te + 1;       // the lane's stages
    const bool ra = sa_ < N, rb = sb_ < N;          // real stages
    const bool ina = FULL ? true : ra, inb = FULL ? true : rb;
    const lds_double *sc = L + mp.sc;
    const lds_double *cw = (const lds_double *)(L + mp.cw);
    const double x0 = sc[SC_X0], y0 = sc[SC_Y0], th0 = sc[SC_TH0];
    const double xf = sc[SC_XF], yf = sc[SC_YF], thf = sc[SC_THF];
    D2 vref;
    { const dbl2 v = ((const lds_double2 *)(L + mp.vr))[te]; vref = D2{v.x, v.y}; }

    // rollout (:88-90) as three prefix sums
    const D2 thn = fma2(ts, gprefix2(zw, lane), th0);
    const D2 th = prev2(thn, lane, th0);
    D2 sn, cs;
    sincos_cw_t(th.a, cw, sn.a, cs.a);
    sincos_cw_t(th.b, cw, sn.b, cs.b);
    const D2 xn = fma2(ts, gprefix2(zv * cs, lane), x0);
    const D2 yn = fma2(ts, gprefix2(zv * sn, lane), y0);
    const D2 xp = prev2(xn, lane, x0);
    const D2 yp = prev2(yn, lane, y0);
    const double half_c = 0.5 * c;

    D2 acc = (sc[SC_RV] * zv) * zv;                                               // (:84)
    acc = fma2(sc[SC_RW] * zw, zw, acc);
    const D2 dv = zv - vref;                                                      // (:85)
    acc = fma2(sc[SC_QV] * dv, dv, acc);
    {
        const D2 ddx = xp - d2s(xf), ddy = yp - d2s(yf), dth = th - d2s(thf);     // (:86, 59-64)
        acc = fma2(sc[SC_Q], fma2(ddx, ddx, ddy * ddy), acc);
        acc = fma2(sc[SC_QTH] * dth, dth, acc);
    }
    // (the shifts of the accelerations are issued here: their LDS round trips then run under the cross-track loop)
    const D2 vprev = prev2(zv, lane, sc[SC_VINIT]);
    const D2 wprev = prev2(zw, lane, sc[SC_WINIT]);
    // cross-track error: min over the N-1 reference segments (:121-144); a segment is read once for both stages
    D2 best = d2s(__builtin_inf());
    int bia = 0, bib = 0;
    bool full_scan = true;
    int i0a = 0, i0b = 0;               // first segment of the window whose clearance the full scan measures, per stage
    if constexpr (WIN > 0) {
        // windowed search around each stage's window centre, accepted only if provably the full scan's (eval_psi in nmpc_kernels.hip
        // has the argument): ws[0] belongs to stage a of this lane, ws[1] to stage b
        const int nseg = N - 1;
        if (nseg >= 2 * WIN + 1) {
#define NMPC2_CLAMP_WIN(C_) ((C_) - WIN < 0 ? 0 : ((C_) - WIN > nseg - (2 * WIN + 1) ? nseg - (2 * WIN + 1) : (C_) - WIN))
            { const int ca = ws[0].ctr < 0 ? 0 : (ws[0].ctr > nseg - 1 ? nseg - 1 : ws[0].ctr); i0a = NMPC2_CLAMP_WIN(ca); }
            { const int cb = ws[1].ctr < 0 ? 0 : (ws[1].ctr > nseg - 1 ? nseg - 1 : ws[1].ctr); i0b = NMPC2_CLAMP_WIN(cb); }
            if (!__any((ra & !(ws[0].mo2 > 0.0)) | (rb & !(ws[1].mo2 > 0.0)))) {      // (& |, not && ||: lane masks, no divergent branches)
                bool sure_a, sure_b;
#define NMPC2_WINDOW(S_, BI_, I0_, WS_, SURE_)                                                             \
                do {                                                                                       \
                    const lds_double *sg = L + mp.seg + SEG_STRIDE * (I0_);                                \
                    double bst = __builtin_inf();                                                          \
                    int bi_ = 0;                                                                           \
                    _Pragma("unroll") for (int j = 0; j <= 2 * WIN; ++j) {                                 \
                        const double s0 = sg[j * SEG_STRIDE], s1 = sg[j * SEG_STRIDE + 1], s2 = sg[j * SEG_STRIDE + 2], \
                                     s3 = sg[j * SEG_STRIDE + 3], s4 = sg[j * SEG_STRIDE + 4];             \
                        const double px = xn.S_ - s0, py = yn.S_ - s1;                                     \
                        const double dot = fma(s2, px, s3 * py);                                           \
                        const double tst = fmin(fmax(s4 * dot, 0.0), 1.0);                                 \
                        const double ex = fma(s2, tst, -px), ey = fma(s3, tst, -py);                       \
                        const double d2 = fma(ex, ex, ey * ey);                                            \
                        bi_ = d2 < bst ? (I0_) + j : bi_;                                                  \
                        bst = fmin(bst, d2);                                                               \
                    }                                                                                      \
                    const double ax = xn.S_ - (WS_).xr, ay = yn.S_ - (WS_).yr;                             \
                    SURE_ = window_is_global(fma(ax, ax, ay * ay), bst, (WS_).mo2);                        \
                    best.S_ = bst; BI_ = bi_;                                                              \
                } while (0)
                NMPC2_WINDOW(a, bia, i0a, ws[0], sure_a);
                NMPC2_WINDOW(b, bib, i0b, ws[1], sure_b);
#undef NMPC2_WINDOW
                full_scan = __any((ra & !sure_a) | (rb & !sure_b));
#ifdef NMPC_WIN_STATS
                if (lane == 0) { atomicAdd(&nmpc_win_stats[0], 1ull); if (full_scan) atomicAdd(&nmpc_win_stats[1], 1ull); }
#endif
                if (full_scan) {            // the full scan measures the clearance of the windows around what the old windows found nearest
                    i0a = NMPC2_CLAMP_WIN(bia); i0b = NMPC2_CLAMP_WIN(bib);
                    best = d2s(__builtin_inf()); bia = bib = 0;
                }
            }
#undef NMPC2_CLAMP_WIN
        }
    }
    if (full_scan) {
        const lds_double *sg = L + mp.seg;
        const int nseg4 = (N - 1 + 3) & ~3;
        double cur[2][5], nxt[2][5];
        D2 mout = d2s(__builtin_inf());                     // (WIN) nearest segment outside each stage's window
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int f = 0; f < 5; ++f) cur[j][f] = sg[j * SEG_STRIDE + f];
#pragma unroll SH::N > 0 ? 4 : 1      // (measured on MI355X, us per lone pass: unroll 2 8.43, 4 8.28, all twenty trips 19.6)
        for (int i = 0; i < nseg4; i += 2) {
            sg += 2 * SEG_STRIDE;                           // table is padded: reading one pair past the end is safe
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < 5; ++f) nxt[j][f] = sg[j * SEG_STRIDE + f];
            NMPC_SCHED_BARRIER();
            D2 d2v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const D2 px = xn - d2s(cur[j][0]), py = yn - d2s(cur[j][1]);
                const D2 dot = fma2(cur[j][2], px, cur[j][3] * py);
                const D2 that = cur[j][4] * dot;
                const D2 tst = D2{fmin(fmax(that.a, 0.0), 1.0), fmin(fmax(that.b, 0.0), 1.0)};
                const D2 ex = fma2(cur[j][2], tst, -px), ey = fma2(cur[j][3], tst, -py);
                d2v[j] = fma2(ex, ex, ey * ey);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {                   // strict <: the first minimum keeps its index
                bia = d2v[j].a < best.a ? i + j : bia;
                bib = d2v[j].b < best.b ? i + j : bib;
                best = D2{fmin(best.a, d2v[j].a), fmin(best.b, d2v[j].b)};
                if constexpr (WIN > 0) {                    // (a padding entry repeats the last segment)
                    const int ie = i + j < N - 1 ? i + j : N - 2;
                    mout.a = (unsigned)(ie - i0a) <= 2u * WIN ? mout.a : fmin(mout.a, d2v[j].a);
                    mout.b = (unsigned)(ie - i0b) <= 2u * WIN ? mout.b : fmin(mout.b, d2v[j].b);
                }
            }
            NMPC_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < 5; ++f) cur[j][f] = nxt[j][f];
        }
        if constexpr (WIN > 0) {            // each stage's certificate for the evaluations to come (eval_psi)
            const bool inwa = (unsigned)(bia - i0a) <= 2u * WIN, inwb = (unsigned)(bib - i0b) <= 2u * WIN;
            ws[0].ctr = inwa ? i0a + WIN : bia; ws[0].xr = xn.a; ws[0].yr = yn.a; ws[0].mo2 = inwa && mout.a > 1e-8 ? mout.a : 0.0;
            ws[1].ctr = inwb ? i0b + WIN : bib; ws[1].xr = xn.b; ws[1].yr = yn.b; ws[1].mo2 = inwb && mout.b > 1e-8 ? mout.b : 0.0;
        }
    }
    acc = fma2(sc[SC_QCTE], best, acc);                                           // (:144)
    // accelerations (:160-161), their cost (:170-171) and the ALM term
    D2 av = inv_ts * (zv - vprev), aw = inv_ts * (zw - wprev);
    acc = fma2(sc[SC_PA] * av, av, acc);
    acc = fma2(sc[SC_PW] * aw, aw, acc);
    const D2 tv = fma2(cbar_inv, yv, av), tw = fma2(cbar_inv, yw, aw);
    D2 sv = tv - clamp2(tv, a.pb.amin, a.pb.amax);
    D2 sw = tw - clamp2(tw, -a.pb.awmax, a.pb.awmax);
    acc = fma2(half_c, fma2(sv, sv, sw * sw), acc);
    if (sa_ == N - 1) {                                                           // terminal (:148)
        const double tx = xn.a - xf, ty = yn.a - yf, tth = thn.a - thf;
        acc.a = fma(sc[SC_QN], fma(tx, tx, ty * ty), acc.a);
        acc.a = fma(sc[SC_QTHN] * tth, tth, acc.a);
    }
    if (sb_ == N - 1) {
        const double tx = xn.b - xf, ty = yn.b - yf, tth = thn.b - thf;
        acc.b = fma(sc[SC_QN], fma(tx, tx, ty * ty), acc.b);
        acc.b = fma(sc[SC_QTHN] * tth, tth, acc.b);
    }
    if (!ina) { acc.a = 0.0; av.a = aw.a = sv.a = sw.a = 0.0; }
    if (!inb) { acc.b = 0.0; av.b = aw.b = sv.b = sw.b = 0.0; }
    av_out = av;
    aw_out = aw;
    const double fsum = gsum2(acc, lane);

    // obstacle penalties on the post-update state (:106-119); an obstacle no stage of any query point of this wave is
    // inside of contributes exactly 0 and is skipped (wave-uniform branch); the adjoint terms of a touched one are added
    // right where its F2_k has just been summed (same operations in the same order as eval_psi)
    double pen = 0.0;
    unsigned long long act = 0ull;
    unsigned act_dyn = 0u;
    bool scan = true;
    if (oc) {       // the activity scan is skipped while no stage has moved as far as its clearance from the untouched obstacles (eval_psi)
        const D2 ox = xn - oc->xo, oy = yn - oc->yo;
        const D2 o2 = fma2(ox, ox, oy * oy);
        if (!__any((ra & !(o2.a < oc->m2.a)) | (rb & !(o2.b < oc->m2.b)))) {
            act = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(oc->act_hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(oc->act_lo);
            act_dyn = (unsigned)__builtin_amdgcn_readfirstlane(oc->act_dyn);
            scan = false;
        }
#ifdef NMPC_WIN_STATS
        if (lane == 0) { atomicAdd(&nmpc_win_stats[2], 1ull); if (scan) atomicAdd(&nmpc_win_stats[3], 1ull); }
#endif
    }
    if (scan) {
        D2 mg = d2s(__builtin_inf());
        const lds_double *ob = L + mp.obs;
        const int nobs4 = (nobs + 3) & ~3;
#pragma unroll SH::NOBS >= 0 && SH::NOBS <= 16 ? 16 : 1
        for (int k = 0; k < nobs4; k += 4, ob += 4 * OBS_STRIDE) {
            double od[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                od[4 * j] = ob[OBS_STRIDE * j]; od[4 * j + 1] = ob[OBS_STRIDE * j + 1]; od[4 * j + 2] = ob[OBS_STRIDE * j + 2];
                od[4 * j + 3] = oc ? ob[OBS_STRIDE * j + 3] : 0.0;
            }
            NMPC_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (SH::NOBS >= 0 && SH::NOBS <= 16 && k + j >= SH::NOBS) continue;      // (unrolled: the padding slots of a fixed shape cost nothing)
                const D2 dx = xn - d2s(od[4 * j]), dy = yn - d2s(od[4 * j + 1]);
                const D2 h = fma2(-dy, dy, fma2(-dx, dx, d2s(od[4 * j + 2])));    // (:112)
                if (__any((ra & (h.a > 0.0)) | (rb & (h.b > 0.0)))) act |= 1ull << (k + j);
                else if (oc) {
                    mg.a = fmin(mg.a, __builtin_amdgcn_sqrt(od[4 * j + 2] - h.a) - od[4 * j + 3]);
                    mg.b = fmin(mg.b, __builtin_amdgcn_sqrt(od[4 * j + 2] - h.b) - od[4 * j + 3]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NDYN_MAX; ++k) {
            if (k < ndyn) {
                const D2 ca = dyn2(L, mp, te, k, DY_CA), sa = dyn2(L, mp, te, k, DY_SA);
                const D2 dx = xn - dyn2(L, mp, te, k, DY_EX), dy = yn - dyn2(L, mp, te, k, DY_EY);
                const D2 ea = fma2(dx, ca, dy * sa);
                const D2 eb = fma2(dx, sa, -(dy * ca));
                const D2 irx2 = dyn2(L, mp, te, k, DY_IRX2), iry2 = dyn2(L, mp, te, k, DY_IRY2);
                const D2 h = fma2(-(eb * eb), iry2, fma2(-(ea * ea), irx2, d2s(1.0)));   // (:118)
                if (__any((ra & (h.a > 0.0)) | (rb & (h.b > 0.0)))) act_dyn |= 1u << k;
                else if (oc) {      // the ellipse lies inside the disc of its larger half axis
                    mg.a = fmin(mg.a, __builtin_amdgcn_sqrt(fma(dx.a, dx.a, dy.a * dy.a)) - __builtin_amdgcn_rsq(fmin(irx2.a, iry2.a)));
                    mg.b = fmin(mg.b, __builtin_amdgcn_sqrt(fma(dx.b, dx.b, dy.b * dy.b)) - __builtin_amdgcn_rsq(fmin(irx2.b, iry2.b)));
                }
            }
        }
        if (oc) {
            const D2 m = fma2(0.99, mg, -1e-6);
            oc->xo = xn; oc->yo = yn;
            oc->m2 = D2{m.a > 0.0 ? (m.a < 1e100 ? m.a * m.a : 1e200) : 0.0, m.b > 0.0 ? (m.b < 1e100 ? m.b * m.b : 1e200) : 0.0};
            oc->act_lo = opaque_i((int)(unsigned)act); oc->act_hi = opaque_i((int)(unsigned)(act >> 32)); oc->act_dyn = opaque_i((int)act_dyn);
        }
    }
    // ---- adjoint, first term: the cross-track error through the arg-min segment of each stage ----
    D2 gx = d2s(0.0), gy = d2s(0.0);
    if (want_grad) {
        const double two_q = 2.0 * sc[SC_QCTE];
#define NMPC_CTE_ADJ(S_, BI_)                                                              \
        do {                                                                               \
            const lds_double *sg = L + mp.seg + SEG_STRIDE * (BI_);                        \
            const double px = xn.S_ - sg[0], py = yn.S_ - sg[1];                           \
            const double dot = fma(px, sg[2], py * sg[3]);                                 \
            const double that = dot * sg[4];                                               \
            const double tst = fmin(fmax(that, 0.0), 1.0);                                 \
            const double ex = fma(tst, sg[2], -px), ey = fma(tst, sg[3], -py);             \
            const double ed = fma(ex, sg[2], ey * sg[3]);                                  \
            const double m = (that > 0.0 && that < 1.0) ? ed * sg[4] : 0.0;                \
            gx.S_ = two_q * fma(m, sg[2], -ex);                                            \
            gy.S_ = two_q * fma(m, sg[3], -ey);                                            \
        } while (0)
        NMPC_CTE_ADJ(a, bia);
        NMPC_CTE_ADJ(b, bib);
#undef NMPC_CTE_ADJ
    }
    // ---- touched obstacles: F2_k, its square into the penalty, its adjoint terms ----
    if ((act | act_dyn) != 0ull) {
        for (unsigned long long rem = act; rem;) {
            const int k0 = __builtin_ctzll(rem);
            rem &= rem - 1;
            const lds_double *o0 = L + mp.obs + OBS_STRIDE * k0;
            const double ax = o0[0], ay = o0[1], ar = o0[2];
            const D2 dx0 = xn - d2s(ax), dy0 = yn - d2s(ay);
            const D2 h0 = fma2(-dy0, dy0, fma2(-dx0, dx0, d2s(ar)));
            const double f20 = gsum2(D2{ina ? fmax(h0.a, 0.0) : 0.0, inb ? fmax(h0.b, 0.0) : 0.0}, lane);
            if (WRITE_F2 && te == 0) L[f2off + k0] = f20;
            pen = fma(f20, f20, pen);
            if (want_grad) {
                const double w0 = -2.0 * (c * f20);
                if (h0.a > 0.0) { gx.a = fma(w0, dx0.a, gx.a); gy.a = fma(w0, dy0.a, gy.a); }
                if (h0.b > 0.0) { gx.b = fma(w0, dx0.b, gx.b); gy.b = fma(w0, dy0.b, gy.b); }
            }
        }
#pragma unroll
        for (int k = 0; k < NDYN_MAX; ++k) {
            if (act_dyn & (1u << k)) {
                const D2 ca = dyn2(L, mp, te, k, DY_CA), sa = dyn2(L, mp, te, k, DY_SA);
                const D2 irx2 = dyn2(L, mp, te, k, DY_IRX2), iry2 = dyn2(L, mp, te, k, DY_IRY2);
                const D2 dx = xn - dyn2(L, mp, te, k, DY_EX), dy = yn - dyn2(L, mp, te, k, DY_EY);
                const D2 ea = fma2(dx, ca, dy * sa);
                const D2 eb = fma2(dx, sa, -(dy * ca));
                const D2 h = fma2(-(eb * eb), iry2, fma2(-(ea * ea), irx2, d2s(1.0)));      // (:118)
                const double f2 = gsum2(D2{ina ? fmax(h.a, 0.0) : 0.0, inb ? fmax(h.b, 0.0) : 0.0}, lane);
                if (WRITE_F2 && te == 0) L[f2off + nobs + k] = f2;
                pen = fma(f2, f2, pen);
                if (want_grad) {
                    const double wk = -2.0 * (c * f2);
                    const D2 A = ea * irx2, Bq = eb * iry2;
                    const D2 hx = fma2(A, ca, Bq * sa);
                    const D2 hy = fma2(A, sa, -(Bq * ca));
                    if (h.a > 0.0) { gx.a = fma(wk, hx.a, gx.a); gy.a = fma(wk, hy.a, gy.a); }
                    if (h.b > 0.0) { gx.b = fma(wk, hx.b, gx.b); gy.b = fma(wk, hy.b, gy.b); }
                }
            }
        }
    }
    psi = fma(half_c, pen, fsum);
    pen_out = pen;
    if (!want_grad) return;

    // ---- adjoint sweep, continued ----
    const D2 wq = D2{sa_ < N - 1 ? sc[SC_Q] : sc[SC_QN], sb_ < N - 1 ? sc[SC_Q] : sc[SC_QN]};
    const D2 wth = D2{sa_ < N - 1 ? sc[SC_QTH] : sc[SC_QTHN], sb_ < N - 1 ? sc[SC_QTH] : sc[SC_QTHN]};
    gx = fma2(2.0 * wq, xn - d2s(xf), gx);
    gy = fma2(2.0 * wq, yn - d2s(yf), gy);
    D2 gt = (2.0 * wth) * (thn - d2s(thf));
    D2 qa = fma2(c, sv, (2.0 * sc[SC_PA]) * av);
    D2 qw = fma2(c, sw, (2.0 * sc[SC_PW]) * aw);
    if (!ina) { gx.a = gy.a = gt.a = qa.a = qw.a = 0.0; }
    if (!inb) { gx.b = gy.b = gt.b = qa.b = qw.b = 0.0; }
    const D2 Sx = gsuffix2(gx, lane);
    const D2 Sy = gsuffix2(gy, lane);
    const D2 e = fma2(Sy, cs, -(Sx * sn));
    D2 Dt = (ts * zv) * e;
    if (!ina) Dt.a = 0.0;
    if (!inb) Dt.b = 0.0;
    D2 stin = gt + next2(Dt, lane);
    if (!ina) stin.a = 0.0;
    if (!inb) stin.b = 0.0;
    const D2 St = gsuffix2(stin, lane);
    const D2 qan = next2(qa, lane), qwn = next2(qw, lane);
    const D2 dynv = fma2(Sx, cs, Sy * sn);
    D2 g1 = fma2(2.0 * sc[SC_RV], zv, (2.0 * sc[SC_QV]) * dv);
    g1 = fma2(inv_ts, qa - qan, g1);
    g1 = fma2(ts, dynv, g1);
    D2 g2 = (2.0 * sc[SC_RW]) * zw;
    g2 = fma2(inv_ts, qw - qwn, g2);
    g2 = fma2(ts, St, g2);
    gv = D2{ina ? g1.a : 0.0, inb ? g1.b : 0.0};
    gw = D2{ina ? g2.a : 0.0, inb ? g2.b : 0.0};
}

// ---------------------------------------------------------------------------------------------
// cost-layer kernel for 20 < N_hor <= 40: three instances per wave (one per lane group), the pair-form arithmetic
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void nmpc_eval2_kernel(KArgs a)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, g = lay_group<20>(lane), te = lay_stage<20>(lane);
    const LdsMap2 mp = lds_layout2(a.pb.N, a.pb.nobs, a.pb.ndyn);
    const int N = a.pb.N;
    // one slice per group; the groups are set up one after the other (prepare_instance2 is a whole-wave routine)
    int b3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int inst = blockIdx.x * 3 + k;
        b3[k] = inst < a.B ? inst : a.B - 1;            // surplus groups redo the last instance, write nothing
        lds_double *Lk = (lds_double *)lds + k * mp.total;
        prepare_instance2<ShapeAny>(a, Lk, mp, a.p + (size_t)b3[k] * a.n_p, lane);
        for (int j = lane; j < a.n2; j += 64) Lk[mp.f2 + j] = 0.0;
    }
    NMPC_WAVE_SYNC();
    lds_double *L = (lds_double *)lds + g * mp.total;
    const int inst = blockIdx.x * 3 + g;
    const int b = b3[0] * (g == 0) + b3[1] * (g == 1) + b3[2] * (g == 2);
    const bool ra = 2 * te < N, rb = 2 * te + 1 < N;
    const double *u = a.u + (size_t)b * a.n_u;
    const D2 zv = D2{ra ? u[4 * te] : 0.0, rb ? u[4 * te + 2] : 0.0}, zw = D2{ra ? u[4 * te + 1] : 0.0, rb ? u[4 * te + 3] : 0.0};
    const double c = a.ev_c ? a.ev_c[b] : 0.0;
    const double *yb = a.ev_y ? a.ev_y + (size_t)b * a.n1 : nullptr;
    const D2 yv = D2{(yb && ra) ? yb[2 * te] : 0.0, (yb && rb) ? yb[2 * te + 1] : 0.0};
    const D2 yw = D2{(yb && ra) ? yb[N + 2 * te] : 0.0, (yb && rb) ? yb[N + 2 * te + 1] : 0.0};
    double psi, pen;
    D2 gv, gw, av, aw;
    eval_psi2<ShapeAny, true>(a, L, mp, mp.f2, lane, te, zv, zw, c, 1.0 / fmax(c, 1.0), yv, yw, true, psi, pen, gv, gw, av, aw);
    NMPC_WAVE_SYNC();
    if (inst >= a.B) return;
    if (te == 0 && a.ev_psi) a.ev_psi[b] = psi;
    if (a.ev_grad) {
        double *go = a.ev_grad + (size_t)b * a.n_u;
        if (ra) { go[4 * te] = gv.a; go[4 * te + 1] = gw.a; }
        if (rb) { go[4 * te + 2] = gv.b; go[4 * te + 3] = gw.b; }
    }
    if (a.ev_F1) {
        double *fo = a.ev_F1 + (size_t)b * a.n1;
        if (ra) { fo[2 * te] = av.a; fo[N + 2 * te] = aw.a; }
        if (rb) { fo[2 * te + 1] = av.b; fo[N + 2 * te + 1] = aw.b; }
    }
    if (a.ev_F2 && te < 20) for (int k = te; k < a.n2; k += 20) a.ev_F2[(size_t)b * a.n2 + k] = L[mp.f2 + k];
}

// ---------------------------------------------------------------------------------------------
// the solver: nmpc_solve_hyb.h's state machine on two-stage vectors
// ---------------------------------------------------------------------------------------------
// the pipelined recurrence steps of nmpc_solve_hyb.h for two stages per lane (gram_fwd_step / gram_bwd_step there have the account)
template <int J>
__device__ __forceinline__ double gram2_fwd_step(double &ga1, double &ga2, double &dva, double &dvb, double &dwa, double &dwb, double rho, double gs,
                                                 double gy, double alp, double y1a, double y1b, double y2a, double y2b)
{
    double al;
    asm("v_mul_f64 %0, %7, %1\n\t"
        "v_fmac_f64_dpp %3, -%10, %11 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, -%10, %12 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, -%10, %13 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, -%10, %14 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%0, %8 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, -%0, %9 row_newbcast:%15 row_mask:0xf bank_mask:0xf"
        : "=&v"(al), "+v"(ga1), "+v"(ga2), "+v"(dva), "+v"(dvb), "+v"(dwa), "+v"(dwb)
        : "v"(rho), "v"(gs), "v"(gy), "v"(alp), "v"(y1a), "v"(y1b), "v"(y2a), "v"(y2b), "n"(J), "n"(J - 1));
    return al;
}
template <int J>
__device__ __forceinline__ double gram2_bwd_step(double &ga2, double &dva, double &dvb, double &dwa, double &dwb, double rho, double alv, double gr,
                                                 double abp, double s1a, double s1b, double s2a, double s2b)
{
    double ab;
    asm("v_mul_f64 %0, %6, %1\n\t"
        "v_add_f64 %0, %7, -%0\n\t"
        "v_fmac_f64_dpp %2, %9, %10 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %9, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %9, %12 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %9, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %0, %8 row_newbcast:%14 row_mask:0xf bank_mask:0xf"
        : "=&v"(ab), "+v"(ga2), "+v"(dva), "+v"(dvb), "+v"(dwa), "+v"(dwb)
        : "v"(rho), "v"(alv), "v"(gr), "v"(abp), "v"(s1a), "v"(s1b), "v"(s2a), "v"(s2b), "n"(J), "n"(J + 1));
    return ab;
}
template <int J>
__device__ __forceinline__ void fnma4_row_bcast(double &a1, double &a2, double &a3, double &a4, double x, double y1, double y2, double y3, double y4,
                                                double after)
{
    asm("v_fmac_f64_dpp %0, -%4, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%4, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, -%4, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, -%4, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
        : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(x), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(after), "n"(J));
}

// forward-backward envelope at the point whose cost / gradient step / half step / gradient are given (state layout)
__device__ __forceinline__ double fbe_value2(double cost, double gamma, double hig, D2 sv, D2 sw, D2 hv, D2 hw, D2 gv, D2 gw, int lane)
{
    const D2 e1 = sv - hv, e2 = sw - hw;
    double dist2, gg;
    pair_sum(hdot2(e1, e2, e1, e2), hdot2(gv, gw, gv, gw), lane, dist2, gg);
    return cost - (0.5 * gamma) * gg + dist2 * hig;      // hig = 0.5 / gamma, formed when gamma changes
}

#define NMPC2_HALF_STEP(xv, xw)                                                                    \
    do {                                                                                           \
        const D2 s1_ = fma2(-gamma, gv, (xv)), s2_ = fma2(-gamma, gw, (xw));                       \
        hv = D2{ina ? clampd(s1_.a, vmin, vmax) : s1_.a, inb ? clampd(s1_.b, vmin, vmax) : s1_.b}; \
        hw = D2{ina ? clampd(s2_.a, -wmax, wmax) : s2_.a, inb ? clampd(s2_.b, -wmax, wmax) : s2_.b}; \
    } while (0)
#define NMPC2_FBE(xv, xw) fbe_value2(cost, gamma, Lpar[19], fma2(-gamma, gv, (xv)), fma2(-gamma, gw, (xw)), hv, hw, gv, gw, lane)
// gradient pair of query point K's evaluation, from the LDS area the evaluation lanes have filled (zero beyond the horizon)
#define NMPC2_LOAD_GRAD(BASE, OV, OW)                                          \
    do {                                                                       \
        D2 fv_, fw_;                                                           \
        ld4<H2_ENT>((BASE), tz, fv_, fw_);                                        \
        OV = D2{ina ? fv_.a : 0.0, inb ? fv_.b : 0.0};                         \
        OW = D2{ina ? fw_.a : 0.0, inb ? fw_.b : 0.0};                         \
    } while (0)

template <class SH>
__global__ __launch_bounds__(64 * TEAM_WAVES, 1) void nmpc_solve_hyb2_kernel(KArgs a)
{
    constexpr int P = 32;                       // state layout: stage pair t at lane t of both 32-lane halves
    extern __shared__ double lds[];
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LdsMap2 mp = the_map2<SH>(a);
    const int slice = mp.total;
    lds_double *L = (lds_double *)lds + wid * slice;
    lds_int *ctl = (lds_int *)((lds_double *)lds + TEAM_WAVES * slice);
    const int lane = threadIdx.x & 63, h = lane >> 5, t = lane & 31;
    const int q = lay_group<20>(lane), te = lay_stage<20>(lane);
    const int N = shape_N<SH>(a), m = a.op.lbfgs_memory;
    const int NP = (N + 1) >> 1;                // lane pairs in use
    const bool in = t < NP, ina = in, inb = 2 * t + 1 < N;          // state layout: this lane's stages inside the horizon
    constexpr bool FULL = SH::N == 40;
    const bool rea = 2 * te < N, reb = 2 * te + 1 < N;              // evaluation layout: real stages
    const bool inea = FULL ? true : rea, ineb = FULL ? true : reb;
    const int n2 = shape_nobs<SH>(a) + shape_ndyn<SH>(a);
    const int f2off = mp.f2 + q * (n2 + 1);
    const int tz = in ? t : 0, tt = t < H2_NS - 1 ? t : H2_NS - 1;  // LDS entry of this state lane (ring: the zero column beyond its 20 entries; entries beyond the horizon are zeros)
    const int tc = t < H2_COLS ? t : H2_COLS - 1;                   // ... in the parked columns (lanes 24..31 share the last one: zeros)
    lds_double2 *Lnv = (lds_double2 *)(L + mp.nv);                  // Gram-form L-BFGS (nmpc_solve_hyb.h): s | y | r | g of the iteration ...
    lds_double *Lgsy = L + mp.gsy, *Lgyy = L + mp.gyy;              // ... the kept inner products
    const int c16 = lane & 15, q4 = lane >> 4;                      // ... batch lane = (ring pair / age, quarter of the horizon)
#define NMPC2_LB_ZERO()                                                                                    \
    do {                                                                                                   \
        lds_double2 *z_ = (lds_double2 *)(L + mp.gsy);                                                     \
        for (int i_ = lane; i_ < MAXMEM * GRAM_LD + 4 * MAXMEM * H2_NS; i_ += 64) z_[i_] = dbl2{0.0, 0.0};  \
        if (lane < MAXMEM) Lrho[lane] = 0.0;                                                               \
    } while (0)
    lds_double2 *LS = (lds_double2 *)(L + mp.S);
    lds_double2 *LY = (lds_double2 *)(L + mp.Y);
    lds_double *Lrho = L + mp.rho;
    lds_double2 *V = (lds_double2 *)(L + mp.vec);
    // parked columns (bases; the entry is t in the state layout, te in the evaluation layout)
    lds_double2 *Cos = V + 2 * H2_COLS * 0, *Cog = V + 2 * H2_COLS * 1, *Cq = V + 2 * H2_COLS * 2, *Cyp = V + 2 * H2_COLS * 3, *Cy = V + 2 * H2_COLS * 4,
                *Cgk = V + 2 * H2_COLS * 6;
    lds_double2 *Pts = (lds_double2 *)(L + mp.pts), *Grd = (lds_double2 *)(L + mp.grd), *Lreq = (lds_double2 *)(L + mp.req);
    if (threadIdx.x < TEAM_CTL_INTS) ctl[threadIdx.x] = threadIdx.x == CTL_OWNERS ? a.team_owners : 0;
    __syncthreads();
    unsigned team_seq = 0;
    const double vmin = a.pb.vmin, vmax = a.pb.vmax, wmax = a.pb.wmax;
    const unsigned max_inner = (unsigned)a.op.max_inner;
    const unsigned budget = (unsigned)a.op.max_total_inner;
    lds_double *Lpar = L + mp.par;

    for (; wid < a.team_owners;) {
        // next instance: a fresh one from the queue; when that is empty, one that stepped aside at an outer-iteration boundary (long ones first)
        int fetched = -1, from_pool = 0;
        if (lane == 0) {
            const unsigned nxt = atomicAdd(a.queue, 1u);
            if (nxt < (unsigned)a.B) fetched = a.order ? a.order[nxt] : (int)nxt;
            else if (a.sched_mode > 0) {
                for (int c = 0; c < NPOOLS && fetched < 0; ++c) fetched = pool_pop(a, c);
                from_pool = fetched >= 0;
            }
        }
        const int inst = __builtin_amdgcn_readfirstlane(fetched);
        const bool resumed = __builtin_amdgcn_readfirstlane(from_pool) != 0;
        if (inst < 0) break;
        long long t_start = (long long)__builtin_amdgcn_s_memrealtime();
        if (lane == 0) ctl_store(ctl + CTL_INST + wid, -1);  // (nmpc_solve_hyb.h: the id is away while the tables change)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        prepare_instance2<SH>(a, L, mp, a.p + (size_t)inst * a.n_p, lane);
        WinState ws[2] = {{2 * te < N - 1 ? 2 * te : N - 2, 0.0, 0.0, 0.0}, {2 * te + 1 < N - 1 ? 2 * te + 1 : N - 2, 0.0, 0.0, 0.0}};      // this lane's cross-track windows
        ObsCert2 oc = {d2s(0.0), d2s(0.0), d2s(0.0), 0, 0, 0};      // ... and its obstacle certificate
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) ctl_store(ctl + CTL_INST + wid, inst); // (helpers tell by it whether their windows are still this instance's)

        // a fresh instance starts from the caller's u0 / y0, a resumed one from its parked state (acquired by pool_pop): u | y | previous gradient
        // in the layout of the caller's arrays, then 16 scalars (park_stride)
        const double *pk = a.park + (size_t)inst * park_stride(N);
        const double *u0 = resumed ? pk : a.u + (size_t)inst * a.n_u;
        D2 uv = D2{ina ? u0[4 * t] : 0.0, inb ? u0[4 * t + 2] : 0.0}, uw = D2{ina ? u0[4 * t + 1] : 0.0, inb ? u0[4 * t + 3] : 0.0};
        {
            const double *yb = resumed ? pk + 2 * N : (a.y0 ? a.y0 + (size_t)inst * a.n1 : nullptr);
            const D2 yv0 = D2{(yb && rea) ? yb[2 * te] : 0.0, (yb && reb) ? yb[2 * te + 1] : 0.0};
            const D2 yw0 = D2{(yb && rea) ? yb[N + 2 * te] : 0.0, (yb && reb) ? yb[N + 2 * te + 1] : 0.0};
            st4<H2_COLS>(Cyp, te, yv0, yw0);
            st4<H2_COLS>(Cy, te, yv0, yw0);
        }
        {   // gradient_u_previous (AKKT residual): zero at the start of a solve, carried across its inner solves
            const double *qb = pk + 4 * N;
            const bool ld = resumed && ina;
            st4<H2_COLS>(Cq, tc, D2{ld ? qb[4 * t] : 0.0, (ld && inb) ? qb[4 * t + 2] : 0.0}, D2{ld ? qb[4 * t + 1] : 0.0, (ld && inb) ? qb[4 * t + 3] : 0.0});
        }
        D2 gv = d2s(0.0), gw = d2s(0.0), hv = d2s(0.0), hw = d2s(0.0), rv = d2s(0.0), rw = d2s(0.0), dv = d2s(0.0), dw = d2s(0.0);
        D2 pv = d2s(0.0), pw = d2s(0.0);          // line-search trial point being consumed
        D2 xv = d2s(0.0), xw = d2s(0.0);          // query point X of THIS half (-> evaluation points 0 and 1)
        D2 yqv = d2s(0.0), yqw = d2s(0.0);        // query point Y (-> evaluation point 2)
        unsigned fl = 0u;                         // the state machine's flags: bits of one scalar (FlagBit, nmpc_solve_hyb.h)
        FlagBit need_grad{fl, 1u << 0}; need_grad = true;
        double cost = 0, gamma = 0, nr2 = 0, norm_r = 0, tau = 1, rhs_ls = 0;
        double Lc = 0, sigma = 0, H0 = 1, c_lip = 0, gr = 0, norm_h = 0;
        double eps_nu = a.op.initial_tolerance, dy_norm = 0, f2_norm = 0, dy_norm_plus = DBL_MAX, f2_norm_plus = 0, last_fpr = 0, last_cost = 0;
        double fbe_u = 0;
        FlagBit fbe_ok{fl, 1u << 1};
        int iteration = 0, lip_it = 0, ls_n = 0, lb_active = 0, lb_head = 0;
        FlagBit lb_first{fl, 1u << 2}; lb_first = true;
        int n_active = 0, n_head = 0;
        FlagBit n_first{fl, 1u << 3}, n_take_old{fl, 1u << 4}; n_first = true;
        double n_H0 = 1;
        unsigned num_iter = 0;
        const double c0 = a.c0 ? a.c0[inst] : 0.0;
        double pen_c = c0 > 0.0 ? c0 : a.op.initial_penalty;
        double cbar_inv = 1.0 / fmax(pen_c, 1.0);
        int nu = 0, inner_status = 0, state = D_INIT, final_status = 0;
        unsigned inner_total = 0, n_cost = 0, n_grad = 0, n_pass = 0;
        FlagBit f_start{fl, 1u << 7}, f_back{fl, 1u << 8}, f_trials{fl, 1u << 9}, f_end{fl, 1u << 10}, f_begin{fl, 1u << 11}, f_done{fl, 1u << 12}, f_fb{fl, 1u << 13};
        FlagBit running{fl, 1u << 14}, timed_out{fl, 1u << 15}, posted{fl, 1u << 16};
        f_start = true; running = true;

        unsigned q_pass = 0;                            // n_pass at the last outer-iteration boundary (or at the start of this leg)
        FlagBit parked{fl, 1u << 5}, long_counted{fl, 1u << 6};      // (nmpc_solve_hyb.h: stepping aside at outer-iteration boundaries)
        int park_cls = POOL_LONG;
        if (resumed) {
            const double *pks = pk + 6 * N;
            pen_c = pks[0]; cbar_inv = 1.0 / fmax(pen_c, 1.0);
            eps_nu = pks[1]; dy_norm = pks[2]; f2_norm = pks[3]; dy_norm_plus = pks[4]; f2_norm_plus = pks[5]; last_fpr = pks[6]; last_cost = pks[7];
            nu = (int)pks[8]; inner_total = (unsigned)pks[9]; n_cost = (unsigned)pks[10]; n_grad = (unsigned)pks[11]; n_pass = (unsigned)pks[12];
            t_start = (long long)pks[13]; long_counted = pks[15] != 0.0; q_pass = n_pass;
        }
#if defined(NMPC_MARKS)     // section markers in the ISA dump (hipcc -S -DNMPC_MARKS; scripts/isa_stats.py)
#define NMPC2_TK(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " #i); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NMPC2_TK(i) do { } while (0)
#endif

        for (;;) {
            bool lb_batch = false;                     // this pass starts with the batch of inner products (f_back, f_begin)
            // (the two rare phases behind ONE test: nmpc_solve_hyb.h)
            if (fl & (f_back.bit | f_fb.bit)) {
                // ---------------------------------------------------------------- backtrack: L <- 2L, gamma <- gamma/2
                if (f_back) {
                    if (posted) { posted = false; if (lane == 0) ctl_store(ctl + CTL_CLAIM + wid, 0); }
                    lb_active = 0; lb_first = true;
                    NMPC2_LB_ZERO();
                    fbe_ok = false;
                    Lc *= 2.0; gamma /= 2.0;
                    sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * gamma);
                    c_lip = GAMMA_L_COEFF / (2.0 * gamma);
                    Lpar[19] = 0.5 / gamma;      // (the envelope's factor; helpers read it with the request)
                    NMPC2_HALF_STEP(uv, uw);
                    lb_batch = true;
                }
                // ---------------------------------------------------------------- every trial failed (opts.ls_failure = 1)
                if (f_fb) {
                    f_fb = false;
                    tau = 0.0;
                    ld4<H2_COLS>(Cgk, tc, gv, gw);
                    NMPC2_HALF_STEP(uv, uw);
                    xv = yqv = hv; xw = yqw = hw; need_grad = true; state = D_FB;
                }
            }
            // ---------------------------------------------------------------- line-search trials (tau, ls_n) | (tau/2, ls_n+1) | (tau/4, ls_n+2)
            if (f_trials) {
                f_trials = false;
                const double th_ = h == 0 ? tau : tau / 2.0, omt = 1.0 - th_;
                xv = fma2(-th_, dv, fma2(-omt, rv, uv));
                xw = fma2(-th_, dw, fma2(-omt, rw, uw));
                const double t4_ = tau / 4.0, om4_ = 1.0 - t4_;
                yqv = fma2(-t4_, dv, fma2(-om4_, rv, uv));
                yqw = fma2(-t4_, dw, fma2(-om4_, rw, uw));
                need_grad = true; state = D_LS;
            }
            // ---------------------------------------------------------------- an iteration finished
            // (Measured, round 6: handled where it is raised, as in nmpc_solve_hyb.h, this kernel is 3 % SLOWER -- 125.1 against 121.1 ms on
            // cfg 2, with 226 spilled scalars instead of 197 --, although each of the round's three state-machine changes alone is worth 0.3 .. 1.3 %.)
            if (f_end) {
                f_end = false;
                iteration++;
                if (!(num_iter < max_inner)) f_done = true;
                else {
                    num_iter++;
                    if (budget > 0u && inner_total + num_iter >= budget) { timed_out = true; f_done = true; }
                    else f_begin = true;
                }
            }
            // ---------------------------------------------------------------- start of a PANOC step
            if (f_begin) lb_batch = true;
            // ---- the batch of inner products of this step (Gram-form L-BFGS: nmpc_solve_hyb.h has the design; here a quarter is ten stages --
            // five entries of two planes -- and the oracle's qdot runs over 40 stages)
            double gU = 0.0;
            D2 gs1 = d2s(0.0), gs2 = d2s(0.0), gy1 = d2s(0.0), gy2 = d2s(0.0);
            if (lb_batch) {
                rv = uv - hv; rw = uw - hw;                // the residual of the step (a back-off has just halved gamma and renewed the half step)
                if (f_begin && iteration >= 1 && !lb_first) {
                    D2 o1, o2, g1, g2;
                    ld4<H2_COLS>(Cos, tc, o1, o2);
                    ld4<H2_COLS>(Cog, tc, g1, g2);
                    gs1 = uv - o1; gs2 = uw - o2; gy1 = rv - g1; gy2 = rw - g2;
                }
                if (t < H2_NS - 1 && h == 0) {
                    st4<H2_NS>(Lnv, t, gs1, gs2); st4<H2_NS>(Lnv + 2 * H2_NS, t, gy1, gy2);
                    st4<H2_NS>(Lnv + 4 * H2_NS, t, rv, rw); st4<H2_NS>(Lnv + 6 * H2_NS, t, gv, gw);
                }
                const lds_double2 *X1 = (c16 < MAXMEM ? LS + 2 * H2_NS * c16 : (c16 == 11 ? Lnv + 4 * H2_NS : Lnv)) + 5 * q4;
                const lds_double2 *X2 = (c16 < MAXMEM ? LY + 2 * H2_NS * c16 : (c16 == 11 ? Lnv + 6 * H2_NS : Lnv + 2 * H2_NS)) + 5 * q4;
                const lds_double2 *Z1 = (c16 < MAXMEM ? Lnv + 2 * H2_NS : (c16 == 10 ? Lnv : Lnv + 4 * H2_NS)) + 5 * q4;
                const lds_double2 *Z2 = (c16 < MAXMEM ? Lnv + 4 * H2_NS : (c16 == 10 ? Lnv + 2 * H2_NS : Lnv + 4 * H2_NS)) + 5 * q4;
                double V1 = 0.0, V2 = 0.0, V3 = 0.0, V4 = 0.0;
#pragma unroll
                for (int e = 0; e < 5; ++e) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {          // plane 0: the entry's first stage, plane 1: its second
                        const dbl2 x1 = X1[pl * H2_NS + e], x2 = X2[pl * H2_NS + e], z1 = Z1[pl * H2_NS + e], z2 = Z2[pl * H2_NS + e];
                        V1 = fma(x1.x, z1.x, V1); V1 = fma(x1.y, z1.y, V1);
                        V2 = fma(x1.x, z2.x, V2); V2 = fma(x1.y, z2.y, V2);
                        V3 = fma(x2.x, z1.x, V3); V3 = fma(x2.y, z1.y, V3);
                        V4 = fma(x2.x, z2.x, V4); V4 = fma(x2.y, z2.y, V4);
                    }
                }
                swap_rows(V1, V2);
                double W12 = V1 + V2;
                swap_rows(V3, V4);
                double W34 = V3 + V4;
                swap_halves(W12, W34);
                gU = W12 + W34;
                nr2 = lane_scalar(gU, 11);                       // <r, r>
                gr = lane_scalar(gU, 32 + 11);                   // <g, r>
                norm_r = sqrt(nr2);
            }
            if (f_begin) {
                f_begin = false;
                bool exit_now = false;
                if (__any(norm_r < a.op.tolerance)) {
                    if (a.op.akkt_gradient == 2) exit_now = true;
                    else if (a.op.akkt_gradient == 1 && iteration >= 1) {
                        exit_now = __any(norm_r < eps_nu * gamma);      // (nmpc_solve_hyb.h: the residual is r / gamma)
                    } else {
                        D2 q1, q2;
                        ld4<H2_COLS>(Cq, tc, q1, q2);
                        const bool top = a.op.akkt_gradient == 1;       // iteration 0: grad_prev is still the zero vector
                        const D2 b1 = top ? gv : gv - q1;
                        const D2 b2 = top ? gw : gw - q2;
                        const D2 c1 = D2{rv.a / gamma + b1.a, rv.b / gamma + b1.b}, c2 = D2{rw.a / gamma + b2.a, rw.b / gamma + b2.b};
                        exit_now = __any(sqrt(group_sum<P>(hdot2(c1, c2, c1, c2), lane)) < eps_nu);
                    }
                }
                if (exit_now) {
                    f_done = true;
                } else if (iteration == 0) {
                    lip_it = 0;
                    xv = yqv = hv; xw = yqw = hw; need_grad = true; state = D_LIP;
                } else {
                    lip_it = 0;
                    // ---- tentative L-BFGS update with (s, y) = (u - u_old, r - r_old) ----
                    n_first = lb_first; n_head = lb_head; n_active = lb_active; n_H0 = H0; n_take_old = false;
                    bool took = false;
                    if (lb_first) {
                        n_first = false; n_take_old = true;
                    } else {
                        const double ss = lane_scalar(gU, 10), ys = lane_scalar(gU, 16 + 10);
                        bool ok = !(ss <= DBL_MIN || ys <= LBFGS_SY_EPSILON);
                        if (ok) ok = ys > (LBFGS_CBFGS_EPSILON * norm_r) * ss;
                        if (__any(ok)) {
                            took = true;
                            n_take_old = true;
                            n_head = lb_head == 0 ? MAXMEM - 1 : lb_head - 1;      // (the ring always turns over its ten slots)
                            if (in && h == 0) { st4<H2_NS>(LS + 2 * H2_NS * (n_head), t, gs1, gs2); st4<H2_NS>(LY + 2 * H2_NS * (n_head), t, gy1, gy2); }
                            if (lane == 0) Lrho[n_head] = 1.0 / ys;
                            const double yy = lane_scalar(gU, 48 + 10);
                            n_H0 = ys / yy;
                            if (n_active < m) n_active++;
                            if (c16 < MAXMEM && q4 < 3) {
                                const bool dg = c16 == n_head;
                                lds_double *wa = q4 == 0 ? Lgsy + c16 * GRAM_LD + n_head : (q4 == 1 ? Lgsy + n_head * GRAM_LD + c16 : Lgyy + c16 * GRAM_LD + n_head);
                                const double wv = q4 == 1 ? 0.0 : (dg ? (q4 == 0 ? 0.0 : yy) : gU);
                                *wa = wv;
                                if (q4 == 2) Lgyy[n_head * GRAM_LD + c16] = wv;
                            }
                            if (m < MAXMEM) {      // a shorter memory: the pair that has just reached age m leaves, its slot goes back to zeros
                                const int ev = n_head + m >= MAXMEM ? n_head + m - MAXMEM : n_head + m;
                                if (t < H2_NS && h == 0) { st4<H2_NS>(LS + 2 * H2_NS * (ev), t, d2s(0.0), d2s(0.0)); st4<H2_NS>(LY + 2 * H2_NS * (ev), t, d2s(0.0), d2s(0.0)); }
                                if (lane == 0) Lrho[ev] = 0.0;
                                if (c16 < MAXMEM && q4 < 2) {
                                    (q4 == 0 ? Lgsy : Lgyy)[c16 * GRAM_LD + ev] = 0.0;
                                    (q4 == 0 ? Lgsy : Lgyy)[ev * GRAM_LD + c16] = 0.0;
                                }
                            }
                        }
                    }
                    // ---- d = H r over the tentative buffer: the two recurrences in lanes 0..9 of every row, the direction updated along ----
                    dv = rv; dw = rw;
                    if (n_active > 0) {
                        const int pk_ = c16 < MAXMEM ? (n_head + c16 >= MAXMEM ? n_head + c16 - MAXMEM : n_head + c16) : MAXMEM - 1;
                        const int pkrow = pk_ * GRAM_LD;
                        double ga1 = lane_get(gU, 16 + pk_), ga2 = lane_get(gU, 48 + pk_);      // <s_k, r>, <y_k, r>
                        if (took && c16 == 0) { ga1 = lane_scalar(gU, 12); ga2 = lane_scalar(gU, 32 + 12); }
                        const double rho_k = Lrho[pk_];
// (nmpc_solve_hyb.h: the twenty steps once per head position H of the ring -- every LDS address a per-lane base plus an immediate -- and
// software-pipelined: the direction updates of the step before fill the two wait states between a coefficient and its DPP read)
#define NMPC2_GRAM_LDF(H, J)                                                                       \
                            constexpr int fp##J = ((H) + (J)) % MAXMEM;                            \
                            const double fgs##J = Lgsy[pkrow + fp##J], fgy##J = Lgyy[pkrow + fp##J]; \
                            D2 fy1##J, fy2##J;                                                     \
                            ld4<H2_NS>(LY + 2 * H2_NS * (fp##J), tt, fy1##J, fy2##J)
#define NMPC2_GRAM_FWD(H, J, JP)                                                                   \
                            NMPC2_GRAM_LDF(H, J);                                                  \
                            const double fal##J = gram2_fwd_step<(J)>(ga1, ga2, dv.a, dv.b, dw.a, dw.b, rho_k, fgs##J, fgy##J, fal##JP, fy1##JP.a, fy1##JP.b, fy2##JP.a, fy2##JP.b)
#define NMPC2_GRAM_LDB(H, J)                                                                       \
                            constexpr int bp##J = ((H) + (J)) % MAXMEM;                            \
                            const double bgr##J = Lgsy[bp##J * GRAM_LD + pk_];                     \
                            D2 bs1##J, bs2##J;                                                     \
                            ld4<H2_NS>(LS + 2 * H2_NS * (bp##J), tt, bs1##J, bs2##J)
#define NMPC2_GRAM_BWD(H, J, JP)                                                                   \
                            NMPC2_GRAM_LDB(H, J);                                                  \
                            const double bab##J = gram2_bwd_step<(J)>(ga2, dv.a, dv.b, dw.a, dw.b, rho_k, alv, bgr##J, bab##JP, bs1##JP.a, bs1##JP.b, bs2##JP.a, bs2##JP.b)
#define NMPC2_GRAM_BOTH(H)                                                                                                                \
                        do {                                                                                                              \
                            NMPC2_GRAM_LDF(H, 0);                                                                                         \
                            const double fal0 = gram_fwd_first(ga1, ga2, rho_k, fgs0, fgy0);                                              \
                            NMPC2_GRAM_FWD(H, 1, 0); NMPC2_GRAM_FWD(H, 2, 1); NMPC2_GRAM_FWD(H, 3, 2); NMPC2_GRAM_FWD(H, 4, 3);          \
                            NMPC2_GRAM_FWD(H, 5, 4); NMPC2_GRAM_FWD(H, 6, 5); NMPC2_GRAM_FWD(H, 7, 6); NMPC2_GRAM_FWD(H, 8, 7);          \
                            NMPC2_GRAM_FWD(H, 9, 8);                                                                                      \
                            fnma4_row_bcast<9>(dv.a, dv.b, dw.a, dw.b, fal9, fy19.a, fy19.b, fy29.a, fy29.b, ga1);                        \
                            const double alv = rho_k * ga1;                                                                               \
                            ga2 = n_H0 * ga2;                                                                                             \
                            dv = n_H0 * dv; dw = n_H0 * dw;                                                                               \
                            NMPC2_GRAM_LDB(H, 9);                                                                                         \
                            const double bab9 = alv - rho_k * ga2;                                                                        \
                            ga2 = fma_row_bcast<9>(ga2, bab9, bgr9);                                                                      \
                            NMPC2_GRAM_BWD(H, 8, 9); NMPC2_GRAM_BWD(H, 7, 8); NMPC2_GRAM_BWD(H, 6, 7); NMPC2_GRAM_BWD(H, 5, 6);          \
                            NMPC2_GRAM_BWD(H, 4, 5); NMPC2_GRAM_BWD(H, 3, 4); NMPC2_GRAM_BWD(H, 2, 3); NMPC2_GRAM_BWD(H, 1, 2);          \
                            NMPC2_GRAM_BWD(H, 0, 1);                                                                                      \
                            fma4_row_bcast<0>(dv.a, dv.b, dw.a, dw.b, bab0, bs10.a, bs10.b, bs20.a, bs20.b, ga2);                         \
                        } while (0)
                        switch (n_head) {
                        case 0: NMPC2_GRAM_BOTH(0); break;
                        case 1: NMPC2_GRAM_BOTH(1); break;
                        case 2: NMPC2_GRAM_BOTH(2); break;
                        case 3: NMPC2_GRAM_BOTH(3); break;
                        case 4: NMPC2_GRAM_BOTH(4); break;
                        case 5: NMPC2_GRAM_BOTH(5); break;
                        case 6: NMPC2_GRAM_BOTH(6); break;
                        case 7: NMPC2_GRAM_BOTH(7); break;
                        case 8: NMPC2_GRAM_BOTH(8); break;
                        default: NMPC2_GRAM_BOTH(9); break;
                        }
#undef NMPC2_GRAM_FWD
#undef NMPC2_GRAM_LDF
#undef NMPC2_GRAM_LDB
#undef NMPC2_GRAM_BWD
#undef NMPC2_GRAM_BOTH
                    }
                    if (!fbe_ok) { fbe_u = NMPC2_FBE(uv, uw); fbe_ok = true; }
                    rhs_ls = fbe_u - sigma * nr2;
                    tau = 1.0; ls_n = 0;
                    xv = h ? fma2(-1.0, dv, fma2(-0.0, rv, uv)) : hv;      // X: u_bar | u+(tau = 1) = u - 0 r - d
                    xw = h ? fma2(-1.0, dw, fma2(-0.0, rw, uw)) : hw;
                    yqv = fma2(-0.5, dv, fma2(-0.5, rv, uv));               // Y: u+(tau = 1/2)
                    yqw = fma2(-0.5, dw, fma2(-0.5, rw, uw));
                    need_grad = true; state = D_ITER;
                    // team: idle waves of this workgroup evaluate the trials tau = 2^-2 .. 2^-10 of this direction meanwhile
                    if (a.team_help && __builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_HELPERS)) > 0) {
                        if (t < H2_ENT && h == 0) {
                            st4<H2_ENT>(Lreq, t, uv, uw);
                            st4<H2_ENT>(Lreq + 2 * H2_ENT * 1, t, rv, rw);
                            st4<H2_ENT>(Lreq + 2 * H2_ENT * (2), t, dv, dw);
                        }
                        if (lane == 0) { Lpar[15] = pen_c; Lpar[16] = cbar_inv; Lpar[17] = gamma; }
                        team_seq = team_seq >= 0xffff0u ? 1u : team_seq + 1u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) ctl_store(ctl + CTL_CLAIM + wid, (int)(team_seq << 8));
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
                    last_fpr = norm_r; last_cost = cost;
                    uv = hv; uw = hw;                                        // PANOC returns the feasible half step
                    const bool fin = __builtin_isfinite(uv.a) && __builtin_isfinite(uw.a) && __builtin_isfinite(uv.b) && __builtin_isfinite(uw.b) &&
                                     __builtin_isfinite(cost) && __builtin_isfinite(norm_r);
                    if (__any(in & !fin)) { final_status = NMPC_NOT_CONVERGED_NOT_FINITE; running = false; }
                    else { xv = yqv = uv; xw = yqw = uw; need_grad = false; state = D_ALM; }
                }
                // ---------------------------------------------------------------- start an inner solve
                if (f_start) {
                    f_start = false;
                    { D2 y1, y2; ld4<H2_COLS>(Cy, te, y1, y2); st4<H2_COLS>(Cy, te, clamp2(y1, -1e12, 1e12), clamp2(y2, -1e12, 1e12)); }      // y <- Pi_Y(y)
                    lb_active = 0; lb_first = true; iteration = 0; num_iter = 0; tau = 1.0;
                    NMPC2_LB_ZERO();
                    const D2 h1 = D2{EPSILON_LIPSCHITZ * uv.a > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uv.a : DELTA_LIPSCHITZ,
                                     EPSILON_LIPSCHITZ * uv.b > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uv.b : DELTA_LIPSCHITZ};
                    const D2 h2 = D2{EPSILON_LIPSCHITZ * uw.a > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uw.a : DELTA_LIPSCHITZ,
                                     EPSILON_LIPSCHITZ * uw.b > DELTA_LIPSCHITZ ? EPSILON_LIPSCHITZ * uw.b : DELTA_LIPSCHITZ};
                    norm_h = sqrt(group_sum<P>((ina ? fma(h1.a, h1.a, h2.a * h2.a) : 0.0) + (inb ? fma(h1.b, h1.b, h2.b * h2.b) : 0.0), lane));
                    xv = h == 1 ? D2{ina ? uv.a + h1.a : 0.0, inb ? uv.b + h1.b : 0.0} : uv;
                    xw = h == 1 ? D2{ina ? uw.a + h2.a : 0.0, inb ? uw.b + h2.b : 0.0} : uw;
                    yqv = uv; yqw = uw;
                    need_grad = true; state = D_INIT;
                }
            }
            if (!running) break;

            // ================================================================ one pass: psi at three points
            double psi, pen;
            D2 egv = d2s(0.0), egw = d2s(0.0), eav, eaw;
            n_pass++;
            NMPC2_TK(0);
            // query points: state layout -> LDS -> evaluation layout (X of half 0 | X of half 1 | Y)
            if (t < H2_ENT) {
                st4<H2_ENT>(Pts + 2 * H2_ENT * (h), t, xv, xw);
                if (h == 0) st4<H2_ENT>(Pts + 2 * H2_ENT * (2), t, yqv, yqw);
            }
            NMPC_WAVE_SYNC();
            D2 zv, zw, yv, yw;
            ld4<H2_ENT>(Pts + 2 * H2_ENT * (q), te, zv, zw);
            ld4<H2_COLS>(Cy, te, yv, yw);
            NMPC2_TK(1);
            eval_psi2<SH, false, NMPC_WIN2>(a, L, mp, f2off, lane, te, zv, zw, pen_c, cbar_inv, yv, yw, need_grad, psi, pen, egv, egw, eav, eaw, ws, &oc);
            NMPC2_TK(2);
            if (need_grad) st4<H2_ENT>(Grd + 2 * H2_ENT * (q), te, egv, egw);
            NMPC_WAVE_SYNC();
            const double psiA = point_scalar(psi, 0), psiB = point_scalar(psi, 1), psiC = point_scalar(psi, 2);
            NMPC2_TK(3);
#define NMPC2_TAKE_TRIAL(PSI, K) NMPC2_TAKE_TRIAL_(PSI, NMPC2_LOAD_GRAD(Grd + 2 * (K) * H2_ENT, gv, gw))
#define NMPC2_TAKE_TRIAL_(PSI, FETCH)                                                  \
            do {                                                                       \
                n_grad++;                                                              \
                st4<H2_COLS>(Cq, tc, gv, gw);                     /* cache_previous_gradient */     \
                cost = (PSI);                                                          \
                FETCH;                                                                 \
                const double omt_ = 1.0 - tau;                                         \
                pv = fma2(-tau, dv, fma2(-omt_, rv, uv));                              \
                pw = fma2(-tau, dw, fma2(-omt_, rw, uw));                              \
                NMPC2_HALF_STEP(pv, pw);                                               \
                lhs = NMPC2_FBE(pv, pw);                                               \
                const bool bad_ = __any(lhs > rhs_ls);                                 \
                rejected = bad_ && ls_n < MAX_LINESEARCH_ITERATIONS;                   \
                exhausted = bad_ && !rejected && a.op.ls_failure == 1;                 \
                if (rejected) { tau /= 2.0; ls_n++; }                                  \
            } while (0)
            double lhs = 0.0;
            bool rejected = false, exhausted = false;

            if (state == D_INIT) {
                n_grad += 2;
                cost = psiA;
                NMPC2_LOAD_GRAD(Grd, gv, gw);
                D2 g1v_, g1w_;
                NMPC2_LOAD_GRAD(Grd + 2 * H2_ENT, g1v_, g1w_);
                const D2 d1 = g1v_ - gv, d2_ = g1w_ - gw;
                Lc = sqrt(group_sum<P>(hdot2(d1, d2_, d1, d2_), lane)) / norm_h;
                gamma = GAMMA_L_COEFF / fmax(Lc, MIN_LIPSCHITZ_CONSTANT);
                sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * gamma);
                c_lip = GAMMA_L_COEFF / (2.0 * gamma);
                Lpar[19] = 0.5 / gamma;      // (the envelope's factor; helpers read it with the request)
                NMPC2_HALF_STEP(uv, uw);
                fbe_ok = false;
                f_begin = true;
            } else if (state == D_LIP || state == D_ITER) {
                n_cost++;
                const double rhs = cost + LIPSCHITZ_UPDATE_EPSILON * fabs(cost) - gr + c_lip * nr2;
                if (lip_it < MAX_LIPSCHITZ_UPDATE_ITERATIONS && __any((Lc < MAX_LIPSCHITZ_CONSTANT) & (psiA > rhs))) {
                    f_back = true;
                } else {
                    if (state == D_LIP) {
                        lb_first = false; st4<H2_COLS>(Cos, tc, uv, uw); st4<H2_COLS>(Cog, tc, rv, rw);
                        if (iteration == 0) {
                            n_grad++;
                            uv = hv; uw = hw;
                            cost = psiA;
                            NMPC2_LOAD_GRAD(Grd, gv, gw);
                            NMPC2_HALF_STEP(uv, uw);
                            fbe_ok = false;
                            f_end = true;
                        } else {
                            dv = rv; dw = rw;                            // empty buffer: d = r
                            rhs_ls = NMPC2_FBE(uv, uw) - sigma * nr2;
                            tau = 1.0; ls_n = 0;
                            if (a.op.ls_failure == 1) st4<H2_COLS>(Cgk, tc, gv, gw);
                            f_trials = true;
                        }
                    } else {
                        lb_first = n_first; lb_head = n_head; lb_active = n_active; H0 = n_H0;      // commit
                        if (n_take_old) { st4<H2_COLS>(Cos, tc, uv, uw); st4<H2_COLS>(Cog, tc, rv, rw); }
                        if (a.op.ls_failure == 1) st4<H2_COLS>(Cgk, tc, gv, gw);
                        NMPC2_TAKE_TRIAL(psiB, 1);                       // tau = 1
                        if (rejected) NMPC2_TAKE_TRIAL(psiC, 2);         // tau = 1/2
                        if (posted) {
                            const lds_double2 *prev_ag = nullptr;
                            for (int k = 0; k < 3 && rejected; ++k) {
                                int hid = -1;
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
                                        lds_int *dn = ctl + CTL_DONE + (wid * 3 + k) * TEAM_WAVES;
                                        for (;;) {
#pragma unroll
                                            for (int w2 = 0; w2 < TEAM_WAVES; ++w2) if (ctl_load(dn + w2) == (int)team_seq) hid = w2;
                                            if (hid >= 0) break;
                                            __builtin_amdgcn_s_sleep(1);
                                        }
                                    }
                                }
                                hid = __builtin_amdgcn_readfirstlane(hid);
                                if (hid < 0) break;
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                n_pass++;
                                const lds_double *ar = (const lds_double *)lds + hid * slice + (wid * 3 + k) * TEAM2_AREA_DOUBLES;
                                const lds_double2 *ag = (const lds_double2 *)ar;
                                const lds_double *sc_ = ar + 3 * H2_ENT * 4;      // psi[3] | (pad) | envelope[3]
                                int jstop = 3;
#pragma unroll
                                for (int j = 2; j >= 0; --j) {
                                    const bool rej_j = __any(sc_[4 + j] > rhs_ls) && ls_n + j < MAX_LINESEARCH_ITERATIONS;
                                    if (!rej_j) jstop = j;
                                }
                                if (jstop == 3) {
                                    n_grad += 3; ls_n += 3; tau *= 0.125;
                                    prev_ag = ag + 2 * (2 * H2_ENT);
                                } else {
                                    n_grad += (unsigned)jstop + 1u; ls_n += jstop;
                                    tau *= jstop == 0 ? 1.0 : (jstop == 1 ? 0.5 : 0.25);
                                    if (jstop > 0) prev_ag = ag + 2 * ((jstop - 1) * H2_ENT);
                                    if (prev_ag) { D2 p1, p2; NMPC2_LOAD_GRAD(prev_ag, p1, p2); st4<H2_COLS>(Cq, tc, p1, p2); }
                                    else st4<H2_COLS>(Cq, tc, gv, gw);
                                    cost = sc_[jstop];
                                    NMPC2_LOAD_GRAD(ag + 2 * (jstop * H2_ENT), gv, gw);
                                    const double omt_ = 1.0 - tau;
                                    pv = fma2(-tau, dv, fma2(-omt_, rv, uv));
                                    pw = fma2(-tau, dw, fma2(-omt_, rw, uw));
                                    NMPC2_HALF_STEP(pv, pw);
                                    lhs = sc_[4 + jstop];
                                    exhausted = __any(lhs > rhs_ls) && a.op.ls_failure == 1;
                                    rejected = false;
                                }
                            }
                            if (rejected && prev_ag) NMPC2_LOAD_GRAD(prev_ag, gv, gw);
                            posted = false;
                            if (lane == 0) ctl_store(ctl + CTL_CLAIM + wid, 0);
                        }
                        if (rejected) f_trials = true;
                        else if (exhausted) f_fb = true;
                        else { uv = pv; uw = pw; fbe_u = lhs; fbe_ok = true; f_end = true; }
                    }
                }
            } else if (state == D_LS) {
                NMPC2_TAKE_TRIAL(psiA, 0);
                if (rejected) NMPC2_TAKE_TRIAL(psiB, 1);
                if (rejected) NMPC2_TAKE_TRIAL(psiC, 2);
                if (rejected) f_trials = true;
                else if (exhausted) f_fb = true;
                else { uv = pv; uw = pw; fbe_u = lhs; fbe_ok = true; f_end = true; }
            } else if (state == D_FB) {
                n_grad++;
                uv = hv; uw = hw;
                cost = psiA;
                NMPC2_LOAD_GRAD(Grd, gv, gw);
                NMPC2_HALF_STEP(uv, uw);
                fbe_ok = false;
                f_end = true;
            } else {    // D_ALM: F1, F2 at the inner solution (evaluation layout)
                n_cost++;
                const D2 tv = fma2(cbar_inv, yv, eav), tw = fma2(cbar_inv, yw, eaw);
                const D2 cv = clamp2(tv, a.pb.amin, a.pb.amax), cw_ = clamp2(tw, -a.pb.awmax, a.pb.awmax);
                const D2 ypv = D2{inea ? fma(pen_c, eav.a - cv.a, yv.a) : 0.0, ineb ? fma(pen_c, eav.b - cv.b, yv.b) : 0.0};
                const D2 ypw = D2{inea ? fma(pen_c, eaw.a - cw_.a, yw.a) : 0.0, ineb ? fma(pen_c, eaw.b - cw_.b, yw.b) : 0.0};
                st4<H2_COLS>(Cyp, te, ypv, ypw);
                const D2 d1 = ypv - yv, d2_ = ypw - yw;
                dy_norm_plus = sqrt(group_sum<20>((inea ? fma(d1.a, d1.a, d2_.a * d2_.a) : 0.0) + (ineb ? fma(d1.b, d1.b, d2_.b * d2_.b) : 0.0), lane));
                dy_norm_plus = point_scalar(dy_norm_plus, 0);
                f2_norm_plus = point_scalar(sqrt(pen), 0);
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
                    st4<H2_COLS>(Cy, te, ypv, ypw);
                    dy_norm = dy_norm_plus; f2_norm = f2_norm_plus;
                    nu++;
                    if (nu == a.op.max_outer) { final_status = NMPC_NOT_CONVERGED_ITERATIONS; running = false; }
                    else if (timed_out) { final_status = NMPC_NOT_CONVERGED_OUT_OF_TIME; running = false; nu--; }
                    else {
                        // an outer-iteration boundary: cold instances step aside while others wait, long ones time-share once they outnumber
                        // the waves (the argument and the numbers: nmpc_solve_hyb.h; here there is no favoured wave slot, hence no migration)
                        bool yield_ = false;
                        if (a.sched_mode > 0) {
                            int dec = 0;
                            if (lane == 0) {
                                const bool fresh_left = (int)__hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.B;
                                const bool long_now = !(crit2 && dy_norm_plus <= pen_c * a.op.delta_tolerance + SMALL);
                                unsigned int *n_long = a.pool_ctr + POOL_CTRS * NPOOLS;
                                if (long_now != long_counted) __hip_atomic_fetch_add(n_long, long_now ? 1u : ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const bool long_wait = pool_depth(a, POOL_LONG) > 0;
                                int y;
                                const int alive = (int)__hip_atomic_load(n_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (!long_now) y = (fresh_left || long_wait) && alive >= a.sched_cold_cap && n_pass >= 150u;
                                else y = (fresh_left || long_wait) && alive >= a.sched_long_cap && n_pass - q_pass >= 150u;
                                dec = y | (long_now ? 2 : 0);
                            }
                            dec = __builtin_amdgcn_readfirstlane(dec);
                            yield_ = (dec & 1) != 0; long_counted = (dec & 2) != 0; park_cls = long_counted ? POOL_LONG : POOL_COLD;
                        }
                        q_pass = n_pass;
                        if (yield_) { parked = true; running = false; }
                        else f_start = true;
                    }
                }
            }
        }

        // ------------------------------------------------------------------ stepped aside: state out, into the pool
        if (parked) {
            double *po = a.park + (size_t)inst * park_stride(N);
            NMPC_WAVE_SYNC();          // (the multipliers were stored by the evaluation lanes, the state lanes read them)
            if (in && h == 0) {
                D2 q1, q2;
                ld4<H2_COLS>(Cq, tc, q1, q2);
                po[4 * t] = uv.a; po[4 * t + 1] = uw.a; po[4 * N + 4 * t] = q1.a; po[4 * N + 4 * t + 1] = q2.a;
                if (inb) { po[4 * t + 2] = uv.b; po[4 * t + 3] = uw.b; po[4 * N + 4 * t + 2] = q1.b; po[4 * N + 4 * t + 3] = q2.b; }
                D2 y1, y2;
                ld4<H2_COLS>(Cy, tc, y1, y2);
                po[2 * N + 2 * t] = y1.a; po[3 * N + 2 * t] = y2.a;
                if (inb) { po[2 * N + 2 * t + 1] = y1.b; po[3 * N + 2 * t + 1] = y2.b; }
            }
            if (lane == 0) {
                double *ps_ = po + 6 * N;
                ps_[0] = pen_c; ps_[1] = eps_nu; ps_[2] = dy_norm; ps_[3] = f2_norm; ps_[4] = dy_norm_plus; ps_[5] = f2_norm_plus; ps_[6] = last_fpr; ps_[7] = last_cost;
                ps_[8] = (double)nu; ps_[9] = (double)inner_total; ps_[10] = (double)n_cost; ps_[11] = (double)n_grad; ps_[12] = (double)n_pass;
                ps_[13] = (double)t_start; ps_[14] = 0.0; ps_[15] = long_counted ? 1.0 : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) pool_push(a, park_cls, inst);
            NMPC_WAVE_SYNC();
            continue;
        }
        // ------------------------------------------------------------------ results
        if (lane == 0 && long_counted) __hip_atomic_fetch_add(a.pool_ctr + POOL_CTRS * NPOOLS, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (in && h == 0) {
            double *uo = a.u + (size_t)inst * a.n_u;
            uo[4 * t] = uv.a; uo[4 * t + 1] = uw.a;
            if (inb) { uo[4 * t + 2] = uv.b; uo[4 * t + 3] = uw.b; }
            if (a.y_out) {
                D2 y1, y2;
                ld4<H2_COLS>(Cyp, tc, y1, y2);
                double *yo = a.y_out + (size_t)inst * a.n1;
                yo[2 * t] = y1.a; yo[N + 2 * t] = y2.a;
                if (inb) { yo[2 * t + 1] = y1.b; yo[N + 2 * t + 1] = y2.b; }
            }
        }
        if (lane == 0 && a.st) {
            nmpc_status s;
            s.exit_status = final_status;
            s.num_outer_iterations = (uint32_t)(final_status == NMPC_NOT_CONVERGED_NOT_FINITE ? nu + 1 : (nu < a.op.max_outer ? nu + 1 : nu));
            s.num_inner_iterations = inner_total;
            s.num_cost_evals = n_cost;
            s.num_grad_evals = n_grad;
            s.reserved = n_pass;
            s.last_problem_norm_fpr = last_fpr;
            s.delta_y_norm_over_c = dy_norm_plus / pen_c;
            s.f2_norm = f2_norm_plus;
            s.penalty = pen_c;
            s.cost = last_cost;
            s.solve_time_ms = (double)((long long)__builtin_amdgcn_s_memrealtime() - t_start) * 1e-5;
            if (a.dbg) {       // experiments (NMPC_DEBUG_PRIO, scripts/utilisation.py): start and finish on the 100 MHz clock, workgroup and wave
                s.delta_y_norm_over_c = (double)t_start; s.cost = (double)(long long)__builtin_amdgcn_s_memrealtime();
                s.f2_norm = (double)(blockIdx.x * TEAM_WAVES + wid);
            }
            a.st[inst] = s;
        }
        NMPC_WAVE_SYNC();          // the LDS slice is reused by the next instance
    }

    // ====================================================================== helper: no work of its own (any more)
    if (lane == 0) ctl_store(ctl + CTL_INST + wid, -1);      // (nmpc_solve_hyb.h: before the first result area lands in this slice)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        if (wid < a.team_owners) ctl_add(ctl + CTL_OWNERS, -1);
        ctl_add(ctl + CTL_HELPERS, 1);
    }
    WinState ws_h[2] = {{2 * te < N - 1 ? 2 * te : N - 2, 0.0, 0.0, 0.0}, {2 * te + 1 < N - 1 ? 2 * te + 1 : N - 2, 0.0, 0.0, 0.0}};      // valid for instance `ws_inst`
    ObsCert2 oc_h = {d2s(0.0), d2s(0.0), d2s(0.0), 0, 0, 0};
    int ws_inst = -1;
    for (;;) {
        if (!a.team_help || __builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_OWNERS)) <= 0) break;      // (nobody will ask: NMPC_TEAM_HELP=0)
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
        const int seq = (got & 0x0fffffff) >> 8;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        lds_double *Lw = (lds_double *)lds + w * slice;
        const lds_double2 *rq = (const lds_double2 *)(Lw + mp.req);
        D2 u1, u2, r1, r2, e1, e2, yv, yw;
        ld4<H2_ENT>(rq, te, u1, u2);
        ld4<H2_ENT>(rq + 2 * H2_ENT * 1, te, r1, r2);
        ld4<H2_ENT>(rq + 2 * H2_ENT * (2), te, e1, e2);
        const double c_w = Lw[mp.par + 15], cbar_w = Lw[mp.par + 16], gam_w = Lw[mp.par + 17], hig_w = Lw[mp.par + 19];
        ld4<H2_COLS>((const lds_double2 *)(Lw + mp.vec) + 2 * H2_COLS * (4), te, yv, yw);
        const double tau_w = __hiloint2double((1023 - (2 + 3 * k + q)) << 20, 0), omt_w = 1.0 - tau_w;
        const D2 zv = fma2(-tau_w, e1, fma2(-omt_w, r1, u1)), zw = fma2(-tau_w, e2, fma2(-omt_w, r2, u2));
        double psi, pen;
        D2 egv = d2s(0.0), egw = d2s(0.0), eav, eaw;
        {                                   // another instance's reference: what this lane knew about its windows is void
            const int inst_w = __builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_INST + w));
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (inst_w != ws_inst || inst_w < 0) { ws_inst = inst_w; ws_h[0].mo2 = 0.0; ws_h[1].mo2 = 0.0; oc_h.m2 = d2s(0.0); }
        }
        eval_psi2<SH, false, NMPC_WIN2>(a, Lw, mp, f2off, lane, te, zv, zw, c_w, cbar_w, yv, yw, true, psi, pen, egv, egw, eav, eaw, ws_h, &oc_h);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (__builtin_amdgcn_readfirstlane(ctl_load(ctl + CTL_INST + w)) != ws_inst) { ws_inst = -1; ws_h[0].mo2 = 0.0; ws_h[1].mo2 = 0.0; oc_h.m2 = d2s(0.0); }      // (the owner moved on meanwhile)
        // the trial's forward-backward envelope, in the evaluation layout (the same canonical sums as the state layout's)
        const D2 s1_ = fma2(-gam_w, egv, zv), s2_ = fma2(-gam_w, egw, zw);
        const D2 x1_ = D2{s1_.a - (inea ? clampd(s1_.a, vmin, vmax) : s1_.a), s1_.b - (ineb ? clampd(s1_.b, vmin, vmax) : s1_.b)};
        const D2 x2_ = D2{s2_.a - (inea ? clampd(s2_.a, -wmax, wmax) : s2_.a), s2_.b - (ineb ? clampd(s2_.b, -wmax, wmax) : s2_.b)};
        const double dist2_ = group_sum<20>((inea ? fma(x1_.a, x1_.a, x2_.a * x2_.a) : 0.0) + (ineb ? fma(x1_.b, x1_.b, x2_.b * x2_.b) : 0.0), lane);
        const double gg_ = group_sum<20>((inea ? fma(egv.a, egv.a, egw.a * egw.a) : 0.0) + (ineb ? fma(egv.b, egv.b, egw.b * egw.b) : 0.0), lane);
        const double lhs_ = psi - (0.5 * gam_w) * gg_ + dist2_ * hig_w;
        lds_double *ar = L + (w * 3 + k) * TEAM2_AREA_DOUBLES;
        st4<H2_ENT>((lds_double2 *)ar + 2 * H2_ENT * (q), te, egv, egw);
        if (te == 0) { ar[3 * H2_ENT * 4 + q] = psi; ar[3 * H2_ENT * 4 + 4 + q] = lhs_; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ctl_store(ctl + CTL_DONE + (w * 3 + k) * TEAM_WAVES + wid, seq);
    }
}

#undef NMPC2_TK
#undef NMPC2_HALF_STEP
#undef NMPC2_FBE
#undef NMPC2_LOAD_GRAD
#undef NMPC2_TAKE_TRIAL
#undef NMPC2_TAKE_TRIAL_

}  // namespace nmpc
