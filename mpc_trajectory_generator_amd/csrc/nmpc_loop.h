// nmpc_loop.h -- the receding-horizon loop on device: B robots following one route in lock step.
//
// Counterpart of the body of the reference's `PathGenerator.run` loop (src/path_generator.py:290-403:
// closest reference sample in the sliding window :320-325, horizon padded with the end pose :326-341,
// braking `vel_ref` :343-361, dynamic block rotated left and refreshed :306-316, `last_u` :371-374,
// parameter concatenation :378-379, terminal test :397), of the closed-form obstacle predictor
// (src/visibility/visibility.py:156-166,199-216) and of `MpcModule.run`'s Euler state advance
// (src/mpc/mpc_generator.py:223-235).  One step = assemble p -> batched solve (warm start from the
// previous controls and multipliers) -> advance; nothing crosses PCIe between steps.
//
// Arithmetic: index selections (window arg-min, vertex window, braking-table filter) use the same
// unfused IEEE operations as the host mirror (`trajectory.VectorizedRecedingHorizon`), so they are
// bit-identical to it; sin / cos are the kernels' own sincos_cw (the mirror takes it as a hook).
#pragma once

namespace nmpc {

struct LoopArgs {
    int B, N, nobs, ndyn, K, n_p, n_u, n_ref, n_vert, n_brake, s, t;
    double ts, base, radius, pad;
    double end[3];
    double w[10];
    const double *xr, *yr, *thr, *vert, *bv, *bd;
    const double *dynpar;     // [B][K][10]: p1x p1y p2x p2y freq rx ry angle | sinusoidal law (0 / 1) | atan2(p2 - p1)
    double *state;            // [B][3]
    double *last_u;           // [B][2]
    int *idx;                 // [B]
    const double *dyn_in;     // [B][ndyn][N][5]
    double *dyn_out;
    double *P;                // [B][n_p]
    const double *U;          // [B][n_u]
    unsigned char *done;      // [B]
    double *traj;             // [(steps * s + 1)][B][3] or NULL
    int traj_row;             // rows already written
};

// (d, j) lexicographic minimum over the wave, result in every lane: the FIRST minimal index, like np.argmin
__device__ __forceinline__ void wave_argmin(double &d, int &j)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(d, off);
        const int oj = __shfl_xor(j, off);
        if (od < d || (od == d && oj < j)) { d = od; j = oj; }
    }
}

// times = numpy.linspace(t0, t0 + H * ts, H)[i]   (visibility.py:204)
__device__ __forceinline__ double linspace_at(double t0, double ts, int H, int i)
{
    const double stop = t0 + (double)H * ts;
    if (H == 1) return t0;
    if (i == H - 1) return stop;
    const double step = (stop - t0) / (double)(H - 1);
    return (double)i * step + t0;
}

// one wave per robot: fills p[b] and the rotated dynamic block
__global__ __launch_bounds__(64) void nmpc_loop_assemble_kernel(LoopArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N, n = a.n_ref, s = a.s;
    const double x = a.state[3 * b], y = a.state[3 * b + 1], th = a.state[3 * b + 2];
    double *p = a.P + (size_t)b * a.n_p;
    constexpr int NZ_ = 20;

    // ---- static circles (path_generator.py:295-304 + visibility.py:141-148, look-back 0)
    {
        double *pc = p + NZ_ + N;
        const int nv = a.n_vert;
        int lb = 0, ub = nv;
        if (nv > a.nobs) {
            double best = __builtin_inf();
            int bj = 0x7fffffff;
            for (int j = lane; j < nv; j += 64) {
                const double dx = a.vert[2 * j] - x, dy = a.vert[2 * j + 1] - y;
                double d = sqrt(dx * dx + dy * dy);
                if (d != d) d = -__builtin_inf();
                if (d < best) { best = d; bj = j; }
            }
            wave_argmin(best, bj);
            lb = bj;
            ub = a.nobs;          // (sic: the reference's window is [lb, min(nv, Nobs)) )
        }
        for (int k = lane; k < a.nobs; k += 64) {
            const int j = lb + k;
            const bool ok = nv > 0 && j < ub;
            const int jj = j < nv ? j : (nv > 0 ? nv - 1 : 0);
            pc[3 * k] = ok ? a.vert[2 * jj] : 0.0;
            pc[3 * k + 1] = ok ? a.vert[2 * jj + 1] : 0.0;
            pc[3 * k + 2] = ok ? a.radius : 0.0;
        }
    }
    // ---- dynamic ellipses (path_generator.py:306-316; visibility.py:156-166,199-216)
    {
        const int per = N * 5, tot = a.ndyn * per;
        const double *din = a.dyn_in + (size_t)b * tot;
        double *dout = a.dyn_out + (size_t)b * tot;
        double *pd = p + NZ_ + N + 3 * a.nobs;
        for (int e = lane; e < tot; e += 64) {
            const int k = e / per, r = e - k * per, st = r / 5, f = r - st * 5;
            double v;
            const bool fresh = k < a.K && (a.t == 0 || st >= N - s);
            if (fresh) {
                const double *q = a.dynpar + ((size_t)b * a.K + k) * 10;
                if (f < 2) {
                    const int H = a.t == 0 ? N : s, i = a.t == 0 ? st : st - (N - s);
                    const double t0 = a.t == 0 ? 0.0 : (double)(a.t + N - s) * a.ts;
                    const double tm = linspace_at(t0, a.ts, H, i);
                    double sn, cs;
                    sincos_cw(q[4] * tm, sn, cs);
                    const double w = fabs(sn);
                    v = w * q[f] + (1.0 - w) * q[2 + f];
                    if (q[8] != 0.0) {
                        // sinusoidal law (visibility.py:177-196): offset across the p1 -> p2 line, amplitude 1.5
                        const double p3x = w * q[0] + (1.0 - w) * q[2], p3y = w * q[1] + (1.0 - w) * q[3];
                        double s10, c10, sa_, ca_;
                        sincos_cw((10.0 * q[4]) * tm, s10, c10);
                        sincos_cw(q[9], sa_, ca_);
                        const double add = 1.5 * c10;
                        const double dx = p3x - q[0], dy = p3y - q[1];
                        const double rx = ca_ * dx - sa_ * dy;
                        double ry = sa_ * dx + ca_ * dy;
                        ry = ry + add;
                        const double qx = ca_ * rx - (-sa_) * ry, qy = (-sa_) * rx + ca_ * ry;      // rotate back by -angle
                        v = f == 0 ? qx + q[0] : qy + q[1];
                    }
                } else {
                    v = f == 2 ? q[5] + a.pad : (f == 3 ? q[6] + a.pad : q[7]);
                }
            } else if (a.t == 0) {
                v = din[e];                                    // padding block as initialised
            } else {
                // the reference rotates the WHOLE flat list left by 5 s entries (path_generator.py:312): a shift by s
                // stages inside a block, and a block's last s stages take the next block's first s (the last block's
                // take block 0's, so a padding slot can inherit stale ellipses of obstacle 0 when 0 < K < Ndynobs)
                const int src = e + 5 * s;
                v = din[src < tot ? src : src - tot];
            }
            dout[e] = v;
            pd[e] = v;
        }
    }
    // ---- closest reference sample in the sliding window (:320-325)
    int idx;
    {
        const int i0 = a.idx[b];
        const int lb = i0 - s > 0 ? i0 - s : 0, ub = i0 + 5 * s < n ? i0 + 5 * s : n;
        double best = __builtin_inf();
        int bj = 0x7fffffff;
        for (int j = lb + lane; j < ub; j += 64) {
            const double dx = a.xr[j] - x, dy = a.yr[j] - y;
            double d = sqrt(dx * dx + dy * dy);
            if (d != d) d = -__builtin_inf();
            if (d < best) { best = d; bj = j; }
        }
        wave_argmin(best, bj);
        idx = bj;
        if (lane == 0) a.idx[b] = idx;
    }
    // ---- head: state, last_u, target, last_u again, weights (:378-379)
    {
        const bool far = idx + N < n;
        const int jf = far ? idx + N : n - 1;
        if (lane < 3) p[lane] = lane == 0 ? x : (lane == 1 ? y : th);
        if (lane >= 3 && lane < 5) p[lane] = a.last_u[2 * b + lane - 3];
        if (lane >= 5 && lane < 8) {
            const int f = lane - 5;
            p[lane] = far ? (f == 0 ? a.xr[jf] : (f == 1 ? a.yr[jf] : a.thr[jf])) : a.end[f];
        }
        if (lane >= 8 && lane < 10) p[lane] = a.last_u[2 * b + lane - 8];
        if (lane >= 10 && lane < 20) p[lane] = a.w[lane - 10];
    }
    // ---- horizon references (:326-341) and velocity reference with the braking profile (:343-361)
    {
        double *pv = p + NZ_, *pr = p + NZ_ + N + 3 * a.nobs + 5 * a.ndyn * N;
        const bool brake = (double)(idx + N) >= (double)n - a.bd[0] / a.base;
        const int nbase = n - idx - 1 < N ? n - idx - 1 : N;
        const double ddx = x - a.end[0], ddy = y - a.end[1];
        const double dist_to_goal = sqrt(ddx * ddx + ddy * ddy);
        for (int k = lane; k < N; k += 64) {
            const int j = idx + k;
            const bool ok = j < n;
            const int jj = ok ? j : n - 1;
            pr[3 * k] = ok ? a.xr[jj] : a.end[0];
            pr[3 * k + 1] = ok ? a.yr[jj] : a.end[1];
            pr[3 * k + 2] = ok ? a.thr[jj] : a.end[2];
            double v = a.base;
            if (brake) {
                if (nbase == 0) {
                    // inside the last sample: the k-th braking entry whose distance is within reach (:347-351)
                    v = 0.0;
                    int cnt = 0;
                    for (int i = 0; i < a.n_brake; ++i) {
                        if (a.bd[i] <= dist_to_goal) {
                            if (cnt == k) { v = a.bv[i]; break; }
                            ++cnt;
                        }
                    }
                } else if (k >= nbase) {
                    v = k - nbase < a.n_brake ? a.bv[k - nbase] : 0.0;
                }
            }
            pv[k] = v;
        }
    }
}

// one thread per robot: Euler advance over the steps taken (mpc_generator.py:225-235), terminal test (:397)
__global__ void nmpc_loop_advance_kernel(LoopArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const double *u = a.U + (size_t)b * a.n_u;
    double x = a.state[3 * b], y = a.state[3 * b + 1], th = a.state[3 * b + 2];
    for (int i = 0; i < a.s; ++i) {
        const double v = u[2 * i], w = u[2 * i + 1];
        double sn, cs;
        sincos_cw(th, sn, cs);
        x = x + a.ts * (v * cs);
        y = y + a.ts * (v * sn);
        th = th + a.ts * w;
        if (a.traj) {
            double *row = a.traj + ((size_t)(a.traj_row + i) * a.B + b) * 3;
            row[0] = x; row[1] = y; row[2] = th;
        }
    }
    a.state[3 * b] = x; a.state[3 * b + 1] = y; a.state[3 * b + 2] = th;
    const double lv = u[2 * (a.s - 1)], lw = u[2 * (a.s - 1) + 1];
    a.last_u[2 * b] = lv; a.last_u[2 * b + 1] = lw;
    a.done[b] = (fabs(x - a.end[0]) <= 0.05 && fabs(y - a.end[1]) <= 0.05 && fabs(lv) < 0.005) ? 1 : 0;
}

}  // namespace nmpc
