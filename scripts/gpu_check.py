"""Development check on a GPU box: primitives, cost layer and solver vs the CPU oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from oracle import Oracle


def oracle_for(cfg, **kw):
    return Oracle(cfg.N_hor, cfg.Nobs, cfg.Ndynobs, cfg.ts, cfg.lin_vel_min, cfg.lin_vel_max, cfg.ang_vel_max,
                  cfg.lin_acc_min, cfg.lin_acc_max, cfg.ang_acc_max, **kw)


cfg = named_config("default")
sol = BatchSolver(cfg, max_batch=8192)
orc = oracle_for(cfg)
rng = np.random.default_rng(1)
x = np.concatenate([rng.uniform(-20, 20, 5000), rng.uniform(-1e3, 1e3, 1000), [0.0, np.pi / 2, -np.pi, 1e-300]])
s, c = sol.test_sincos(x)
so, co = np.array([orc.sincos(v) for v in x]).T
print("sincos bit-exact:", np.array_equal(s, so), np.array_equal(c, co), "max err vs libm", np.abs(s - np.sin(x)).max())
a = np.abs(rng.normal(0, 1, 100000)) * 10.0 ** rng.integers(-30, 30, 100000)
b = rng.normal(0, 1, 100000) * 10.0 ** rng.integers(-30, 30, 100000)
q, r = sol.test_divsqrt(a, b)
print("div bit-exact:", np.array_equal(q, a / b), " sqrt bit-exact:", np.array_equal(r, np.sqrt(a)),
      "mismatches", (q != a / b).sum(), (r != np.sqrt(a)).sum())

for name in ["default", "n40", "nobs50", "smooth"]:
    d = np.load(f"tests/golden/cost_{name}.npz")
    kv = dict(zip(d["cfg_keys"], d["cfg_vals"]))
    c2 = named_config({"default": "default", "n40": "cfg2", "nobs50": "cfg3", "smooth": "cfg4"}[name])
    s2 = BatchSolver(c2, max_batch=64)
    o2 = oracle_for(c2)
    nc = len(d["u"])
    for j, (cc, yy) in enumerate(zip(d["xi_c"], d["xi_y"])):
        psi, g, F1, F2 = s2.evaluate(d["p"], d["u"], np.full(nc, cc), np.tile(yy, (nc, 1)))
        ok = True
        for i in range(nc):
            po, go, F1o, F2o = o2.eval(d["p"][i], d["u"][i], cc, yy)
            ok &= (psi[i] == po) and np.array_equal(g[i], go) and np.array_equal(F1[i], F1o) and np.array_equal(F2[i], F2o)
        rel = np.max(np.abs(psi - d["psi"][:, j]) / np.abs(d["psi"][:, j]))
        relg = np.max(np.abs(g - d["grad_psi"][:, j]) / np.max(np.abs(d["grad_psi"][:, j]), axis=1, keepdims=True))
        print(f"{name} xi{j}: GPU==oracle bitwise {ok}; vs golden psi {rel:.2e} grad {relg:.2e}")
    s2.close()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = synthetic_batch(cfg, 11, B, 12345)
t = time.time(); u, y, st = sol.solve(P); dt = time.time() - t
print(f"GPU solve B={B}: {dt*1e3:.1f} ms wall, kernel {sol.last_batch_ms:.2f} ms")
t = time.time(); uo, yo, sto = orc.solve_batch(P, threads=8); dto = time.time() - t
print(f"oracle: {dto:.2f} s")
same_u = np.array([np.array_equal(u[i], uo[i]) for i in range(B)])
same_y = np.array([np.array_equal(y[i], yo[i]) for i in range(B)])
fields = ["exit_status", "num_outer_iterations", "num_inner_iterations", "num_cost_evals", "num_grad_evals",
          "last_problem_norm_fpr", "delta_y_norm_over_c", "f2_norm", "penalty", "cost"]
same_st = np.array([all(st[f][i] == sto[f][i] for f in fields) for i in range(B)])
print("bitwise equal: u", same_u.sum(), "/", B, " y", same_y.sum(), " status", same_st.sum())
if not same_st.all():
    i = int(np.argmin(same_st)); print("first diff", i, st[i], sto[i], np.abs(u[i] - uo[i]).max())
print("inner iters mean", st["num_inner_iterations"].mean(), "exit", np.bincount(st["exit_status"]))
