"""Reproducer for the schedule-dependent failures of nmpc_solve_hyb2_kernel (config 2, N_hor = 40).

One process per library build (NMPC_LIB_PATH), so that builds can be compared on one box:
  python scripts/hyb2_repro.py ref                      # shipped library: writes gpurun_out/hyb2_ref.npz (checked against the oracle on a sample)
  NMPC_LIB_PATH=scripts/variants/x.so python scripts/hyb2_repro.py probe x
For a build whose full batch differs from the reference (or from itself between launches) the probe narrows it down:
  one instance per wave (B = resident waves: no wave ever takes a second instance), and the differing instances alone (B = 1).
Prints one JSON line per experiment."""
import json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import oracle_for, STATUS_FIELDS
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

REF = "gpurun_out/hyb2_ref.npz"
cfg = named_config("cfg2")
B = 8192
P = synthetic_batch(cfg, 11, B, 0, routes=random_routes(cfg, 11, 32, seed=1000))


def solve(P_, env=None, reps=1):
    for k, v in (env or {}).items():
        os.environ[k] = v
    s = BatchSolver(cfg, max_batch=max(len(P_), 1))
    out = []
    for _ in range(reps):
        u, y, st = s.solve(P_)
        out.append((u, y, st, s.last_batch_ms))
    s.close()
    for k in (env or {}):
        os.environ.pop(k, None)
    return out


def differs(a, b):
    """indices of the instances whose bits differ between two results"""
    (u, y, st, _), (u2, y2, st2, _) = a, b
    bad = np.any(u != u2, axis=1) | np.any(y != y2, axis=1)
    for f in STATUS_FIELDS:
        bad |= st[f] != st2[f]
    return np.nonzero(bad)[0]


mode = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "shipped"
os.makedirs("gpurun_out", exist_ok=True)
if mode == "ref":
    r = solve(P, reps=2)
    d = differs(r[0], r[1])
    idx = np.random.default_rng(7).choice(B, 32, replace=False)
    uo, yo, sto = oracle_for(cfg).solve_batch(P[idx], threads=16)
    ok = np.array_equal(r[0][0][idx], uo) and np.array_equal(r[0][1][idx], yo) and all(np.array_equal(r[0][2][f][idx], sto[f]) for f in STATUS_FIELDS)
    np.savez(REF, u=r[0][0], y=r[0][1], st=r[0][2])
    print(json.dumps({"exp": "ref", "lib": tag, "launch_vs_launch": len(d), "sample_equals_oracle": bool(ok), "ms": round(r[0][3], 1)}), flush=True)
    sys.exit(0)

z = np.load(REF)
ref = (z["u"], z["y"], z["st"], 0.0)
suspects = set()
for help_ in ("1", "0"):
    r = solve(P, {"NMPC_TEAM_HELP": help_}, reps=3)
    vs_ref = [differs(x, ref) for x in r]
    vs_self = [differs(x, r[0]) for x in r[1:]]
    for d in vs_ref + vs_self:
        suspects.update(int(i) for i in d)
    print(json.dumps({"exp": "full_batch", "lib": tag, "help": help_, "differ_from_ref": [len(d) for d in vs_ref],
                      "differ_from_first_launch": [len(d) for d in vs_self], "ids": sorted(suspects)[:40],
                      "ref_inner": [int(ref[2]["num_inner_iterations"][i]) for i in sorted(suspects)[:40]],
                      "ms": [round(x[3], 1) for x in r]}), flush=True)
if not suspects:
    print(json.dumps({"exp": "verdict", "lib": tag, "ok": True}), flush=True)
    sys.exit(0)

sus = np.array(sorted(suspects))
# (a) exactly one instance per wave: 1024 instances = 256 workgroups x 4 owners; suspects first, filled up with others
rest = np.setdiff1d(np.arange(B), sus)
sel = np.concatenate([sus, rest])[:1024]
refsel = tuple(x[sel] for x in ref[:3]) + (0.0,)
for help_ in ("1", "0"):
    r = solve(P[sel], {"NMPC_TEAM_HELP": help_, "NMPC_TEAM_OWNERS": "4"}, reps=4)
    print(json.dumps({"exp": "one_instance_per_wave", "lib": tag, "help": help_, "n_suspects_in": int(min(len(sus), 1024)),
                      "differ_from_ref": [len(differs(x, refsel)) for x in r],
                      "differ_from_first_launch": [len(differs(x, r[0])) for x in r[1:]]}), flush=True)
# (b) two instances per wave in a fixed order (B = 2048, owners = 4 -> 256 workgroups, every wave takes two; which two is timing)
sel2 = np.concatenate([sus, rest])[:2048]
refsel2 = tuple(x[sel2] for x in ref[:3]) + (0.0,)
r = solve(P[sel2], {"NMPC_TEAM_HELP": "0", "NMPC_TEAM_OWNERS": "4"}, reps=4)
print(json.dumps({"exp": "two_instances_per_wave", "lib": tag, "help": "0", "differ_from_ref": [len(differs(x, refsel2)) for x in r],
                  "differ_from_first_launch": [len(differs(x, r[0])) for x in r[1:]]}), flush=True)
# (c) suspects alone, B = 1 (one owner + three helpers, or helpers off), repeated
for i in sus[:8]:
    row = {"exp": "alone", "lib": tag, "inst": int(i), "ref_inner": int(ref[2]["num_inner_iterations"][i])}
    refi = tuple(x[i:i + 1] for x in ref[:3]) + (0.0,)
    for help_ in ("1", "0"):
        r = solve(P[i:i + 1], {"NMPC_TEAM_HELP": help_}, reps=6)
        row[f"help{help_}_differ_from_ref"] = [len(differs(x, refi)) for x in r]
        row[f"help{help_}_inner"] = [int(x[2]["num_inner_iterations"][0]) for x in r]
    print(json.dumps(row), flush=True)
print(json.dumps({"exp": "verdict", "lib": tag, "ok": False, "n_suspects": len(sus)}), flush=True)
