"""Event-driven model of scheduling policies on measured per-instance work (gpurun_out/passes_<cfg>_<seed>.npz from scripts/order_model.py):
   shipped      : instances in classifier order, each runs to its end on the wave that fetched it
   park_short   : the same fetch order, but after its FIRST outer iteration an instance whose penalty constraints hold (||F2|| <= delta) is parked
                  and the wave fetches the next fresh instance; parked ones are resumed when no fresh instance is left
   lpt          : exact longest-first (the bound no predictor reaches)
CPU only.  usage: python scripts/policy_sim.py cfgN [seed] [us_per_pass]"""
import heapq, json, sys
import numpy as np
sys.path.insert(0, ".")
name = sys.argv[1]; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
z = np.load(f"gpurun_out/passes_{name}_{seed}.npz")
lv = np.load(f"gpurun_out/levels_{name}_{seed}.npy")
p, p1, f2 = z["passes"].astype(float), z["p1"].astype(float), z["f2_1"]
S = 1024 if name == "cfg2" else 2048
us = float(sys.argv[3]) if len(sys.argv) > 3 else (6.49 if name == "cfg2" else 5.9)
B = len(p)
order = np.argsort(-lv, kind="stable")
long_ = (f2 > 1e-4) | (z["dy1"] > 1e-4) if "--dy" in sys.argv else (f2 > 1e-4)
rem = np.maximum(p - p1, 0.0)

def simple(order):
    h = [0.0] * S; heapq.heapify(h); end = 0.0
    for i in order:
        t = heapq.heappop(h) + p[i] * us * 1e-3; end = max(end, t); heapq.heappush(h, t)
    return end

def park_short():
    fresh = list(order)[::-1]; pool = []; h = [(0.0, k) for k in range(S)]; heapq.heapify(h); end = 0.0; last_long_start = 0.0
    while h:
        t, k = heapq.heappop(h)
        if fresh:
            i = fresh.pop()
            t1 = t + min(p1[i], p[i]) * us * 1e-3
            if p[i] <= p1[i]: end = max(end, t1); heapq.heappush(h, (t1, k)); continue      # finished within its first outer iteration
            if long_[i]:
                last_long_start = max(last_long_start, t)
                t2 = t1 + rem[i] * us * 1e-3; end = max(end, t2); heapq.heappush(h, (t2, k))
            else:
                pool.append(i); heapq.heappush(h, (t1 + 0.01, k))      # + parking cost
        elif pool:
            i = pool.pop(0)
            t2 = t + 0.01 + rem[i] * us * 1e-3; end = max(end, t2); heapq.heappush(h, (t2, k))
    return end, last_long_start

e_ps, lls = park_short()
print(json.dumps({"cfg": name, "us_per_pass": us, "slots": S, "work_ms": float(p.sum() * us * 1e-3 / S), "longest_ms": float(p.max() * us * 1e-3),
                  "shipped_ms": simple(order), "park_short_ms": e_ps, "park_short_last_long_start_ms": lls, "lpt_ms": simple(np.argsort(-p)),
                  "long_frac": float(long_.mean())}))
