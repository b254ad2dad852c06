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


# ---- round-robin variants at outer-iteration boundaries (quantum = one outer iteration; the remaining passes of an instance are spread evenly
#      over its remaining outer iterations) ----
def quanta(i):
    n = int(z["outer"][i])
    if n <= 1: return [p[i]]
    return [min(p1[i], p[i])] + [rem[i] / (n - 1)] * (n - 1)

def rr(policy):
    """policy 'pure': one FIFO, every instance re-enqueued after each outer iteration.
       policy 'classes': fresh first (discovery), then revealed-long instances (time-shared when they outnumber the waves), shorts fill in"""
    from collections import deque
    Q = {i: quanta(i) for i in range(B)}
    fresh = deque(order); longq = deque(); shortq = deque()
    h = [(0.0, k) for k in range(S)]; heapq.heapify(h); end = 0.0
    run = {}          # wave -> (instance, next quantum index)
    while h:
        t, k = heapq.heappop(h)
        job = run.pop(k, None)
        if job is not None:
            i, q = job
            if q < len(Q[i]):
                if policy == "pure":
                    fresh.append((i, q))
                else:
                    if fresh: (longq if long_[i] else shortq).append((i, q))
                    elif long_[i]:
                        if longq: longq.append((i, q))
                        else: run[k] = (i, q)          # a wave of its own: goes on
                    else:
                        if longq: shortq.append((i, q))
                        else: run[k] = (i, q)
            else:
                end = max(end, t)
        if k not in run:
            nxt = None
            if fresh:
                x = fresh.popleft(); nxt = x if isinstance(x, tuple) else (x, 0)
            elif longq: nxt = longq.popleft()
            elif shortq: nxt = shortq.popleft()
            if nxt is None: continue
            run[k] = nxt
            t += 0.008          # fetch / park + prepare_instance
        i, q = run[k]
        run[k] = (i, q + 1)
        heapq.heappush(h, (t + Q[i][q] * us * 1e-3, k))
    return end

print(json.dumps({"cfg": name, "rr_pure_ms": rr("pure"), "rr_classes_ms": rr("classes")}))


def v5(theta=1.0):
    """fresh first; after its first outer iteration an instance is hot (||F2|| > delta), warm (only the multiplier criterion fails) or cold (will
    finish in its next outer iteration).  cold / warm step aside while anything more urgent waits; hot instances go on unless they outnumber the
    waves (then they time-share, one outer iteration at a time)."""
    from collections import deque
    hot_c = f2 > 1e-4; warm_c = (~hot_c) & (z["dy1"] > 1e-4)
    Q = {i: quanta(i) for i in range(B)}
    fresh = deque((i, 0) for i in order); pools = {"hot": deque(), "warm": deque(), "cold": deque()}
    h = [(0.0, k) for k in range(S)]; heapq.heapify(h); end = 0.0; run = {}
    n_hot_alive = 0; revealed = set()
    while h:
        t, k = heapq.heappop(h)
        job = run.pop(k, None)
        if job is not None:
            i, q = job
            if q >= len(Q[i]):
                end = max(end, t)
                if hot_c[i] and i in revealed: n_hot_alive -= 1
            else:
                cls = "hot" if hot_c[i] else ("warm" if warm_c[i] else "cold")
                if cls == "hot" and i not in revealed: revealed.add(i); n_hot_alive += 1
                if cls == "cold": y = bool(fresh or pools["hot"] or pools["warm"])
                elif cls == "warm": y = bool(fresh or pools["hot"])
                else: y = n_hot_alive >= theta * S and bool(fresh or pools["hot"])
                if y: pools[cls].append((i, q))
                else: run[k] = (i, q)
        if k not in run:
            nxt = fresh.popleft() if fresh else next((pools[c].popleft() for c in ("hot", "warm", "cold") if pools[c]), None)
            if nxt is None: continue
            run[k] = nxt; t += 0.008
        i, q = run[k]; run[k] = (i, q + 1)
        heapq.heappush(h, (t + Q[i][q] * us * 1e-3, k))
    return end

print(json.dumps({"cfg": name, "v5_theta0.9_ms": v5(0.9), "v5_theta0.75_ms": v5(0.75), "v5_theta0.5_ms": v5(0.5), "v5_no_rr_ms": v5(1e9)}))
