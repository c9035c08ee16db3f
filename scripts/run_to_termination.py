"""A workload of bensolve_amd.synth.CONFIGS run to TERMINATION on one GPU (phase 2, primal variant): one JSON line with the
whole-run figures (LPs, cuts, pivots, seconds, LPs/s) and a SHA-256 of the canonicalised final sets (vertices rounded to 1e-6).
usage: run_to_termination.py WORKLOAD [batch] [eps]"""
import os, sys, time, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
name = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
prob = synth.CONFIGS[name]()
eng = BensonEngine(prob, eps=eps, pool_slots=max(4 * batch, 64))
t0 = time.perf_counter()
assert eng.start() == 0
t_start = time.perf_counter() - t0
steps = red = conf = 0
ms_lp = ms_poly = 0.0
while True:
    s = eng.step(batch); steps += 1
    red += s["redundant"]; conf += s["confirmed"]; ms_lp += s["ms_lp"]; ms_poly += s["ms_poly"]
    if steps % 100 == 0:
        print("  step %d: %d LPs so far, %d left, %.1f s" % (steps, eng.totals()["lps"], s["left"], time.perf_counter() - t0), file=sys.stderr, flush=True)
    if s["lps"] == 0 and s["left"] == 0:
        break
sec = time.perf_counter() - t0
tot = eng.totals()
eng.poly_call("dual_adjacency")
d = eng.poly_dump()
live = d["pu"].astype(bool)
X = np.round(d["X"][live], 6) + 0.0
key = np.lexsort(X.T[::-1])
sha = hashlib.sha256(np.ascontiguousarray(X[key]).tobytes() + np.ascontiguousarray(d["pi"][live][key]).tobytes()).hexdigest()
dims = eng.lp_dims() if hasattr(eng, "lp_dims") else None
row = dict(workload=name, q=prob["q"], n=prob["n"], m=prob["m"], lp_dims=dims, batch=batch, eps=eps, finished=True, seconds=round(sec, 2), start_seconds=round(t_start, 2),
           steps=steps, lps=tot["lps"], cuts=tot["cuts"], redundant=red, confirmed=conf, pivots=tot["pivots"], lps_per_sec=round(tot["lps"] / sec, 1),
           useful_lps_per_sec=round((tot["cuts"] + conf) / sec, 1), ms_lp=round(ms_lp, 1), ms_poly=round(ms_poly, 1),
           vertices=int(live.sum()), directions=int((d["pi"][live] != 0).sum()), facets=int(d["du"].sum()), edges=len(d["E"]),
           vertex_slots=len(live), rounds2=eng.poly_call("rounds2_stats"), vertices_sha256_at_1e6=sha)
eng.close()
print(json.dumps(row))
