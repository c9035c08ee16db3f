"""Probe: wall time of the two variants of phase 2 on a complete run (covering problem, default cone, -b)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.vlp import solve_primal
m, n, q, seed, batch = [int(x) for x in sys.argv[1:6]]
prob = synth.covering_vlp(m, n, q, seed)
for alg in ("primal", "dual"):
    t0 = time.time()
    out = solve_primal(prob, bounded=True, batch=batch, alg_phase2=alg)
    d = out["dump"]
    print(alg, out["status"], "LPs", out["lps"], "steps", out["steps"], "upper image elements", int(d["pu"].sum()), "lower image vertices", int(d["du"].sum()), "%.2fs" % (time.time() - t0), flush=True)
