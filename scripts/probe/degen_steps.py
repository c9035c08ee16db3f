"""Probe: first steps of a folded S-degenerate family member; prints pivots per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
from degen_fold import fold
m, n, q, steps, batch = [int(x) for x in sys.argv[1:6]]
prob = synth.degenerate_vlp(m, n, q, 3)      # (the driver's presolve folds the hypercube rows)
t0 = time.time()
eng = BensonEngine(prob, eps=1e-7, pool_slots=max(40, 2 * batch + 8))
print("start", eng.start(), eng.totals(), "%.1fs" % (time.time() - t0), flush=True)
for k in range(steps):
    nl, nt = eng.collect(batch, 0, 1)
    if nl == 0: break
    rec, piv, ls = eng.solve_local(nl)
    bad = [int(x) for x in rec[:, 1] if int(x) != 4]
    print("step", k, "lps", nl, "pivots", piv, "lockstep", ls, "not optimal", bad, "%.1fs" % (time.time() - t0), flush=True)
    if bad: break
    t1 = time.time()
    st = eng.apply(rec)
    print("     cuts", st["cuts"], "apply %.2fs" % (time.time() - t1), "paths", eng.poly_call("path_stats"), flush=True)
