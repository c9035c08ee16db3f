"""Probe: find the first LP of the folded S-degenerate family member that fails / needs the most pivots, and trace it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from degen_fold import fold

m, n, q = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (1000, 500, 6)
def run(trace_at=None):
    prob = fold(synth.degenerate_vlp(m, n, q, 3))
    eng = BensonEngine(prob, eps=1e-7, pool_slots=40)
    assert eng.start() == 0
    for k in range(4):
        nl, nt = eng.collect(16, 0, 1)
        if trace_at and trace_at[0] == k: os.environ["BSLV_LP_TRACE"] = str(trace_at[1])
        rec, piv, ls = eng.solve_local(nl)
        os.environ.pop("BSLV_LP_TRACE", None)
        if trace_at and trace_at[0] == k: return None
        bad = [i for i in range(nl) if int(rec[i, 1]) != 4]
        print("step", k, "lps", nl, "pivots", piv, "lockstep", ls, "status", [int(x) for x in rec[:, 1]], flush=True)
        if bad:
            its = eng.lp_iters() if hasattr(eng, "lp_iters") else None
            return (k, bad[0] if bad else 0)
        eng.apply(rec)
    return None
t = run()
print("trace target", t, flush=True)
if t: run(t)
