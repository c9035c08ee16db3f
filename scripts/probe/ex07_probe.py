"""Probe: ex07 (1211 x 1143, q = 3, sparse) on the dense engine with -b semantics: where do LPs stall?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
prob = synth.read_vlp(sys.argv[1])
print("rows", np.unique(prob["rtype"], return_counts=True), "cols", np.unique(prob["ctype"], return_counts=True), "nnz row min/max", (prob["A"] != 0).sum(1).min(), (prob["A"] != 0).sum(1).max(),
      "A values", np.unique(prob["A"])[:10], "P values", np.unique(prob["P"])[:10], flush=True)
t0 = time.time()
eng = BensonEngine(prob, eps=1e-7, pool_slots=600)
print("dims", eng.lp_dims(), flush=True)
print("start", eng.start(), eng.totals(), "%.1fs" % (time.time() - t0), flush=True)
for k in range(100000):
    nl, nt = eng.collect(128, 0, 1)
    if nl == 0: break
    if len(sys.argv) > 3 and int(sys.argv[2]) == k: os.environ["BSLV_LP_TRACE"] = sys.argv[3]
    rec, piv, ls = eng.solve_local(nl)
    os.environ.pop("BSLV_LP_TRACE", None)
    bad = [(i, int(x)) for i, x in enumerate(rec[:, 1]) if int(x) != 4]
    if bad or k % 50 == 0 or ls > 100:
        print("step", k, "lps", nl, "pivots", piv, "lockstep", ls, "not optimal", bad, eng.lp_call("last_stats"), "%.1fs" % (time.time() - t0), flush=True)
    if bad:
        its = rec[:, 0]
        break
    st = eng.apply(rec)
print("done", eng.totals(), "%.1fs" % (time.time() - t0))
