import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
prob = synth.CONFIGS["S-mid"]()
q = prob["q"]
eng = BensonEngine(prob, eps=1e-7, pool_slots=4160)
eng.start()
seen = set()
for it in range(14):
    nl, nt = eng.collect(1024)
    rec, piv, ls = eng.solve_local(nl)
    cuts = rec[rec[:, 2] == 1][:, 4:4 + q]
    key = [tuple(np.round(c / max(1.0, np.abs(c).max()), 10)) for c in cuts]
    uniq_in_batch = len(set(key))
    new = len(set(key) - seen)
    st = eng.apply(rec)
    seen |= set(key)
    print("step", it, "LPs", nl, "cut records", len(cuts), "unique in batch", uniq_in_batch, "not seen before", new, "applied", st["cuts"], "redundant", st["redundant"], flush=True)
