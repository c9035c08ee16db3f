import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import poly_harness as ph
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
from bensolve_amd.poly import PolyEngine
which = sys.argv[1]
if which == "nan":
    prob = synth.covering_vlp(120, 60, 4, 11)
    eng = BensonEngine(prob, eps=1e-9, pool_slots=2048)
    eng.set_policy(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    assert eng.start() == 0
    for it in range(2000):
        s = eng.step(128)
        d = eng.poly_dump()
        live = d["pu"].astype(bool)
        bad = ~np.isfinite(d["X"][live]).all(axis=1)
        if bad.any():
            print("step", it, "non-finite live vertices:", bad.sum(), "of", live.sum(), eng.poly_call("rounds2_stats"), s)
            idx = np.nonzero(live)[0][bad][:5]
            print(idx, d["X"][idx])
            break
        if s["lps"] == 0 and s["left"] == 0:
            print("finished clean after", it, "steps", eng.poly_call("rounds2_stats"))
            break
elif which == "few":
    q, N, seed = 4, 300, 42
    D = np.vstack([ph.tangent_halfspaces(q, q + 3, seed), ph.tangent_halfspaces(q, N, seed + 100)])
    G = PolyEngine(q, 0, None)
    for i in range(q + 3):
        G.add(D[i], 0)
    assert G.init() == 0
    rc = G.add_cuts(D[q + 3:], None)
    print(G.rounds2_stats(), rc.sum())
