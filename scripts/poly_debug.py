import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.poly import PolyEngine
import oracle_api, poly_harness as ph
m, n, q, seed = 30, 15, 3, 5
prob = synth.covering_vlp(m, n, q, seed)
rc, fp, st = oracle_api.benson_phase2_primal(prob, eps=1e-7)
od = fp.dump()
Y = od["Y"][1:]
c = np.ones(q)
O = ph.FlatPoly("oracle", q, 1, c); G = PolyEngine(q, 1, c)
for k in range(len(Y)):
    if k == q:
        assert O.init() == 0 and G.init() == 0
    ro, rg = O.add(Y[k], 0), G.add(Y[k], 0)
    do, dg = O.dump(), G.dump()
    same = ro == rg and np.array_equal(do["pu"], dg["pu"]) and np.array_equal(do["E"], dg["E"]) and np.array_equal(do["I"], dg["I"])
    if not same:
        print("DIVERGE at cut", k, "rc", ro, rg, "nprimal", len(do["pu"]), len(dg["pu"]), "live", do["pu"].sum(), dg["pu"].sum(), "edges", len(do["E"]), len(dg["E"]))
        break
else:
    print("all", len(Y), "cuts identical; live", do["pu"].sum())
