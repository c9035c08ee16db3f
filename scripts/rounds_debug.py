import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import poly_harness as ph
from bensolve_amd.poly import PolyEngine
q, N, seed, k0 = 3, 600, 31, 6
D = ph.tangent_halfspaces(q, N, seed)
O = ph.FlatPoly("oracle", q)
rco = ph.run_sequence(O, D, init_after=k0)
for nb in (2, 3, 8, 64, 594):
    G = PolyEngine(q); G.set_batch_mode(1)
    for i in range(k0): G.add(D[i])
    G.init()
    rcg = []
    for s in range(k0, len(D), nb):
        rcg += list(G.add_cuts(D[s:s + nb]))
    bad = [i for i, (a, b) in enumerate(zip(rco[k0:], rcg)) if a != b]
    d = G.dump()
    print("chunk", nb, "rounds", G.rounds_run(), "bad rc", bad[:10], "live", d["pu"].sum(), "oracle live", O.dump()["pu"].sum())
    G.close()
