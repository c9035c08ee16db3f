import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
import oracle_api, poly_harness as ph
m, n, q, seed, batch = 30, 15, 3, 5, 16
mode = sys.argv[1]
prob = synth.covering_vlp(m, n, q, seed)
eng = BensonEngine(prob, eps=1e-7, pool_slots=64)
eng.start()
O = ph.FlatPoly("oracle", q, 1, np.ones(q))
d0 = eng.poly_dump()
for k in range(1, len(d0["Y"])):
    if k == q + 1: pass
    O.add(d0["Y"][k], 0)
    if k == q: assert O.init() == 0
for it in range(200):
    nl, nt = eng.collect(batch)
    if nt == 0: break
    rec, piv, ls = eng.solve_local(nl)
    order = np.argsort(rec[:, 0])
    # mirror into the oracle poly: same order as apply()
    for k in order:
        if rec[k, 2]: O.add(rec[k, 4:4 + q], 0)
    if mode == "seq":
        # cannot split apply (slot bookkeeping) -> just apply; sequential variant handled by batch=1 run
        pass
    eng.apply(rec)
    dg, do = eng.poly_dump(), O.dump()
    lg, lo = dg["pu"].sum(), do["pu"].sum()
    # compare live coordinate sets (ignoring sltn marks)
    cg, co = ph.canonical(dg), ph.canonical(do)
    ok = cg["X"].shape == co["X"].shape and np.allclose(cg["X"], co["X"], atol=1e-9) and cg["E"] == co["E"]
    print("step", it, "batch", nt, "live gpu", lg, "oracle-replay", lo, "same", ok)
    if not ok: break
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/dbg_dump.npz", **{k: v for k, v in dg.items() if k != "d"})
