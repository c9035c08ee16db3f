import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
from bensolve_amd.lp import P2Model
import oracle_api, poly_harness as ph
m, n, q, seed, batch = 30, 15, 3, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = synth.covering_vlp(m, n, q, seed)
model = P2Model(prob)
olp = oracle_api.OracleLP(model.L, model.lo, model.up, model.cost)
eng = BensonEngine(prob, eps=1e-9, pool_slots=64)
print("start", eng.start())
rc, fp, st = oracle_api.benson_phase2_primal(prob, eps=1e-9)
od = fp.dump(); XP = od["X"][(od["pu"] == 1) & (od["pi"] == 0)]
print("oracle P vertices", len(XP))
bad = 0
for it in range(200):
    nl, nt = eng.collect(batch)
    if nt == 0:
        break
    rec, piv, ls = eng.solve_local(nl)
    d = eng.poly_dump()
    for k in range(nl):
        slot = int(rec[k, 0]); v = d["X"][slot]
        ub = model.ub_for(v[None, :])[0]
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, ub[j])
        assert olp.solve(1) == 4
        if abs(olp.obj() - rec[k, 3]) > 1e-8:
            bad += 1
            print("step", it, "LP", k, "slot", slot, "v", v, "gpu z", rec[k, 3], "oracle z", olp.obj(), "status", rec[k, 1])
        w = np.concatenate([rec[k, 4:4 + q - 1], [1 - rec[k, 4:4 + q - 1].sum()]]); rhs = rec[k, 4 + q - 1]
        viol = (XP @ w - rhs).min()
        if rec[k, 2] and viol < -1e-7:
            print("step", it, "LP", k, "slot", slot, "INVALID CUT w", w, "rhs", rhs, "min slack", viol, "z", rec[k, 3], "oracle w", olp.dual(model.w_first, q), "y", olp.primal(model.y_first, q), "wy", olp.dual(model.w_first, q) @ olp.primal(model.y_first, q))
    st = eng.apply(rec)
print("steps", it, "bad LPs", bad)
d = eng.poly_dump()
print("live", d["pu"].sum(), "sltn", (d["ps"] & d["pu"]).sum())
