"""ex09: the same batch of warm-started P2(v) LPs through the tableau form and the revised form of the LP engine: statuses, optimal values, pivots, time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bensolve_amd.synth import read_vlp
from bensolve_amd.lp import P2Model, LpEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = read_vlp("tests/golden/ex/ex09.vlp")
model = P2Model(prob)
v0 = np.full((1, prob["q"]), 1e3)
rng = np.random.default_rng(9)
V = v0 * rng.uniform(0.6, 0.98, size=(B, prob["q"]))
res = {}
for rev in ("1", "0"):
    os.environ["BSLV_LP_REV"] = rev
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    t0 = time.time()
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), model.ub_for(v0)[:1])
    print("rev", rev, "cold: status", st, "pivots", it, "%.1f s" % (time.time() - t0), "obj", eng.obj([0]), flush=True)
    t0 = time.time()
    st, it = eng.solve_batch(np.zeros(B, np.int32), np.arange(1, B + 1, dtype=np.int32), np.full((B, model.r), -np.inf), model.ub_for(V))
    obj = eng.obj(np.arange(1, B + 1, dtype=np.int32))
    print("rev", rev, "batch: statuses", st.tolist(), "pivots", int(it.sum()), "%.1f s" % (time.time() - t0), flush=True)
    res[rev] = (st.copy(), obj.copy())
    eng.close()
ok = (res["0"][0] == 4) & (res["1"][0] == 4)
print("optimal in both:", int(ok.sum()), "of", B, "| max rel diff of the optimal values:", float(np.max(np.abs(res["0"][1][ok] - res["1"][1][ok]) / (1 + np.abs(res["0"][1][ok])))) if ok.any() else None)
print("objs rev:", res["1"][1][:8], "\nobjs tab:", res["0"][1][:8])
