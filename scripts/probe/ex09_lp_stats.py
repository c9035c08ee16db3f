"""ex09: one cold P2 LP on the dense engine -- pivots, passes over the 1.36 GB tableau, time in k_flush"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bensolve_amd.synth import read_vlp
from bensolve_amd.lp import P2Model, LpEngine
prob = read_vlp("tests/golden/ex/ex09.vlp")
model = P2Model(prob)
eng = LpEngine.from_model(model, pool_slots=3)
eng.reset_slot(0)
v = np.full((1, prob["q"]), 1e3)
ub = model.ub_for(v)
t0 = time.time()
st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
print("status", st, "iters", it, "%.1f s" % (time.time() - t0), eng.last_stats(), flush=True)
v2 = v * 0.9
t0 = time.time()
st, it = eng.solve_batch([0], [1], np.full((1, model.r), -np.inf), model.ub_for(v2)[:1])
print("warm: status", st, "iters", it, "%.1f s" % (time.time() - t0), eng.last_stats(), flush=True)
eng.close()
