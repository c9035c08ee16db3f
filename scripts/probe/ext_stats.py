"""Probe: which parts of the extended selection the boxed degenerate LP batches exercise."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.lp import P2Model, LpEngine
for (m, n, q, seed, B) in [(60, 30, 3, 3, 40), (240, 120, 4, 5, 96), (1000, 500, 6, 3, 128), (4000, 2000, 10, 3, 64)]:
    prob = synth.fold_singleton_rows(synth.degenerate_vlp(m, n, q, seed))
    model = P2Model(prob)
    rng = np.random.default_rng(seed)
    V = rng.random((B, n)) @ prob["P"].T + rng.normal(scale=0.5, size=(B, q))
    V[: B // 4] = np.round(V[: B // 4])
    ub = model.ub_for(V)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    print((m, n, q), "cold", st, it, eng.last_stats())
    st, it = eng.solve_batch(np.zeros(B, np.int32), np.arange(1, B + 1, dtype=np.int32), np.full((B, model.r), -np.inf), ub)
    print("   batch", np.bincount(st, minlength=5), "pivots max", it.max(), "mean %.1f" % it.mean(), eng.last_stats())
    eng.close()
print("forced clean-up: BSLV_STALL_LIMIT=1 BSLV_PERT_SCALE=1e4")
os.environ["BSLV_STALL_LIMIT"] = "1"; os.environ["BSLV_PERT_SCALE"] = "1e4"
for (m, n, q, seed, B) in [(60, 30, 3, 3, 40), (240, 120, 4, 5, 96), (1000, 500, 6, 3, 128)]:
    prob = synth.fold_singleton_rows(synth.degenerate_vlp(m, n, q, seed))
    model = P2Model(prob)
    rng = np.random.default_rng(seed)
    V = rng.random((B, n)) @ prob["P"].T + rng.normal(scale=0.5, size=(B, q))
    V[: B // 4] = np.round(V[: B // 4])
    ub = model.ub_for(V)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    print((m, n, q), "cold", st, it, eng.last_stats())
    st, it = eng.solve_batch(np.zeros(B, np.int32), np.arange(1, B + 1, dtype=np.int32), np.full((B, model.r), -np.inf), ub)
    print("   batch", np.bincount(st, minlength=5), "pivots max", it.max(), "mean %.1f" % it.mean(), eng.last_stats())
    eng.close()
