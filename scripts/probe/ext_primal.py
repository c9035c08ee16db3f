"""Probe: force the primal clean-up on covering LPs (one-sided bounds only) with a large perturbation from the first pivot."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.lp import P2Model, LpEngine
for env in [{}, {"BSLV_LP_EXT": "1", "BSLV_STALL_LIMIT": "0", "BSLV_PERT_SCALE": "1e4"}, {"BSLV_LP_EXT": "1", "BSLV_STALL_LIMIT": "0", "BSLV_PERT_SCALE": "1e6"}]:
    for k in ("BSLV_LP_EXT", "BSLV_STALL_LIMIT", "BSLV_PERT_SCALE"): os.environ.pop(k, None)
    os.environ.update(env)
    print(env)
    for (m, n, q, seed, B) in [(60, 30, 3, 7, 64), (200, 100, 3, 1, 300)]:
        prob = synth.covering_vlp(m, n, q, seed)
        model = P2Model(prob)
        rng = np.random.default_rng(seed)
        X = rng.random((B, n)) * (3.0 / n) + 1.0 / n
        Y = X @ prob["P"].T
        V = Y * rng.uniform(0.2, 1.2, size=(B, 1)) + rng.normal(scale=0.05, size=Y.shape)
        ub = model.ub_for(V)
        eng = LpEngine.from_model(model, pool_slots=B + 1)
        eng.reset_slot(0)
        st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
        print((m, n, q), "cold", st, it, {k: v for k, v in eng.last_stats().items() if k in ("flip_iterations", "perturbations", "primal_steps", "wrong_sign_removals")})
        st, it = eng.solve_batch(np.zeros(B, np.int32), np.arange(1, B + 1, dtype=np.int32), np.full((B, model.r), -np.inf), ub)
        obj = eng.obj(np.arange(1, B + 1, dtype=np.int32))
        print("   batch", np.bincount(st, minlength=5), "pivots max", it.max(), "mean %.1f" % it.mean(), "obj sum %.12f" % obj.sum(),
              {k: v for k, v in eng.last_stats().items() if k in ("flip_iterations", "perturbations", "primal_steps", "wrong_sign_removals")})
        eng.close()
