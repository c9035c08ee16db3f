"""Probe: P1(w) LPs (min w.y, Px - y = 0, Ax >= 1, x >= 0) through bslv_lpq_solve_batch_obj against scipy-HiGHS."""
import os, sys
import numpy as np
from scipy.optimize import linprog
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.lp import LpEngine
for (m, n, q, seed, B) in [(20, 10, 2, 11, 8), (60, 30, 3, 7, 40), (200, 100, 3, 1, 100)]:
    prob = synth.covering_vlp(m, n, q, seed)
    A, P = prob["A"], prob["P"]
    M, N = m + q, n + q
    L = np.zeros((M, N)); L[:m, :n] = A; L[m:, :n] = P; L[m:, n:] = -np.eye(q)
    lo = np.concatenate([np.ones(m), np.zeros(q), np.zeros(n), np.full(q, -np.inf)])
    up = np.concatenate([np.full(m, np.inf), np.zeros(q), np.full(n, np.inf), np.full(q, np.inf)])
    eng = LpEngine(M, N, L, lo, up, np.zeros(N + 1), 0, 0, B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
    print((m, n, q), "feasibility solve", st, it)
    rng = np.random.default_rng(seed)
    W = rng.random((B, q)) + 0.05
    W /= W.sum(axis=1, keepdims=True)
    st, it = eng.solve_batch_obj(np.zeros(B, np.int32), np.arange(1, B + 1, dtype=np.int32), M + n, W)
    obj = eng.obj(np.arange(1, B + 1, dtype=np.int32))
    y = eng.primal(np.arange(1, B + 1, dtype=np.int32), M + n, q)
    ref = np.array([linprog(W[b] @ P, A_ub=-A, b_ub=-np.ones(m), bounds=[(0, None)] * n, method="highs").fun for b in range(B)])
    print("   status", np.bincount(st, minlength=5), "pivots mean %.1f max %d" % (it.mean(), it.max()), "max |obj - highs| %.3e" % np.abs(obj - ref).max(),
          "max |w.y - obj| %.3e" % np.abs((W * y).sum(1) - obj).max(), {k: v for k, v in eng.last_stats().items() if k in ("primal_steps", "passes", "pivots")})
    # chained: new objectives from the solved slots, in place
    W2 = np.roll(W, 1, axis=0)
    st, it = eng.solve_batch_obj(np.arange(1, B + 1, dtype=np.int32), np.arange(1, B + 1, dtype=np.int32), M + n, W2)
    obj2 = eng.obj(np.arange(1, B + 1, dtype=np.int32))
    print("   in place, rolled objectives: status", np.bincount(st, minlength=5), "pivots mean %.1f" % it.mean(), "max |obj - highs| %.3e" % np.abs(obj2 - np.roll(ref, 1)).max())
    eng.close()
