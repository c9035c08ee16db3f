"""CPU: the oracle LP (oracle/lp_dense.c) against independent HiGHS goldens and live scipy."""
import json
import os
import numpy as np
import pytest

import oracle_api
from bensolve_amd import synth
from bensolve_amd.lp import P2Model

HERE = os.path.dirname(os.path.abspath(__file__))


def test_p2_objectives_match_highs_goldens():
    rec = json.load(open(os.path.join(HERE, "golden", "lp_highs.json")))
    cache = {}
    for r in rec:
        key = (r["m"], r["n"], r["q"], r["seed"])
        if key not in cache:
            prob = synth.covering_vlp(*key)
            model = P2Model(prob)
            cache[key] = (model, oracle_api.OracleLP(model.L, model.lo, model.up, model.cost))
        model, olp = cache[key]
        ub = model.ub_for(np.array(r["v"])[None, :])[0]
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, ub[j])
        assert olp.solve(1) == 4
        assert abs(olp.obj() - r["obj"]) <= 1e-9 * (1 + abs(r["obj"]))
        # duals of the rows -Px + y = 0 are the weights w >= 0 with c.w = 1
        w = olp.dual(model.w_first, model.q)
        assert np.all(w >= -1e-9) and abs(w.sum() - 1) < 1e-9
    for model, olp in cache.values():
        olp.close()


@pytest.mark.parametrize("method", [0, 1])
def test_random_bounded_lps_match_scipy(method):
    from scipy.optimize import linprog
    rng = np.random.default_rng(0)
    checked = 0
    for trial in range(150):
        M, N = rng.integers(1, 10), rng.integers(1, 10)
        A = np.round(rng.normal(size=(M, N)) * 3) / 2
        A[rng.random((M, N)) < 0.3] = 0
        c = np.round(rng.normal(size=N) * 3)
        tr = rng.choice(list("fluds"), size=M, p=[.1, .3, .3, .2, .1])
        tc = rng.choice(list("fluds"), size=N, p=[.15, .4, .1, .25, .1])
        rl = np.round(rng.normal(size=M) * 2); ru = rl + rng.integers(0, 4, size=M)
        cl = np.round(rng.normal(size=N) * 2); cu = cl + rng.integers(0, 4, size=N)
        b = lambda t, l, u: {"f": (-np.inf, np.inf), "l": (l, np.inf), "u": (-np.inf, u), "d": (l, u), "s": (l, l)}[t]
        rb = [b(tr[i], rl[i], ru[i]) for i in range(M)]
        cb = [b(tc[j], cl[j], cu[j]) for j in range(N)]
        Aub, bub, Aeq, beq = [], [], [], []
        for i, (lo, up) in enumerate(rb):
            if lo == up:
                Aeq.append(A[i]); beq.append(lo)
            else:
                if np.isfinite(up): Aub.append(A[i]); bub.append(up)
                if np.isfinite(lo): Aub.append(-A[i]); bub.append(-lo)
        kw = dict(A_ub=np.array(Aub) if Aub else None, b_ub=bub if Aub else None, A_eq=np.array(Aeq) if Aeq else None,
                  b_eq=beq if Aeq else None, bounds=[(None if np.isinf(l) else l, None if np.isinf(u) else u) for l, u in cb])
        res = linprog(c, method="highs", **kw)
        if res.status == 2:      # HiGHS' presolve reports 'infeasible' for 'infeasible or unbounded'
            res = linprog(c, method="highs", options={"presolve": False}, **kw)
        lo = np.array([x[0] for x in rb] + [x[0] for x in cb]); up = np.array([x[1] for x in rb] + [x[1] for x in cb])
        olp = oracle_api.OracleLP(A, lo, up, np.concatenate([[0.0], c]))
        st = olp.solve(method)
        if res.status == 0:
            assert st == 4, (trial, st)
            assert abs(olp.obj() - res.fun) <= 1e-7 * (1 + abs(res.fun))
            checked += 1
        elif res.status == 2:
            assert st == 0, (trial, st)
        elif res.status == 3:
            assert st in (1, 0), (trial, st)
        olp.close()
    assert checked > 20
