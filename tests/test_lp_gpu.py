"""GPU parity: batched dual simplex (HIP, through the C ABI) vs the CPU oracle LP.

Objective values are unique -> compared at 1e-9 relative.  Dual/primal vectors are compared through
the size-independent LP identities every optimal solution satisfies (strong duality for P2(v):
z = w.(y - v), c.w = 1, w >= 0, and y - z c <= v componentwise for the default cone)."""
import numpy as np
import pytest

from bensolve_amd import synth
from bensolve_amd.lp import P2Model, LpEngine

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _random_V(model, prob, rng, B):
    """points v spread around the image of feasible points (any v gives a feasible, bounded P2(v))"""
    n = prob["n"]
    X = rng.random((B, n)) * (3.0 / n) + 1.0 / n
    Y = X @ prob["P"].T
    return Y * rng.uniform(0.2, 1.2, size=(B, 1)) + rng.normal(scale=0.05, size=Y.shape)


def _check_identities(model, V, obj, w, y):
    q = model.q
    assert np.all(w >= -1e-9)
    np.testing.assert_allclose(w.sum(axis=1), 1.0, rtol=0, atol=1e-9)          # c.w = 1, c = (1..1)
    np.testing.assert_allclose(np.einsum("bk,bk->b", w, y - V), obj, rtol=1e-8, atol=1e-9)
    assert np.all(y - obj[:, None] <= V + 1e-8)


@pytest.mark.parametrize("m,n,q,seed,B", [(20, 10, 2, 11, 7), (60, 30, 3, 7, 64), (200, 100, 3, 1, 300)])
def test_batch_matches_oracle(oracle, m, n, q, seed, B):
    import oracle_api
    prob = synth.covering_vlp(m, n, q, seed)
    model = P2Model(prob)
    rng = np.random.default_rng(seed)
    V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    # --- oracle: sequential warm-started solves (as the reference drives GLPK) ---
    olp = oracle_api.OracleLP(model.L, model.lo, model.up, model.cost)
    exp_obj = np.empty(B)
    for b in range(B):
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, ub[b, j])
        assert olp.solve(1) == 4
        exp_obj[b] = olp.obj()
    olp.close()
    # --- engine: cold start in slot 0, then the whole batch warm-started from slot 0 ---
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4, "cold start failed"
    np.testing.assert_allclose(eng.obj([0])[0], exp_obj[0], rtol=RTOL, atol=1e-9)
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
    assert np.all(st == 4)
    obj = eng.obj(dst)
    np.testing.assert_allclose(obj, exp_obj, rtol=RTOL, atol=1e-9)
    w = eng.dual(dst, model.w_first, q)
    y = eng.primal(dst, model.y_first, q)
    _check_identities(model, V, obj, w, y)
    # re-solving in place from the optimal slot needs no pivot and gives the same answer
    st2, it2 = eng.solve_batch(dst, dst, np.full((B, model.r), -np.inf), ub)
    assert np.all(st2 == 4) and np.all(it2 == 0)
    np.testing.assert_allclose(eng.obj(dst), obj, rtol=1e-12, atol=1e-12)
    eng.close()


def test_infeasible_and_unbounded_status():
    # x1 + x2 >= 2, x <= 0.5 each -> infeasible ; min -x with x free above -> unbounded
    A = np.array([[1.0, 1.0]])
    lo = np.array([2.0, 0.0, 0.0]); up = np.array([np.inf, 0.5, 0.5])
    eng = LpEngine(1, 2, A, lo, up, np.array([0.0, 1.0, 1.0]), 0, 0, 2)
    eng.reset_slot(0)
    st, _ = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
    assert st[0] == 0
    eng.close()
    lo = np.array([-np.inf, 0.0, 0.0]); up = np.array([np.inf, np.inf, np.inf])
    eng = LpEngine(1, 2, A, lo, up, np.array([0.0, -1.0, 0.0]), 0, 0, 2)
    eng.reset_slot(0)
    st, _ = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
    assert st[0] == 1
    eng.close()


def test_chains_of_warm_starts_through_unpivoted_slots(oracle):
    """The engine does not copy the parent's tableau when a solve starts: the first pivot streams it into the new slot,
    and a solve that ends without any pivot is copied afterwards.  Both kinds of slot must be complete tableaux: they
    are used here as parents of a second generation (and that one of a third), objectives against the oracle."""
    import oracle_api
    prob = synth.covering_vlp(60, 30, 3, 5)
    model = P2Model(prob)
    rng = np.random.default_rng(5)
    B = 48
    V = _random_V(model, prob, rng, B)
    V[B // 2:] = V[0]                       # half the batch repeats the bounds of the root solve: no pivot needed
    ub = model.ub_for(V)
    olp = oracle_api.OracleLP(model.L, model.lo, model.up, model.cost)

    def oracle_obj(u):
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, u[j])
        assert olp.solve(1) == 4
        return olp.obj()

    eng = LpEngine.from_model(model, pool_slots=3 * B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4
    gen1 = np.arange(1, B + 1, dtype=np.int32)
    st, it1 = eng.solve_batch(np.zeros(B, np.int32), gen1, np.full((B, model.r), -np.inf), ub)
    assert np.all(st == 4)
    assert np.count_nonzero(np.asarray(it1) == 0) >= B // 2, "the repeated bounds were meant to need no pivot"
    np.testing.assert_allclose(eng.obj(gen1), [oracle_obj(u) for u in ub], rtol=RTOL, atol=1e-9)
    # second generation: every first-generation slot (pivoted or not) is a parent, bounds shuffled
    perm = rng.permutation(B)
    gen2 = np.arange(B + 1, 2 * B + 1, dtype=np.int32)
    st, it2 = eng.solve_batch(gen1, gen2, np.full((B, model.r), -np.inf), ub[perm])
    assert np.all(st == 4)
    np.testing.assert_allclose(eng.obj(gen2), [oracle_obj(u) for u in ub[perm]], rtol=RTOL, atol=1e-9)
    # third generation from the second, original bounds again
    gen3 = np.arange(2 * B + 1, 3 * B + 1, dtype=np.int32)
    st, it3 = eng.solve_batch(gen2, gen3, np.full((B, model.r), -np.inf), ub)
    assert np.all(st == 4)
    np.testing.assert_allclose(eng.obj(gen3), [oracle_obj(u) for u in ub], rtol=RTOL, atol=1e-9)
    w = eng.dual(gen3, model.m, model.q)
    y = eng.primal(gen3, model.M + model.n, model.q)
    _check_identities(model, V, eng.obj(gen3), w, y)
    olp.close()


@pytest.mark.parametrize("m,n,q,seed,B", [(60, 30, 3, 3, 40), (240, 120, 4, 5, 96)])
def test_boxed_degenerate_batch_matches_oracle(oracle, m, n, q, seed, B):
    """Hypercube LPs of the S-degenerate family (columns boxed after the singleton-row presolve, integer data: ties in every
    ratio test).  Exercises the extended selection of the engine -- long-step ratio test, cost perturbation, clean-up --
    whose work the stats report; objective values against the oracle's dual+primal simplex."""
    import oracle_api
    prob = synth.fold_singleton_rows(synth.degenerate_vlp(m, n, q, seed))
    assert prob["m"] == m - n and np.all(prob["ctype"] == ord("d"))
    model = P2Model(prob)
    rng = np.random.default_rng(seed)
    X = rng.random((B, n))
    V = X @ prob["P"].T + rng.normal(scale=0.5, size=(B, q))
    V[: B // 4] = np.round(V[: B // 4])              # lattice points: the most degenerate right-hand sides
    ub = model.ub_for(V)
    olp = oracle_api.OracleLP(model.L, model.lo, model.up, model.cost)
    exp_obj = np.empty(B)
    for b in range(B):
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, ub[b, j])
        assert olp.solve(1) == 4
        exp_obj[b] = olp.obj()
    olp.close()
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4, "cold start failed"
    cold = eng.last_stats()
    assert cold["flip_iterations"] > 0, cold          # the cold start moves boxed columns to their other bound in bulk
    assert it[0] < 20 * (q + 2), (it, cold)           # ... instead of one degenerate pivot per column
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
    stats = eng.last_stats()
    assert np.all(st == 4), (st, stats)
    obj = eng.obj(dst)
    np.testing.assert_allclose(obj, exp_obj, rtol=RTOL, atol=1e-9)
    w = eng.dual(dst, model.w_first, q)
    y = eng.primal(dst, model.y_first, q)
    _check_identities(model, V, obj, w, y)
    # primal feasibility of the box and of the cover rows at the reported optimum
    x = eng.primal(dst, model.M, n)
    assert np.all(x >= -1e-9) and np.all(x <= 1 + 1e-9)
    assert np.all(x @ prob["A"].T >= 1 - 1e-8)
    np.testing.assert_allclose(x @ prob["P"].T, y, rtol=0, atol=1e-8)
    eng.close()


def test_s_degenerate_full_size_lps_are_certified():
    """BASELINE.json configs[4] at full size (m=4000, n=2000, q=10; 2021 x 2011 tableau after the presolve), too large for
    the CPU oracle: every reported optimum is certified by size-independent bounds -- the returned x is feasible and attains
    z (upper bound), and the returned weights w give the Lagrangian lower bound min_{x in box} w.(Px - v) in closed form;
    where the two meet the LP is solved to optimality, and they must never cross."""
    prob = synth.fold_singleton_rows(synth.CONFIGS["S-degenerate"]())
    m, n, q = prob["m"], prob["n"], prob["q"]
    assert (m, n, q) == (2000, 2000, 10)
    model = P2Model(prob)
    B = 24
    rng = np.random.default_rng(4)
    V = rng.random((B, n)) @ prob["P"].T + rng.normal(scale=2.0, size=(B, q))
    V[:6] = np.round(V[:6])
    ub = model.ub_for(V)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4 and it[0] < 500, (st, it, eng.last_stats())     # (without the long-step ratio test: > 3e5 pivots, no result)
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
    stats = eng.last_stats()
    assert np.all(st == 4), (st, stats)
    assert it.max() < 2000, (it, stats)
    obj = eng.obj(dst)
    w = eng.dual(dst, model.w_first, q)
    y = eng.primal(dst, model.y_first, q)
    x = eng.primal(dst, model.M, n)
    eng.close()
    _check_identities(model, V, obj, w, y)
    assert np.all(x >= -1e-9) and np.all(x <= 1 + 1e-9) and np.all(x @ prob["A"].T >= 1 - 1e-7)
    np.testing.assert_allclose(x @ prob["P"].T, y, rtol=0, atol=1e-7)
    wp = w @ prob["P"]                                   # (B, n): cost of x under the weights
    lower = np.minimum(wp, 0.0).sum(axis=1) - np.einsum("bk,bk->b", w, V)
    assert np.all(lower <= obj + 1e-7), (lower - obj).max()
    assert np.mean(np.abs(lower - obj) <= 1e-7) >= 0.9, np.abs(lower - obj)


@pytest.mark.parametrize("scale", ["1e4", "1e6"])
def test_forced_perturbation_and_primal_cleanup_reach_the_same_optimum(oracle, monkeypatch, scale):
    """The clean-up after a cost perturbation, forced: extended selection on for an LP without boxed variables, costs
    perturbed from the first pivot by 5e-3 .. 1 instead of 5e-7 .. 1e-6.  The perturbed optimum is then a different vertex,
    removing the perturbation leaves reduced costs of the wrong sign on one-sided variables, and primal simplex steps have to
    walk back to the true optimum: same objective values as the oracle, same LP identities."""
    import oracle_api
    m, n, q, seed, B = 200, 100, 3, 1, 200
    prob = synth.covering_vlp(m, n, q, seed)
    model = P2Model(prob)
    rng = np.random.default_rng(seed)
    V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    olp = oracle_api.OracleLP(model.L, model.lo, model.up, model.cost)
    exp_obj = np.empty(B)
    for b in range(B):
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, ub[b, j])
        assert olp.solve(1) == 4
        exp_obj[b] = olp.obj()
    olp.close()
    monkeypatch.setenv("BSLV_LP_EXT", "1")
    monkeypatch.setenv("BSLV_STALL_LIMIT", "0")
    monkeypatch.setenv("BSLV_PERT_SCALE", scale)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(np.zeros(B, np.int32), dst, np.full((B, model.r), -np.inf), ub)
    stats = eng.last_stats()
    assert np.all(st == 4), (st, stats)
    assert stats["perturbations"] > B // 2 and stats["wrong_sign_removals"] > 0 and stats["primal_steps"] > 0, stats
    obj = eng.obj(dst)
    np.testing.assert_allclose(obj, exp_obj, rtol=RTOL, atol=1e-9)
    _check_identities(model, V, obj, eng.dual(dst, model.w_first, q), eng.primal(dst, model.y_first, q))
    eng.close()


def test_wide_rows_read_the_pivot_rows_from_global_memory(oracle):
    """More than ~3000 columns: the six pending pivot rows of k_flush no longer fit in 144 KB of LDS and the kernel's wide
    instance reads them from global memory.  Same objective values as the oracle."""
    import oracle_api
    m, n, q, seed, B = 24, 3300, 2, 12, 6
    prob = synth.covering_vlp(m, n, q, seed)
    model = P2Model(prob)
    assert model.N * 8 * 6 > 144 * 1024
    rng = np.random.default_rng(seed)
    V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    olp = oracle_api.OracleLP(model.L, model.lo, model.up, model.cost)
    exp_obj = np.empty(B)
    for b in range(B):
        for j in range(model.r):
            olp.set_bound(model.var_first + j, -np.inf, ub[b, j])
        assert olp.solve(1) == 4
        exp_obj[b] = olp.obj()
    olp.close()
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4 and it[0] > 6
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(np.zeros(B, np.int32), dst, np.full((B, model.r), -np.inf), ub)
    assert np.all(st == 4)
    obj = eng.obj(dst)
    np.testing.assert_allclose(obj, exp_obj, rtol=RTOL, atol=1e-9)
    _check_identities(model, V, obj, eng.dual(dst, model.w_first, q), eng.primal(dst, model.y_first, q))
    eng.close()


def _p2_highs(prob, v):
    """P2(v) = min z : x in S, P x - z c <= v (c = (1..1)), stated for scipy's HiGHS directly from the problem data --
    independent of oracle/ and of bensolve_amd.lp.P2Model"""
    from scipy.optimize import linprog
    A, P = prob["A"], prob["P"]
    m, n = A.shape
    q = P.shape[0]
    Aub, bub, Aeq, beq = [], [], [], []
    for i in range(m):
        t, lo, up = chr(prob["rtype"][i]), prob["rlb"][i], prob["rub"][i]
        row = np.concatenate([A[i], [0.0]])
        if t == "s":
            Aeq.append(row); beq.append(lo)
        if t in "ld":
            Aub.append(-row); bub.append(-lo)
        if t in "ud":
            Aub.append(row); bub.append(up)
    for k in range(q):
        Aub.append(np.concatenate([P[k], [-1.0]])); bub.append(v[k])
    bounds = []
    for j in range(n):
        t, lo, up = chr(prob["ctype"][j]), prob["clb"][j], prob["cub"][j]
        bounds.append({"f": (None, None), "l": (lo, None), "u": (None, up), "d": (lo, up), "s": (lo, lo)}[t])
    bounds.append((None, None))
    c = np.zeros(n + 1); c[n] = 1.0
    res = linprog(c, A_ub=np.array(Aub), b_ub=np.array(bub), A_eq=np.array(Aeq) if Aeq else None, b_eq=np.array(beq) if Aeq else None,
                  bounds=bounds, method="highs")
    assert res.status == 0, res.message
    return res.fun


@pytest.mark.parametrize("name,B", [("S-small", 12), ("S-mid", 8), ("S-degenerate", 4)])
def test_gpu_lp_matches_highs_at_bench_sizes(name, B):
    """The LP half against an INDEPENDENT solver at the sizes of BASELINE.json's configurations (the CPU oracle is not in this
    test): P2(v) of S-small (200 x 100), S-mid (1000 x 500) and S-degenerate (4000 x 2000 as stated; the engine solves the
    presolved 2021 x 2011 form with boxed columns, long-step ratio test, perturbation) for B points v -- optimal values of the
    batched GPU dual simplex against scipy's HiGHS on the LP written down from the problem data, at HiGHS' own tolerance."""
    prob = synth.CONFIGS[name]()
    folded = synth.fold_singleton_rows(prob) if name == "S-degenerate" else prob
    model = P2Model(folded)
    rng = np.random.default_rng(11)
    n, q = prob["n"], prob["q"]
    if name == "S-degenerate":
        V = rng.random((B, n)) @ prob["P"].T + rng.normal(scale=2.0, size=(B, q))
    else:
        V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
    assert np.all(st == 4)
    obj = eng.obj(dst)
    eng.close()
    ref = np.array([_p2_highs(prob, V[b]) for b in range(B)])
    np.testing.assert_allclose(obj, ref, rtol=1e-7, atol=1e-7)


def test_extended_selection_by_switch_gives_the_same_optima():
    """bslv_lpq_set_extended (the third stage of the callers' retry, and the default for tableaux of 1 GiB: ex09): the LPs of a
    covering problem, which the plain dual simplex solves, solved again with cost perturbation / primal clean-up compiled in --
    same optimal values at 1e-9, the LP identities hold; and the switch is what made the difference (perturbations counted)."""
    prob = synth.covering_vlp(200, 100, 3, 1)
    model = P2Model(prob)
    rng = np.random.default_rng(5)
    B = 64
    V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    res = {}
    for ext in (0, 1):
        eng = LpEngine.from_model(model, pool_slots=B + 1)
        eng.set_extended(ext)
        eng.reset_slot(0)
        st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
        assert st[0] == 4
        src = np.zeros(B, np.int32); dst = np.arange(1, B + 1, dtype=np.int32)
        st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
        assert np.all(st == 4)
        res[ext] = (eng.obj(dst), eng.dual(dst, model.w_first, model.q), eng.primal(dst, model.y_first, model.q))
        eng.close()
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-9, atol=1e-9)
    _check_identities(model, V, *res[1])


def test_lp_layer_presolve_solves_s_degenerate_as_stated():
    """the LP of BASELINE configs[4] AS STATED -- 4021 x 2011, the hypercube written as 2000 rows `d 0 1` over free
    columns (ex/example10.m:21-24) -- went through the LP engine with status UNDEFINED after 3e5 pivots; the presolve that made
    it solvable sat in the Benson driver.  It now sits behind bslv_lpq_create: the model is handed over as the reference would
    hand it to GLPK (bslv_lp.c:60-70), without synth.fold_singleton_rows, and the optimal values equal HiGHS'."""
    import ctypes
    prob = synth.CONFIGS["S-degenerate"]()
    model = P2Model(prob)                                  # every row of A in the model
    assert model.M == prob["m"] + 2 * prob["q"] + 1
    rng = np.random.default_rng(11)
    n, q, B = prob["n"], prob["q"], 4
    V = rng.random((B, n)) @ prob["P"].T + rng.normal(scale=2.0, size=(B, q))
    ub = model.ub_for(V)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    eng.lib.bslv_lpq_rows_folded.argtypes = [ctypes.c_void_p]
    assert eng.lib.bslv_lpq_rows_folded(eng.h) == 2000
    eng.reset_slot(0)
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
    assert st[0] == 4 and it[0] < 500, (st, it)
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)
    st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
    assert np.all(st == 4)
    obj = eng.obj(dst)
    # the folded rows still answer: their primal value is the column's (a = 1), within the row's bounds
    rows = eng.primal(dst, 0, 2000)
    cols = eng.primal(dst, model.M, 2000)
    eng.close()
    np.testing.assert_allclose(rows, cols, rtol=0, atol=0)
    assert rows.min() >= -1e-9 and rows.max() <= 1 + 1e-9
    ref = np.array([_p2_highs(prob, V[b]) for b in range(B)])
    np.testing.assert_allclose(obj, ref, rtol=1e-7, atol=1e-7)


def test_presolve_keeps_the_model_of_the_caller_primal_and_dual():
    """Rows with one non-zero are folded into column bounds inside bslv_lpq_create; what crosses the boundary stays the model as
    given.  Random LPs with a third of the rows of that kind, solved with the presolve and (BSLV_NO_PRESOLVE) without: same optimal
    values, and in both cases the returned (x, row activities, row duals, reduced costs) satisfy the optimality conditions of the
    model AS GIVEN -- r = A x, d = c - A' lambda, every non-zero dual sits on a bound of its variable with the sign of a minimum."""
    import ctypes
    import os
    rng = np.random.default_rng(77)
    for trial in range(6):
        M, N = 24, 16
        A = rng.integers(-3, 4, size=(M, N)).astype(float) * (rng.random((M, N)) < 0.5)
        single = rng.choice(M, size=8, replace=False)
        for i in single:
            A[i] = 0.0
            A[i, rng.integers(N)] = rng.choice([-2.0, -1.0, 1.0, 3.0])
        x0 = rng.normal(size=N)
        r0 = A @ x0
        lo = np.concatenate([r0 - rng.random(M) * 2, x0 - 1 - rng.random(N)])
        up = np.concatenate([r0 + rng.random(M) * 2, x0 + 1 + rng.random(N)])
        lo[rng.choice(M, 6, replace=False)] = -np.inf
        up[rng.choice(M, 6, replace=False)] = np.inf
        cost = np.concatenate([[0.0], rng.normal(size=N)])
        out = {}
        for mode in ("presolve", "plain"):
            if mode == "plain":
                os.environ["BSLV_NO_PRESOLVE"] = "1"
            try:
                eng = LpEngine(M, N, A, lo, up, cost, 0, 0, 4)
            finally:
                os.environ.pop("BSLV_NO_PRESOLVE", None)
            eng.lib.bslv_lpq_rows_folded.argtypes = [ctypes.c_void_p]
            nf = eng.lib.bslv_lpq_rows_folded(eng.h)
            assert (nf > 0) == (mode == "presolve"), (mode, nf)
            eng.reset_slot(0)
            st, it = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
            assert st[0] == 4, (trial, mode, st)
            z = eng.obj([0])[0]
            prim = eng.primal([0], 0, M + N)[0]
            dual = eng.dual([0], 0, M + N)[0]
            eng.close()
            x, r, lam, d = prim[M:], prim[:M], dual[:M], dual[M:]
            np.testing.assert_allclose(r, A @ x, atol=1e-9)
            np.testing.assert_allclose(z, cost[0] + cost[1:] @ x, atol=1e-9)
            np.testing.assert_allclose(d, cost[1:] - A.T @ lam, atol=1e-8)
            assert np.all(prim >= lo - 1e-8) and np.all(prim <= up + 1e-8)
            for k in range(M + N):
                if abs(dual[k]) > 1e-9:                       # a variable with a non-zero dual sits on the bound of the matching sign
                    at_lo, at_up = abs(prim[k] - lo[k]) < 1e-7, abs(prim[k] - up[k]) < 1e-7
                    assert (dual[k] > 0 and at_lo) or (dual[k] < 0 and at_up) or (at_lo and at_up), (trial, mode, k, dual[k], prim[k], lo[k], up[k])
            out[mode] = z
        assert abs(out["presolve"] - out["plain"]) < 1e-9 * (1 + abs(out["plain"]))


@pytest.mark.parametrize("kind", ["covering", "boxed"])
def test_one_selection_launch_per_pass_is_bit_identical_to_six(monkeypatch, kind):
    """k_select loops over the KP = 6 selections between two passes over the tableau inside ONE launch (round 4); BSLV_SELECT_FUSE=0
    launches it once per selection as rounds 1-3 did.  Same pivots, same objective values, primal values and duals bit for bit
    (plain dual simplex on covering LPs; the extended selection -- long steps, perturbation, primal clean-up -- on boxed ones)."""
    import oracle_api  # noqa: F401  (same import order as the other tests)
    rng = np.random.default_rng(12)
    if kind == "covering":
        prob = synth.covering_vlp(200, 100, 3, 1)
        B = 96
        model = P2Model(prob)
        V = _random_V(model, prob, rng, B)
    else:
        prob = synth.fold_singleton_rows(synth.degenerate_vlp(240, 120, 4, 5))
        B = 48
        model = P2Model(prob)
        V = rng.random((B, 120)) @ prob["P"].T + rng.normal(scale=0.5, size=(B, 4))
        V[: B // 4] = np.round(V[: B // 4])
    ub = model.ub_for(V)
    res = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("BSLV_SELECT_FUSE", fuse)
        eng = LpEngine.from_model(model, pool_slots=B + 1)
        eng.reset_slot(0)
        st0, it0 = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
        assert st0[0] == 4
        src = np.zeros(B, np.int32)
        dst = np.arange(1, B + 1, dtype=np.int32)
        st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
        assert np.all(st == 4)
        res[fuse] = (int(it0[0]), it.copy(), eng.obj(dst).copy(), eng.dual(dst, model.w_first, model.q).copy(), eng.primal(dst, model.y_first, model.q).copy(),
                     eng.primal(dst, model.M, prob["n"]).copy())
        eng.close()
    a, b = res["0"], res["1"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]), "pivot counts differ"
    assert a[1].sum() > B, "the batch needs pivots for this to mean anything"
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64))


@pytest.mark.parametrize("env", [{"BSLV_SELECT_NT": "1024"}, {"BSLV_FLUSH_NT": "1024"}, {"BSLV_UPD_GRID": "64"}, {"BSLV_LP_LAZY": "0"}],
                         ids=lambda e: " ".join("%s=%s" % kv for kv in e.items()))
def test_launch_shapes_do_not_change_the_lps(monkeypatch, env):
    """The LP engine chooses its launch shapes by size: 256 or 1024 threads per selection workgroup (rows of 1536 columns and more) and
    per tableau-pass workgroup (pivot rows beyond 53 KB of LDS), a persistent grid for the passes, lazy or eager passes.  None of them may
    change a pivot or a bit of a result: every reduction is an argmax / min with index tie-breaks or a per-entry fma chain, nothing is
    summed across threads.  The switches force the other shape on a problem that would not take it by itself (tuning arms: covered so that
    they cannot go stale unnoticed, as the local-minima arm of the rounds did in round 4)."""
    import oracle_api  # noqa: F401  (same import order as the other tests)
    rng = np.random.default_rng(12)
    prob = synth.covering_vlp(200, 100, 3, 1)
    B = 96
    model = P2Model(prob)
    V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    res = []
    for e in ({}, env):
        for k in ("BSLV_SELECT_NT", "BSLV_FLUSH_NT", "BSLV_UPD_GRID", "BSLV_LP_LAZY"):
            monkeypatch.delenv(k, raising=False)
        for k, v in e.items():
            monkeypatch.setenv(k, v)
        eng = LpEngine.from_model(model, pool_slots=B + 1)
        eng.reset_slot(0)
        st0, it0 = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
        assert st0[0] == 4
        src = np.zeros(B, np.int32)
        dst = np.arange(1, B + 1, dtype=np.int32)
        st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
        assert np.all(st == 4)
        res.append((int(it0[0]), it.copy(), eng.obj(dst).copy(), eng.dual(dst, model.w_first, model.q).copy(), eng.primal(dst, model.y_first, model.q).copy()))
        eng.close()
    a, b = res
    assert a[0] == b[0] and np.array_equal(a[1], b[1]), "pivot counts differ"
    assert a[1].sum() > B
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64))


def test_bounds_that_leave_a_folded_row_no_room_are_reported_infeasible():
    """The presolve folds a row with one non-zero into its column's bounds once, at create time.  New bounds (bslv_lpq_set_bounds:
    lp_set_rows / lp_set_cols of the reference, bslv_lp.c:112-135) that make the folded row and its column contradict each other
    leave no row behind that could say so: the engine must report INFEASIBLE itself, and recover when the bounds are relaxed."""
    # min x0 + x1  s.t.  row0: x0 + x1 >= 1,  row1: 2 x0 in [0, 4] (one non-zero: folded),  x >= 0
    A = np.array([[1.0, 1.0], [2.0, 0.0]])
    lo = np.array([1.0, 0.0, 0.0, 0.0]); up = np.array([np.inf, 4.0, np.inf, np.inf])
    eng = LpEngine(2, 2, A, lo, up, np.array([0.0, 1.0, 1.0]), 0, 0, 4)
    assert eng.rows_folded() == 1
    eng.reset_slot(0)
    st, _ = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
    assert st[0] == 4 and abs(eng.obj([0])[0] - 1.0) < 1e-12
    # row1 now demands 2 x0 >= 6 while the column says x0 <= 2: infeasible in the model as given
    lo2, up2 = lo.copy(), up.copy()
    lo2[1], up2[1], up2[2] = 6.0, 8.0, 2.0
    eng.set_bounds(lo2, up2)
    eng.reset_slot(0)
    st, _ = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
    assert st[0] == 0, st
    # relaxed again (x0 <= 5): feasible, x0 = 3 forced by the folded row, objective 3
    up2[2] = 5.0
    eng.set_bounds(lo2, up2)
    eng.reset_slot(0)
    st, _ = eng.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
    assert st[0] == 4 and abs(eng.obj([0])[0] - 3.0) < 1e-9, (st, eng.obj([0]))
    eng.close()


def test_a_row_on_a_column_with_per_lp_bounds_is_not_folded():
    """A single-variable row whose column gets its bounds PER LP (the var range of solve_batch lies on columns) must stay a row:
    folded into the column's bounds it would be overwritten by the bounds of every LP."""
    # min -x0  s.t.  row0: x0 <= 1 (one non-zero),  row1: x0 + x1 >= 0;  per-LP bounds on column x0
    A = np.array([[1.0, 0.0], [1.0, 1.0]])
    lo = np.array([-np.inf, 0.0, 0.0, 0.0]); up = np.array([1.0, np.inf, 10.0, np.inf])
    eng = LpEngine(2, 2, A, lo, up, np.array([0.0, -1.0, 0.0]), 2, 1, 4)      # var range = column 0 (variable id M + 0 = 2)
    assert eng.rows_folded() == 0
    eng.reset_slot(0)
    st, _ = eng.solve_batch([0], [0], np.array([[0.0]]), np.array([[5.0]]))       # cold start, then the batch from its basis
    assert st[0] == 4, st
    st, _ = eng.solve_batch([0, 0], [1, 2], np.array([[0.0], [0.0]]), np.array([[5.0], [0.5]]))
    assert np.all(st == 4), st
    np.testing.assert_allclose(eng.obj([1, 2]), [-1.0, -0.5], rtol=0, atol=1e-12)     # the row x0 <= 1 holds although LP 1 says x0 <= 5
    eng.close()


def _sparse_covering(m, n, q, seed, per_col=4, dense_cols=0):
    """covering VLP with a sparse A (per_col non-zeros per column, every row hit) and sparse objectives: the shape of ex07 / ex09
    (dense_cols: that many columns with m / 2 non-zeros, as ex09's 75 columns of 512 and 1024)"""
    rng = np.random.default_rng(seed)
    prob = synth.covering_vlp(m, n, q, seed)
    A = np.zeros((m, n))
    for j in range(n):
        k = m // 2 if j < dense_cols else per_col
        rows = rng.choice(m, size=k, replace=False)
        A[rows, j] = rng.uniform(0.5, 1.5, size=k) * (0.2 if j < dense_cols else 1.0)
    for i in range(m):
        if not A[i].any():
            A[i, rng.integers(n)] = 1.0
    P = prob["P"] * (rng.random((q, n)) < 0.3)
    P[:, 0] = prob["P"][:, 0]
    prob = dict(prob, A=A, P=P)
    return prob


@pytest.mark.parametrize("m,n,q,seed,B,dense_cols", [(40, 300, 3, 5, 24, 0), (90, 700, 4, 9, 48, 0), (120, 6000, 3, 11, 8, 6)])
def test_revised_form_equals_the_tableau_form(monkeypatch, m, n, q, seed, B, dense_cols):
    """SURVEY 8f rank 4 (the reference hands A to the solver as COO, bslv_lp.c:60-70): for wide sparse problems the engine keeps the
    basis inverse per LP instead of the tableau and A once as CSC / CSR (BSLV_LP_REV=1 forces that form, 0 the tableau).  Same LPs
    through both: statuses, optimal values (1e-9), primal values of y and of every x, duals w, the LP identities; cold start and a
    warm-started batch; chains of warm starts (a child of a child) included.  The third case has rows of more than 4096 columns and six
    columns of 60 non-zeros: the selection runs with 1024 threads, the sparse products of a tableau row are dealt in slices to helper
    workgroups of the same launch (rev_helper) and the dense columns are summed by whole waves (rev_row_slice)."""
    prob = _sparse_covering(m, n, q, seed, dense_cols=dense_cols)
    model = P2Model(prob)
    rng = np.random.default_rng(seed)
    V = _random_V(model, prob, rng, B)
    ub = model.ub_for(V)
    res = {}
    for rev in ("0", "1"):
        monkeypatch.setenv("BSLV_LP_REV", rev)
        eng = LpEngine.from_model(model, pool_slots=2 * B + 1)
        eng.reset_slot(0)
        st0, it0 = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
        assert st0[0] == 4, (rev, st0)
        src = np.zeros(B, np.int32)
        dst = np.arange(1, B + 1, dtype=np.int32)
        st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
        assert np.all(st == 4), (rev, st)
        # second generation: every LP again with a shifted right-hand side, from its own optimal slot into a new one
        ub2 = model.ub_for(V * 1.07 + 0.01)
        dst2 = np.arange(B + 1, 2 * B + 1, dtype=np.int32)
        st2, it2 = eng.solve_batch(dst, dst2, np.full((B, model.r), -np.inf), ub2)
        assert np.all(st2 == 4), (rev, st2)
        res[rev] = dict(obj=eng.obj(dst).copy(), w=eng.dual(dst, model.w_first, q).copy(), y=eng.primal(dst, model.y_first, q).copy(), x=eng.primal(dst, model.M, n).copy(),
                        obj2=eng.obj(dst2).copy(), w2=eng.dual(dst2, model.w_first, q).copy(), y2=eng.primal(dst2, model.y_first, q).copy(), it=int(it.sum()), it2=int(it2.sum()))
        eng.close()
    a, b = res["0"], res["1"]
    assert b["it"] > 0 and b["it2"] > 0
    np.testing.assert_allclose(b["obj"], a["obj"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(b["obj2"], a["obj2"], rtol=RTOL, atol=1e-9)
    _check_identities(model, V, b["obj"], b["w"], b["y"])
    _check_identities(model, V * 1.07 + 0.01, b["obj2"], b["w2"], b["y2"])
    # feasibility of the revised form's x at its optimum
    assert np.all(b["x"] >= -1e-9) and np.all(b["x"] @ prob["A"].T >= 1 - 1e-8)
    np.testing.assert_allclose(b["x"] @ prob["P"].T, b["y"], rtol=0, atol=1e-8)


def test_lazy_tableaux_give_the_same_lps_and_the_same_children():
    """bslv_lpq_set_lazy: an LP that is finished when its tableau pass would be due keeps its pending pivots -- objective, primal values
    and duals come from its vectors -- and only the slots the caller names (bslv_lpq_materialise) get their tableau.  Against the eager
    engine: the same results bit for bit for the whole batch, most passes skipped, and a second generation of LPs started from the
    materialised slots (every third LP of the first) equal bit for bit as well -- also when the next batch arrives while slots are
    still open (they are given their tableau first)."""
    prob = synth.covering_vlp(200, 100, 3, 1)
    model = P2Model(prob)
    B = 96
    rng = np.random.default_rng(4)
    V = _random_V(model, prob, rng, B)
    ub, ub2 = model.ub_for(V), model.ub_for(V * 1.05 + 0.02)
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)
    keep = dst[::3].copy()
    dst2 = np.arange(B + 1, B + 1 + len(keep), dtype=np.int32)
    res = {}
    for mode in ("eager", "lazy", "lazy, next batch while open"):
        eng = LpEngine.from_model(model, pool_slots=2 * B + 2)
        eng.reset_slot(0)
        st0, _ = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), ub[:1])
        assert st0[0] == 4
        eng.set_lazy(mode != "eager")
        st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), ub)
        assert np.all(st == 4)
        first = [eng.obj(dst).copy(), eng.dual(dst, model.w_first, model.q).copy(), eng.primal(dst, model.y_first, model.q).copy(), eng.primal(dst, model.M, prob["n"]).copy(), it.copy()]
        if mode == "lazy":
            eng.materialise(keep)
            eng.discard_pending()
        st2, it2 = eng.solve_batch(keep, dst2, np.full((len(keep), model.r), -np.inf), ub2[::3])
        assert np.all(st2 == 4), (mode, st2)
        second = [eng.obj(dst2).copy(), eng.dual(dst2, model.w_first, model.q).copy(), eng.primal(dst2, model.y_first, model.q).copy(), it2.copy()]
        res[mode] = (first, second, eng.lazy_stats())
        eng.close()
    for mode in ("lazy", "lazy, next batch while open"):
        for a, b in zip(res["eager"][0] + res["eager"][1], res[mode][0] + res[mode][1]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), mode
    assert res["eager"][2]["skipped"] == 0
    assert res["lazy"][2]["skipped"] > B // 2 and 0 < res["lazy"][2]["on_request"] <= len(keep) + len(keep), res["lazy"][2]
    assert res["lazy, next batch while open"][2]["on_request"] >= res["lazy"][2]["skipped"], res
