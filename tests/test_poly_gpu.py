"""GPU parity: HIP polyhedron engine (through the C ABI) vs the CPU oracle restatement.

The engine reproduces the oracle's slot and edge ORDER (prefix sums), so dumps are compared
slot by slot: flags, incidence pairs, edges and dual adjacency bit-exact, coordinates to 1e-12
(they come from the same fma chains).  The canonical (set-wise) comparison is also run."""
import itertools
import numpy as np
import pytest

import poly_harness as ph
from bensolve_amd.poly import PolyEngine

pytestmark = pytest.mark.gpu


def run_both(q, vals, ideals=None, init_after=None, apex=False, v2h=0, c=None, batched=False):
    O = ph.FlatPoly("oracle", q, v2h, c)
    G = PolyEngine(q, v2h, c)
    if apex:
        O.dual0_apex(); G.dual0_apex()
    rco = ph.run_sequence(O, vals, ideals, init_after)
    if batched:
        vals = np.asarray(vals, float)
        k = len(vals) if init_after is None else init_after
        rcg = [G.add(vals[i], 0 if ideals is None else ideals[i]) for i in range(k)]
        assert G.init() == 0
        if k < len(vals):
            rcg += list(G.add_cuts(vals[k:], None if ideals is None else np.asarray(ideals[k:])))
    else:
        rcg = ph.run_sequence(G, vals, ideals, init_after)
    assert list(rco) == list(rcg)
    O.dual_adjacency(); G.dual_adjacency()
    do, dg = O.dump(), G.dump()
    O.close(); G.close()
    return do, dg


def assert_slotwise_equal(do, dg):
    for k in ("pu", "pi", "du", "di"):
        assert np.array_equal(do[k], dg[k]), k
    assert np.array_equal(do["E"], dg["E"])
    assert np.array_equal(do["I"], dg["I"])
    assert np.array_equal(do["DE"], dg["DE"])
    live = do["pu"].astype(bool)
    np.testing.assert_allclose(do["X"][live], dg["X"][live], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(do["Y"], dg["Y"], rtol=0, atol=0)
    ph.assert_same(ph.canonical(do), ph.canonical(dg))


@pytest.mark.parametrize("q,N,seed", [(2, 30, 9), (3, 50, 1), (3, 2000, 2), (4, 200, 3), (5, 200, 4), (6, 60, 6), (8, 25, 8)])
def test_random_tangent_halfspaces(q, N, seed):
    D = ph.tangent_halfspaces(q, N, seed)
    do, dg = run_both(q, D, init_after=q + 3)
    assert_slotwise_equal(do, dg)


@pytest.mark.parametrize("q", [3, 4, 5, 6])
def test_degenerate_cube_and_crosspolytope(q):
    cube = np.vstack([np.eye(q), -np.eye(q)])
    signs = np.array(list(itertools.product([-1, 1], repeat=q)), float)
    for vals in (cube, np.vstack([cube, signs / (q - 2)]), np.vstack([cube, signs / q]), signs,
                 np.vstack([signs, cube]), np.vstack([signs, cube * 2.0])):
        do, dg = run_both(q, vals)
        assert_slotwise_equal(do, dg)


def test_batched_add_cuts_matches_sequential():
    q, N = 5, 300
    D = ph.tangent_halfspaces(q, N, 12)
    # duplicates and far-away (redundant) halfspaces exercise the batched incidence prefilter
    D = np.vstack([D, D[:20], D[:20] * 0.5])
    do, dg = run_both(q, D, init_after=q + 2, batched=True)
    assert_slotwise_equal(do, dg)


def test_cone_with_ideal_generators():
    # ex05-like ordering cone given by generators: cone_vertenum's call sequence (bslv_algs.c:331-350)
    gens = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0.5], [2, 0, -1], [0, 2, -1]], float)
    do, dg = run_both(3, gens, ideals=[1] * len(gens), apex=True)
    assert_slotwise_equal(do, dg)


def test_classify_batch_matches_numpy():
    q, N = 5, 400
    G = PolyEngine(q)
    D = ph.tangent_halfspaces(q, N, 5)
    ph.run_sequence(G, D, init_after=q + 1)
    d = G.dump()
    rng = np.random.default_rng(3)
    B = 70
    hps = np.hstack([rng.normal(size=(B, q)), -np.abs(rng.normal(size=(B, 1)))])
    # include exact on-plane cases: halfspaces through existing vertices
    live = np.nonzero(d["pu"] & (1 - d["pi"]))[0]
    for k in range(10):
        hps[k, q] = hps[k, :q] @ d["X"][live[k]]
    words, anym, _ = G.classify_batch(hps)
    X, used, ideal = d["X"], d["pu"].astype(bool), d["pi"].astype(bool)
    exp_any = np.zeros(B, int)
    for b in range(B):
        s = np.zeros(len(X))
        for k in range(q):                      # same fma order; numpy has no fma -> compare away from the bands
            s = s + hps[b, k] * X[:, k]
        a = np.where(ideal, 0.0, hps[b, q])
        cls = np.where(s > a + 1e-9, 3, np.where(s > a - 1e-9, 2, 1))
        cls = np.where(used, cls, 0)
        got = (words[b // 32] >> np.uint64(2 * (b % 32))) & np.uint64(3)
        safe = np.abs(np.abs(s - a) - 1e-9) > 1e-12
        assert np.array_equal(got[safe], cls[safe].astype(np.uint64)), b
        exp_any[b] = int(np.any(cls[safe] == 1))
        assert anym[b] >= exp_any[b]
    G.close()
