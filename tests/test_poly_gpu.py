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


def run_both(q, vals, ideals=None, init_after=None, apex=False, v2h=0, c=None, batched=False, mode=0):
    O = ph.FlatPoly("oracle", q, v2h, c)
    G = PolyEngine(q, v2h, c)
    G.set_batch_mode(mode)
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


@pytest.mark.parametrize("mode,when", [(0, "before init"), (1, "before init"), (1, "between batches")])
def test_capacity_reserved_ahead_changes_nothing(mode, when):
    """bslv_poly_reserve: element, edge and pool arrays at their final size before the cuts arrive (no doubling in the middle of a batch)
    give the polyhedron of the engine that grew as it went, slot for slot -- and the oracle's (the reference grows in blocks of VRTXBLCK /
    LSTBLCK, bslv_poly.c:415-440, :452: capacity is not part of the result).  Refused with an argument error for negative sizes."""
    q, N = 4, 400
    D = ph.tangent_halfspaces(q, N, 21)
    k = q + 2
    dumps = []
    for reserve in (False, True):
        G = PolyEngine(q, 0, None)
        G.set_batch_mode(mode)
        if reserve and when == "before init":
            G.reserve(1 << 16, 1 << 18, 1 << 21)
        rc = [G.add(D[i], 0) for i in range(k)]
        assert G.init() == 0
        half = k + (N - k) // 2
        rc += list(G.add_cuts(D[k:half], None))
        if reserve and when == "between batches":
            G.reserve(1 << 16, 1 << 18, 1 << 21)
            G.reserve(0, 0, 0)                       # zeros leave everything alone
            with pytest.raises(Exception):
                G.reserve(-1, 0, 0)
        rc += list(G.add_cuts(D[half:], None))
        G.dual_adjacency()
        dumps.append((rc, G.dump()))
        G.close()
    assert dumps[0][0] == dumps[1][0]
    assert_slotwise_equal(dumps[0][1], dumps[1][1])
    if mode == 0:
        O = ph.FlatPoly("oracle", q, 0, None)
        rco = ph.run_sequence(O, D, None, k)
        O.dual_adjacency()
        assert list(rco) == list(dumps[1][0])
        assert_slotwise_equal(O.dump(), dumps[1][1])
        O.close()


def test_batched_add_cuts_matches_sequential():
    q, N = 5, 300
    D = ph.tangent_halfspaces(q, N, 12)
    # duplicates and far-away (redundant) halfspaces exercise the batched incidence prefilter
    D = np.vstack([D, D[:20], D[:20] * 0.5])
    do, dg = run_both(q, D, init_after=q + 2, batched=True, mode=0)
    assert_slotwise_equal(do, dg)


@pytest.mark.parametrize("q,N,seed,k0", [(3, 600, 31, 6), (4, 300, 32, 8), (5, 400, 33, 7), (6, 80, 34, 9)])
def test_rounds_of_independent_cuts_match_sequential_sets(q, N, seed, k0):
    """multi-cut path: independent cuts of a batch applied in one pass per round.  Slot numbers differ
    from the sequential order; vertex / facet / incidence / adjacency sets and the redundancy verdicts
    must not."""
    D = ph.tangent_halfspaces(q, N, seed)
    D = np.vstack([D, D[:15], D[5:25] * 0.5])
    O = ph.FlatPoly("oracle", q)
    rco = ph.run_sequence(O, D, init_after=k0)
    G = PolyEngine(q)
    G.set_batch_mode(1)
    rcg = [G.add(D[i]) for i in range(k0)]
    assert G.init() == 0
    rcg += list(G.add_cuts(D[k0:]))
    # the verdict "redundant" of an individual cut depends on the order among conflicting cuts (a far
    # parallel copy applied before the near plane is cut away later instead of being rejected); the
    # exact duplicates are rejected in any order, and the final sets below do not depend on the order
    assert sum(rcg) >= 15 and sum(rcg) <= sum(rco)
    assert G.rounds_run() < len(D) - k0          # cuts really were grouped
    O.dual_adjacency(); G.dual_adjacency()
    ph.assert_same(ph.canonical(O.dump()), ph.canonical(G.dump()), rtol=1e-12, atol=1e-12)
    O.close(); G.close()


@pytest.mark.parametrize("q", [3, 4, 5])
def test_rounds_on_degenerate_inputs(q):
    cube = np.vstack([np.eye(q), -np.eye(q)])
    signs = np.array(list(itertools.product([-1, 1], repeat=q)), float)
    for vals in (np.vstack([cube, signs / (q - 2)]), np.vstack([cube, signs / q]), np.vstack([signs, cube * 2.0])):
        k0 = 2 * q + 2
        O = ph.FlatPoly("oracle", q)
        rco = ph.run_sequence(O, vals, init_after=k0)
        G = PolyEngine(q)
        rcg = [G.add(vals[i]) for i in range(k0)]
        assert G.init() == 0
        rcg += list(G.add_cuts(vals[k0:]))
        O.dual_adjacency(); G.dual_adjacency()
        ph.assert_same(ph.canonical(O.dump()), ph.canonical(G.dump()))
        O.close(); G.close()


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


# ---- against golden outputs of the REFERENCE polyhedron engine (tests/golden/poly_ref.npz) ----
import os as _os
_GOLD = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "poly_ref.npz"))
_NAMES = sorted({k.split("/")[0] for k in _GOLD.files})


@pytest.mark.parametrize("name", _NAMES)
def test_gpu_matches_reference_golden(name):
    q, v2h, apex, init_after = [int(x) for x in _GOLD[name + "/in_meta"]]
    G = PolyEngine(q, v2h)
    if apex:
        G.dual0_apex()
    rcs = ph.run_sequence(G, _GOLD[name + "/in_vals"], list(_GOLD[name + "/in_ideals"]), None if init_after < 0 else init_after)
    G.dual_adjacency()
    can = ph.canonical(G.dump())
    G.close()
    g = lambda k: _GOLD[name + "/" + k]
    assert list(rcs) == list(g("rc"))
    exp = dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")},
               DE={tuple(e) for e in g("DE")})
    ph.assert_same(can, exp)


import json as _json
_GOLD_L = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "poly_ref_large.npz"))
_META_L = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "poly_ref_large.json")))


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("name", sorted(_META_L))
def test_gpu_matches_reference_golden_at_survey_sizes(name, batched):
    """the HIP engine against the reference's own results at SURVEY.md 8c's sizes (27 418 vertices at q=5 N=1000, 28 681 at
    q=8 N=60, ...): coordinates at 1e-9, adjacency / incidence / dual adjacency bit-exact (SHA-256 of the canonical sets).
    batched: the cuts after the initial simplex go through bslv_poly_add_cuts (multi-cut rounds / hot chunks) instead of one
    bslv_poly_add each."""
    q, v2h, apex, init_after = [int(x) for x in _GOLD_L[name + "/in_meta"]]
    vals, ideals = _GOLD_L[name + "/in_vals"], list(_GOLD_L[name + "/in_ideals"])
    G = PolyEngine(q, v2h)
    if apex:
        G.dual0_apex()
    if not batched:
        rcs = ph.run_sequence(G, vals, ideals, None if init_after < 0 else init_after)
        assert list(rcs) == list(_GOLD_L[name + "/rc"])
    else:
        k0 = len(vals) if init_after < 0 else init_after
        for k in range(k0):
            G.add(vals[k], ideals[k])
        assert G.init() == 0
        if k0 < len(vals):
            G.add_cuts(vals[k0:], ideals[k0:])
    G.dual_adjacency()
    can = ph.canonical(G.dump())
    G.close()
    ph.assert_matches_large_golden(can, _GOLD_L, _META_L, name)


def test_full_size_properties_q5_N1000():
    """BASELINE-size poly-only synthetic (q=5, N=1000; the reference: 27 496 live vertices):
    size-independent properties instead of an oracle run."""
    q, N = 5, 1000
    D = ph.tangent_halfspaces(q, N, 5)
    G = PolyEngine(q)
    for k in range(q + 3):
        G.add(D[k])
    assert G.init() == 0
    rc = G.add_cuts(D[q + 3:])
    assert G.rounds_run() < len(D) - q - 3     # some cuts were grouped
    d = G.dump()
    live = d["pu"].astype(bool)
    X = d["X"]
    pts = live & (d["pi"] == 0)
    # (1) feasibility: every live point satisfies every applied halfspace d.y >= -1
    slack = X[pts] @ D.T + 1.0
    assert slack.min() > -1e-8
    # (2) incidence is geometric: a vertex lies on its facets, and on at least q of them
    cnt = np.zeros(len(live), int)
    for a, f in d["I"]:
        if f >= 1 and live[a] and not d["pi"][a]:
            assert abs(X[a] @ d["Y"][f] + 1.0) < 1e-7
            cnt[a] += 1
    assert cnt[pts].min() >= q
    # (3) simple polytope: every vertex has exactly q neighbours; Euler-Poincare count of a simple 5-polytope's graph
    deg = np.bincount(d["E"].ravel(), minlength=len(live))
    assert np.all(deg[pts] == q)
    # (4) idempotence: adding the same halfspaces again changes nothing
    rc2 = G.add_cuts(D[:50])
    assert np.all(rc2 == 1)
    d2 = G.dump()
    assert np.array_equal(d2["pu"], np.concatenate([d["pu"]])) and np.array_equal(d2["E"], d["E"])
    assert int(pts.sum()) == 27222       # same input as the oracle/reference comparison run (seed 5)
    G.close()


def test_k1_on_the_matrix_pipe_is_bit_identical_to_the_scalar_kernel():
    """K1 (the batched incidence kernel) has a variant that computes its dot products with v_mfma_f64_16x16x4 (debug key 9; not the
    default: DESIGN.md 4).  (1) The raw results of chained MFMAs equal the scalar fma chain in every bit (same order of
    accumulation): 20 000 random tiles per dimension.  (2) The class words, touch counts, first touched halfspace and 'some
    element violates' flags of the two kernels are identical on random points, on halfspaces laid THROUGH points and +-0.5e-9 /
    +-1.5e-9 beside them (inside and just outside the on-plane band), and on a real polyhedron with directions and freed slots.
    (3) A batched run of a SURVEY 8c golden with the matrix kernel switched on ends in the reference's polyhedron."""
    import ctypes
    from bensolve_amd._lib import load_library, check
    lib = load_library()
    for dim in (2, 3, 4, 5, 8, 10):
        mism = ctypes.c_long(-1)
        check(lib.bslv_k1_mfma_selftest(dim, 20000, ctypes.c_ulonglong(dim), ctypes.byref(mism)))
        assert mism.value == 0, (dim, mism.value)

    def both(G, hps):
        res = {}
        for mode in (1, 0):
            G.debug_set(9, mode)
            words, anym, _ = G.classify_batch(hps)
            w2, tc, t1 = G.classify_batch_touch(hps)
            res[mode] = (words.copy(), anym.copy(), w2, tc, t1)
        for a, b in zip(res[0], res[1]):
            assert np.array_equal(a, b)
        return res[1]

    try:
        for q, nv, B in ((5, 200_000, 96), (3, 50_000, 40), (8, 30_000, 16), (5, 39, 192), (4, 1000, 512)):
            G = PolyEngine(q)
            G.bench_fill(nv, 3)
            X = G.dump()["X"]
            rng = np.random.default_rng(q)
            H = rng.normal(size=(B, q))
            alpha = rng.normal(size=B) * 0.3
            for b in range(0, B, 2):                           # every other halfspace passes (nearly) through a point
                p = X[rng.integers(nv)]
                alpha[b] = H[b] @ p + (0.0, 0.5e-9, -0.5e-9, 1.5e-9, -1.5e-9)[(b // 2) % 5]
            r = both(G, np.hstack([H, alpha[:, None]]))
            G.close()
            assert (((r[0][0] >> np.uint64(0)) & np.uint64(3)) == 2).any()      # the band is exercised
        # a real polyhedron: directions (measured against 0), freed slots, few elements
        name = "tangent_q5_N200"
        q, v2h, apex, init_after = [int(x) for x in _GOLD_L[name + "/in_meta"]]
        vals, ideals = _GOLD_L[name + "/in_vals"], list(_GOLD_L[name + "/in_ideals"])
        for extra in (0, 40):
            G = PolyEngine(q, v2h)
            for k in range(init_after + extra):
                G.add(vals[k], ideals[k])
                if k + 1 == init_after:
                    assert G.init() == 0
            d = G.dump()
            assert (d["pi"] & d["pu"]).any() if extra == 0 else (d["pu"] == 0).any()      # live directions / freed slots
            rng = np.random.default_rng(extra)
            for B in (16, 100, 192):
                both(G, np.hstack([rng.normal(size=(B, G.d)), rng.normal(size=(B, 1))]))
            G.close()
        # batched run with the matrix kernel
        G = PolyEngine(q, v2h)
        G.debug_set(9, 1)
        for k in range(init_after):
            G.add(vals[k], ideals[k])
        assert G.init() == 0
        G.add_cuts(vals[init_after:], ideals[init_after:])
        G.dual_adjacency()
        can = ph.canonical(G.dump())
        G.close()
        ph.assert_matches_large_golden(can, _GOLD_L, _META_L, name)
    finally:
        G = PolyEngine(3)
        G.debug_set(9, 0)
        G.close()


def test_snap_band_of_poly_cut_matches_the_reference_when_switched_on():
    """bslv_poly.c:666-674 (bslv_poly_set_snap): an element between 1e-2 POLY_EPS and POLY_EPS above a cut that removes something is moved
    onto the hyperplane.  Fixtures from the unmodified bslv_poly.c (tests/golden/poly_ref_snap.npz: a crafted cut 5e-10 / 5e-11 / 5e-12 above
    a live vertex, twelve ordinary cuts behind it).  With the switch on the engine agrees with the reference to 1e-12 and with the oracle's
    restatement of the band slot for slot, handing the cuts in one by one or as one batch; the number of moved elements is the oracle's.
    With the switch off (the default) it agrees with the oracle's default slot for slot and with the reference at 1e-9."""
    import ctypes
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poly_ref_snap.npz"))
    for name in sorted({k.split("/")[0] for k in G.files}):
        q, v2h, apex, init_after = [int(x) for x in G[name + "/in_meta"]]
        vals = G[name + "/in_vals"]
        g = lambda k: G[name + "/" + k]
        gold = dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")}, DE={tuple(e) for e in g("DE")})
        delta = float(name.split("delta")[1])
        for snap in (1, 0):
            O = ph.FlatPoly("oracle", q, v2h)
            O.L.opoly_set_snap.argtypes = [ctypes.c_void_p, ctypes.c_int]
            O.L.opoly_snapped.argtypes = [ctypes.c_void_p]
            O.L.opoly_snapped.restype = ctypes.c_long
            O.L.opoly_set_snap(O.h, snap)
            rco = ph.run_sequence(O, vals, None, init_after)
            O.dual_adjacency()
            do, moved_o = O.dump(), O.L.opoly_snapped(O.h)
            O.close()
            assert moved_o == (1 if snap and delta > 1e-11 else 0)
            for batched in (False, True):
                E = PolyEngine(q, v2h, None)
                E.set_batch_mode(1 if batched else 0)
                E.set_snap(snap)
                if batched:
                    rcg = [E.add(vals[i], 0) for i in range(init_after)]
                    assert E.init() == 0
                    rcg += list(E.add_cuts(vals[init_after:], None))
                else:
                    rcg = ph.run_sequence(E, vals, None, init_after)
                E.dual_adjacency()
                dg, moved_g = E.dump(), E.snapped()
                E.close()
                if snap or not batched:
                    assert list(rcg) == list(rco) == list(g("rc"))
                assert moved_g == moved_o
                if snap or not batched:           # (the default's batched form = rounds of independent cuts: same sets, other slot numbers)
                    assert_slotwise_equal(do, dg)
                ph.assert_same(ph.canonical(dg), gold, rtol=0, atol=1e-12 if snap else 1e-9)


def test_snap_band_moves_a_direction_as_the_reference_does():
    """The band for an ideal element (alpha = 0, bslv_poly.c:596,666): tests/golden/poly_ref_snap_dirs.npz, an extreme direction of an
    unbounded polyhedron with w.r = 5e-10 / 5e-11 / 5e-12 for the crafted cut w.  Engine with the switch on against the oracle's
    restatement slot for slot and against the reference's fixture (relative 1e-12: far vertices)."""
    import ctypes
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poly_ref_snap_dirs.npz"))
    for name in sorted({k.split("/")[0] for k in G.files}):
        q, v2h, apex, init_after = [int(x) for x in G[name + "/in_meta"]]
        vals = G[name + "/in_vals"]
        g = lambda k: G[name + "/" + k]
        gold = dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")}, DE={tuple(e) for e in g("DE")})
        O = ph.FlatPoly("oracle", q, v2h)
        O.L.opoly_set_snap.argtypes = [ctypes.c_void_p, ctypes.c_int]
        O.L.opoly_snapped.argtypes = [ctypes.c_void_p]
        O.L.opoly_snapped.restype = ctypes.c_long
        O.L.opoly_set_snap(O.h, 1)
        rco = ph.run_sequence(O, vals, None, init_after)
        O.dual_adjacency()
        do, moved_o = O.dump(), O.L.opoly_snapped(O.h)
        O.close()
        E = PolyEngine(q, v2h, None)
        E.set_batch_mode(0)
        E.set_snap(1)
        rcg = ph.run_sequence(E, vals, None, init_after)
        E.dual_adjacency()
        dg, moved_g = E.dump(), E.snapped()
        E.close()
        assert list(rcg) == list(rco) == list(g("rc")), name
        assert moved_g == moved_o == (1 if float(name.split("delta")[1]) > 1e-11 else 0), name
        assert_slotwise_equal(do, dg)
        ph.assert_same(ph.canonical(dg), gold, rtol=1e-12, atol=1e-12)
