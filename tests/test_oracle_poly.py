"""CPU: the oracle polyhedron restatement (oracle/poly_dd.c) against golden outputs of the REFERENCE
polyhedron engine (tests/golden/poly_ref.npz, made by make_golden.py from the unmodified
bslv_poly.c) and, where oracle/_ref exists, against the compiled reference live."""
import os
import numpy as np
import pytest

import poly_harness as ph

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "poly_ref.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


def golden_case(name):
    q, v2h, apex, init_after = [int(x) for x in GOLD[name + "/in_meta"]]
    return q, v2h, bool(apex), (None if init_after < 0 else init_after), GOLD[name + "/in_vals"], GOLD[name + "/in_ideals"]


def golden_canonical(name):
    g = lambda k: GOLD[name + "/" + k]
    return dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")},
                DE={tuple(e) for e in g("DE")})


def run_engine(P, name):
    q, v2h, apex, init_after, vals, ideals = golden_case(name)
    if apex:
        P.dual0_apex()
    rcs = ph.run_sequence(P, vals, list(ideals), init_after)
    P.dual_adjacency()
    return rcs, ph.canonical(P.dump())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_golden(name):
    q, v2h, apex, init_after, vals, ideals = golden_case(name)
    P = ph.FlatPoly("oracle", q, v2h)
    rcs, can = run_engine(P, name)
    P.close()
    assert list(rcs) == list(GOLD[name + "/rc"])            # same cuts found redundant (EXIT_FAILURE)
    ph.assert_same(can, golden_canonical(name))


import json
GOLD_L = np.load(os.path.join(HERE, "golden", "poly_ref_large.npz"))
META_L = json.load(open(os.path.join(HERE, "golden", "poly_ref_large.json")))


@pytest.mark.parametrize("name", sorted(META_L))
def test_oracle_matches_reference_golden_at_survey_sizes(name):
    """SURVEY.md 8c's sizes (q=3 N=2000, q=5 N=200 / N=1000, q=8 N=60, ex06's dual cone, ex10's lattice directions): the
    reference's own bslv_poly.c produced the fixture (tests/golden/make_golden.py large); coordinates at 1e-9, index sets
    bit-exact by SHA-256 of the canonical labelling"""
    q, v2h, apex, init_after = [int(x) for x in GOLD_L[name + "/in_meta"]]
    P = ph.FlatPoly("oracle", q, v2h)
    if apex:
        P.dual0_apex()
    rcs = ph.run_sequence(P, GOLD_L[name + "/in_vals"], list(GOLD_L[name + "/in_ideals"]), None if init_after < 0 else init_after)
    P.dual_adjacency()
    can = ph.canonical(P.dump())
    P.close()
    assert list(rcs) == list(GOLD_L[name + "/rc"])
    ph.assert_matches_large_golden(can, GOLD_L, META_L, name)


@pytest.mark.skipif(not ph.ref_available(), reason="oracle/_ref/libref_poly.so only exists in the build container")
@pytest.mark.parametrize("q,N,seed", [(3, 500, 21), (4, 150, 22), (5, 150, 23)])
def test_oracle_matches_compiled_reference_live(q, N, seed):
    D = ph.tangent_halfspaces(q, N, seed)
    res = {}
    for kind in ("oracle", "ref"):
        P = ph.FlatPoly(kind, q)
        ph.run_sequence(P, D, init_after=q + 2)
        P.dual_adjacency()
        res[kind] = ph.canonical(P.dump())
        P.close()
    ph.assert_same(res["oracle"], res["ref"])


def test_get_vrtx_order_and_marks():
    # poly__get_vrtx returns the lowest live slot without the sltn mark (bslv_poly.c:210-226)
    P = ph.FlatPoly("oracle", 3)
    ph.run_sequence(P, ph.tangent_halfspaces(3, 20, 5))
    seen = []
    while True:
        nxt = P.next()
        if nxt is None:
            break
        seen.append(nxt[2])
        P.mark(nxt[2])
    d = P.dump()
    assert seen == sorted(seen) and len(seen) == int(d["pu"].sum())
    P.close()


GOLD_S = np.load(os.path.join(HERE, "golden", "poly_ref_snap.npz"))


def run_snap_case(name, snap):
    import ctypes
    q, v2h, apex, init_after = [int(x) for x in GOLD_S[name + "/in_meta"]]
    P = ph.FlatPoly("oracle", q, v2h)
    P.L.opoly_set_snap.argtypes = [ctypes.c_void_p, ctypes.c_int]
    P.L.opoly_snapped.argtypes = [ctypes.c_void_p]
    P.L.opoly_snapped.restype = ctypes.c_long
    P.L.opoly_set_snap(P.h, snap)
    rcs = ph.run_sequence(P, GOLD_S[name + "/in_vals"], list(GOLD_S[name + "/in_ideals"]), init_after)
    P.dual_adjacency()
    can, moved = ph.canonical(P.dump()), P.L.opoly_snapped(P.h)
    P.close()
    g = lambda k: GOLD_S[name + "/" + k]
    gold = dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")}, DE={tuple(e) for e in g("DE")})
    assert list(rcs) == list(g("rc"))
    return can, gold, moved


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in GOLD_S.files}))
def test_snap_band_of_poly_cut_against_the_reference(name):
    """bslv_poly.c:666-674: a neighbour of a removed element that lies between 1e-2 POLY_EPS and POLY_EPS above the cut is moved ONTO the
    hyperplane before it is treated as lying on it.  Fixtures (tests/golden/make_golden.py snap, from the unmodified bslv_poly.c): one
    crafted cut passes delta above a live vertex, twelve ordinary cuts follow.  With opoly_set_snap(1) the oracle restates the band and
    agrees with the reference to 1e-13 -- for delta inside the band (5e-10, 5e-11: one element moved) and below it (5e-12: none moved).
    With the switch off -- the default of the oracle and of the HIP engine (bslv_poly_set_snap) -- the index sets are the
    same and the coordinates differ by less than 1e-9, i.e. invisibly at the tolerance of every other comparison with the reference,
    but by more than 1e-11 where the band is hit: the size of the gap, measured."""
    delta = float(name.split("delta")[1])
    can, gold, moved = run_snap_case(name, 1)
    ph.assert_same(can, gold, rtol=0, atol=1e-13)
    assert moved == (1 if delta > 1e-11 else 0)
    can0, gold, moved0 = run_snap_case(name, 0)
    assert moved0 == 0
    ph.assert_same(can0, gold, rtol=0, atol=1e-9)
    gap = np.abs(can0["X"] - gold["X"]).max()
    if delta > 1e-11:
        assert 1e-11 < gap < 1e-9, gap
    else:
        assert gap < 1e-13, gap


GOLD_SD = np.load(os.path.join(HERE, "golden", "poly_ref_snap_dirs.npz"))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in GOLD_SD.files}))
def test_snap_band_moves_directions_too(name):
    """The same band for an ideal element (alpha = 0 in poly__cut's comparisons, bslv_poly.c:596,666): an unbounded polyhedron, a cut
    crafted so that w.r = delta for one of its extreme directions r while a neighbour of r is removed, eight ordinary cuts behind it
    (tests/golden/make_golden.py snap, from the unmodified bslv_poly.c).  The polyhedron has vertices with coordinates of ~6 next to the
    moved direction, where the reference's plain sums and the oracle's fma chains differ in the last bits: 1e-12 here, not 1e-13.
    The engine runs the same fixture in tests/test_poly_gpu.py::test_snap_band_moves_a_direction_as_the_reference_does."""
    import ctypes
    delta = float(name.split("delta")[1])
    q, v2h, apex, init_after = [int(x) for x in GOLD_SD[name + "/in_meta"]]
    g = lambda k: GOLD_SD[name + "/" + k]
    gold = dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")}, DE={tuple(e) for e in g("DE")})
    gaps = {}
    for snap in (1, 0):
        P = ph.FlatPoly("oracle", q, v2h)
        P.L.opoly_set_snap.argtypes = [ctypes.c_void_p, ctypes.c_int]
        P.L.opoly_snapped.argtypes = [ctypes.c_void_p]
        P.L.opoly_snapped.restype = ctypes.c_long
        P.L.opoly_set_snap(P.h, snap)
        rcs = ph.run_sequence(P, g("in_vals"), list(g("in_ideals")), init_after)
        P.dual_adjacency()
        can, moved = ph.canonical(P.dump()), P.L.opoly_snapped(P.h)
        P.close()
        assert list(rcs) == list(g("rc"))
        assert moved == (1 if snap and delta > 1e-11 else 0)
        ph.assert_same(can, gold, rtol=1e-12 if snap else 1e-8, atol=1e-12 if snap else 1e-8)
        gaps[snap] = (np.abs(can["X"] - gold["X"]) / (1.0 + np.abs(gold["X"]))).max()        # (far vertices of an unbounded polyhedron: relative)
    if delta > 1e-11:
        assert gaps[0] > 10 * gaps[1] and gaps[0] > 1e-11, gaps
