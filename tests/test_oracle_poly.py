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
