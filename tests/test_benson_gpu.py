"""GPU end-to-end parity: batched HIP Benson phase 2 vs the sequential CPU oracle (oracle/benson_cpu.c).

The batched driver applies cuts in a different ORDER than the sequential loop, so slot numbers and
transient polytopes differ; the final vertex / facet / incidence / adjacency SETS must agree
(SURVEY.md 8c comparison rule).  Coordinates: Benson accepts a vertex un-cut when its LP value is
<= eps (bslv_algs.c:1063), so both sides run with eps = 1e-9 here and coordinates are compared at
1e-7 (the LP tolerance), index sets exactly."""
import numpy as np
import pytest

import oracle_api
import poly_harness as ph
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,q,seed,batch", [(12, 6, 2, 3, 4), (30, 15, 3, 5, 16), (60, 30, 3, 7, 64), (40, 20, 4, 9, 128)])
def test_phase2_sets_match_oracle(m, n, q, seed, batch):
    prob = synth.covering_vlp(m, n, q, seed)
    eps = 1e-9
    rc, fp, st = oracle_api.benson_phase2_primal(prob, eps=eps)
    assert rc == 0
    fp.dual_adjacency()
    exp = ph.canonical(fp.dump(), decimals=6)
    fp.close()
    eng = BensonEngine(prob, eps=eps, pool_slots=max(4 * batch, 64))
    assert eng.start() == 0
    eng.run(batch)
    eng.poly_call("dual_adjacency")
    got = ph.canonical(eng.poly_dump(), decimals=6)
    tot = eng.totals()
    eng.close()
    ph.assert_same(got, exp, rtol=1e-7, atol=1e-7)
    # every vertex needs at least one LP, every facet one
    assert tot["lps"] >= len(exp["X"]) - q
