"""Probe: members of the S-degenerate family in higher dimension against the CPU oracle (complete runs, eps 1e-9)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import oracle_api
import poly_harness as ph
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
for (m, n, q, batch) in [(40, 20, 5, 64), (60, 30, 5, 128), (30, 15, 6, 64), (40, 20, 6, 128)]:
    prob = synth.degenerate_vlp(m, n, q, 3)
    t0 = time.time()
    rc, fp, st = oracle_api.benson_phase2_primal(prob, eps=1e-9)
    fp.dual_adjacency()
    exp = ph.canonical(fp.dump(), decimals=6)
    fp.close()
    t1 = time.time()
    eng = BensonEngine(prob, eps=1e-9, pool_slots=max(4 * batch, 64))
    assert eng.start() == 0
    eng.run(batch)
    eng.poly_call("dual_adjacency")
    got = ph.canonical(eng.poly_dump(), decimals=6)
    paths = eng.poly_call("path_stats")
    eng.close()
    try:
        ph.assert_benson_results_agree(got, exp)
        ok = "AGREE"
    except AssertionError as e:
        ok = "DIFFER: " + str(e)[:200]
    print((m, n, q), "oracle %.1fs gpu %.1fs" % (t1 - t0, time.time() - t1), "vertices", len(exp["X"]), "facets", len(exp["Y"]), "edges", len(exp["E"]), "inc", len(exp["I"]), ok, paths, flush=True)
