import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import oracle_api, poly_harness as ph
from bensolve_amd import synth
for (m, n, q, seed) in [(24, 12, 3, 3), (60, 30, 3, 3), (60, 30, 4, 3), (120, 60, 4, 3)]:
    prob = synth.degenerate_vlp(m, n, q, seed)
    t0 = time.time()
    try:
        rc, fp, st = oracle_api.benson_phase2_primal(prob, eps=1e-9, max_lps=50000)
        fp.dual_adjacency(); exp = ph.canonical(fp.dump(), decimals=6); fp.close()
        print("oracle", (m, n, q), "rc", rc, "lps", st.lps, "pivots", st.pivots, "vertices", len(exp["X"]), "%.1fs" % (time.time() - t0), flush=True)
    except Exception as e:
        print("oracle", (m, n, q), "FAILED", repr(e)[:200], flush=True); continue
    if len(sys.argv) > 1:
        from bensolve_amd.benson import BensonEngine
        t0 = time.time()
        try:
            eng = BensonEngine(prob, eps=1e-9, pool_slots=600)
            s = eng.start()
            assert s == 0, s
            eng.run(128)
            eng.poly_call("dual_adjacency")
            got = ph.canonical(eng.poly_dump(), decimals=6); tot = eng.totals(); eng.close()
            ph.assert_benson_results_agree(got, exp)
            print("gpu   ", (m, n, q), "OK lps", tot["lps"], "pivots", tot.get("pivots"), "%.1fs" % (time.time() - t0), flush=True)
        except Exception as e:
            print("gpu   ", (m, n, q), "FAILED", repr(e)[:300], flush=True)
