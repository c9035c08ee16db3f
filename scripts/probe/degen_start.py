import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
for (m, n, q) in [(400, 200, 5), (1000, 500, 6), (2000, 1000, 8), (4000, 2000, 10)]:
    prob = synth.degenerate_vlp(m, n, q, 3)
    t0 = time.time()
    try:
        eng = BensonEngine(prob, eps=1e-7, pool_slots=40)
        s = eng.start()
        tot = eng.totals()
        print((m, n, q), "start status", s, "lps", tot["lps"], "pivots", tot.get("pivots"), "%.1fs" % (time.time() - t0), flush=True)
        if s == 0:
            for k in range(3):
                nl, nt = eng.collect(16, 0, 1)
                rec, piv, ls = eng.solve_local(nl)
                st = eng.apply(rec)
                print("   step", k, "lps", nl, "pivots", piv, "lockstep", ls, "cuts", st["cuts"], "%.1fs" % (time.time() - t0), flush=True)
        eng.close()
    except Exception as e:
        print((m, n, q), "FAILED", repr(e)[:300], "%.1fs" % (time.time() - t0), flush=True)
