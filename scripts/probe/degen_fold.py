"""Probe: S-degenerate with the singleton rows (I_n, 'd 0 1') folded into column bounds -- same VLP, smaller LP,
every structural column boxed.  Does the cold start of PART 1 get through, and at what pivot count?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine

def fold(prob):
    n = prob["n"]; g = prob["m"] - n
    return dict(m=g, n=n, q=prob["q"], A=np.ascontiguousarray(prob["A"][n:]), P=prob["P"], optdir=1,
                rtype=prob["rtype"][n:].copy(), rlb=prob["rlb"][n:].copy(), rub=prob["rub"][n:].copy(),
                ctype=np.full(n, ord("d"), np.uint8), clb=np.zeros(n), cub=np.ones(n))

def main():
    sizes = [(400, 200, 5), (1000, 500, 6), (2000, 1000, 8), (4000, 2000, 10)]
    if len(sys.argv) > 1: sizes = sizes[: int(sys.argv[1])]
    for (m, n, q) in sizes:
        for folded in (1, 0):
            prob = synth.degenerate_vlp(m, n, q, 3)
            if folded: prob = fold(prob)
            t0 = time.time()
            try:
                eng = BensonEngine(prob, eps=1e-7, pool_slots=40)
                s = eng.start()
                tot = eng.totals()
                print((m, n, q), "folded" if folded else "as given", "start status", s, "lps", tot["lps"], "pivots", tot.get("pivots"), "%.1fs" % (time.time() - t0), flush=True)
                if s == 0:
                    for k in range(3):
                        nl, nt = eng.collect(16, 0, 1)
                        rec, piv, ls = eng.solve_local(nl)
                        st = eng.apply(rec)
                        print("   step", k, "lps", nl, "pivots", piv, "lockstep", ls, "cuts", st["cuts"], "%.1fs" % (time.time() - t0), flush=True)
                eng.close()
            except Exception as e:
                print((m, n, q), "FAILED", repr(e)[:300], "%.1fs" % (time.time() - t0), flush=True)
            if not folded and m >= 2000: break

if __name__ == "__main__":
    main()
