#!/usr/bin/env python3
"""the GPU half of bench.py's like-for-like pair on its own: S-small to termination (eps = 1e-7, 2048 LPs per step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
sp = synth.CONFIGS["S-small"]()
e2 = BensonEngine(sp, eps=eps, pool_slots=4 * 2048 + 64)
e2.start()
t = time.perf_counter()
steps = 0
while True:
    s = e2.step(2048); steps += 1
    print("step %d: %s" % (steps, {k: s[k] for k in ("lps", "cuts", "redundant", "confirmed", "left")}), file=sys.stderr, flush=True)
    if s["lps"] == 0 and s["left"] == 0:
        break
print("done: %d steps, %.3f s, %s, rounds2 %s, defer %s" % (steps, time.perf_counter() - t, e2.totals(), e2.poly_call("rounds2_stats"), e2.defer_stats()), flush=True)
e2.close()
