#!/usr/bin/env python3
"""dirty_heap.py CMD-LESS probe: fill a few tens of GB of device memory with a poison pattern, give it back to the runtime, THEN run
S-small to termination -- a process that has created and destroyed engines hands recycled memory to the next one, a fresh process
gets zero pages; a buffer that is only right when it starts out as zeros shows here.  usage: dirty_heap.py [GB] [eps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-7
bufs = []
left = int(gb * (1 << 30))
sizes = [1 << 30, 256 << 20, 64 << 20, 16 << 20, 4 << 20, 1 << 20, 256 << 10, 64 << 10, 16 << 10, 4 << 10]
k = 0
while left > 0:
    sz = sizes[k % len(sizes)]; k += 1
    bufs.append(torch.full((sz // 4,), 0x7F7F7F7F, dtype=torch.int32, device="cuda"))
    left -= sz
torch.cuda.synchronize()
n = len(bufs)
del bufs
torch.cuda.empty_cache()
print("poisoned and released %d buffers (%.0f GB)" % (n, gb), file=sys.stderr, flush=True)
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
sp = synth.CONFIGS["S-small"]()
e2 = BensonEngine(sp, eps=eps, pool_slots=4 * 2048 + 64)
e2.start()
steps = 0
while True:
    s = e2.step(2048); steps += 1
    print("step %d: %s" % (steps, {k2: s[k2] for k2 in ("lps", "cuts", "redundant", "confirmed", "left")}), file=sys.stderr, flush=True)
    if s["lps"] == 0 and s["left"] == 0:
        break
print("done: %d steps, %s, rounds2 %s" % (steps, e2.totals(), e2.poly_call("rounds2_stats")), flush=True)
e2.close()
