#!/usr/bin/env python3
"""Histogram of incidence-list lengths and edge end-point list lengths on S-mid after a few batched steps
(design probe: which list-intersection paths of the cut kernels carry the long tails)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine

B = 1024
prob = synth.CONFIGS["S-mid"]()
eng = BensonEngine(prob, eps=1e-7, pool_slots=4 * B + 64)
assert eng.start() == 0
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    nl, nt = eng.collect(B, 0, 1)
    rec, piv, ls = eng.solve_local(nl)
    eng.apply(rec)
D = eng.poly_dump()
pu, pi, I, E = D["pu"], D["pi"], D["I"], D["E"]
nv = len(pu)
cnt = np.bincount(I[:, 0], minlength=nv)
live = pu.astype(bool)
edges = [0, 6, 9, 17, 33, 65, 257, 1025, 10**9]
out = {"slots": int(nv), "live": int(live.sum()), "ideal_live": int((live & pi.astype(bool)).sum()), "edges": int(len(E)), "facets": int(D["du"].sum())}
out["len_hist_points"] = np.histogram(cnt[live & ~pi.astype(bool)], bins=edges)[0].tolist()
out["len_hist_dirs"] = np.histogram(cnt[live & pi.astype(bool)], bins=edges)[0].tolist()
la, lb = cnt[E[:, 0]], cnt[E[:, 1]]
lo, hi = np.minimum(la, lb), np.maximum(la, lb)
out["edge_both_le16"] = int((hi <= 16).sum())
out["edge_short_long"] = int(((lo <= 16) & (hi > 16)).sum())
out["edge_both_gt16"] = int((lo > 16).sum())
out["edge_both_gt16_hi_gt1024"] = int(((lo > 16) & (hi > 1024)).sum())
out["edge_both_gt64"] = int((lo > 64).sum())
out["bins"] = edges[:-1]
print(json.dumps(out))
