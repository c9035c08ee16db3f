"""Two batch selection rules on the same terminating problem: each final polyhedron is an outer approximation of the same upper image
within eps, so the vertices of one must satisfy the cuts of the other up to ~eps.  usage: policy_check.py WORKLOAD BATCH POLICY_A POLICY_B"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
name, batch = sys.argv[1], int(sys.argv[2])
res = {}
for pol in sys.argv[3:5]:
    os.environ["BSLV_POLICY"] = pol
    prob = synth.CONFIGS[name]()
    eng = BensonEngine(prob, eps=1e-7, pool_slots=4 * batch + 64)
    assert eng.start() == 0
    steps = eng.run(batch)
    left = eng.poly_call("unprocessed", 0)[3]
    d = eng.poly_dump()
    tot = eng.totals()
    eng.close()
    live = d["pu"].astype(bool) & (d["pi"] == 0)
    Y = d["Y"][d["du"].astype(bool) & (d["di"] == 0)]
    res[pol] = dict(X=d["X"][live], Y=Y, steps=steps, left=left, lps=tot["lps"], cuts=tot["cuts"])
    print(pol, "steps", steps, "unprocessed left", left, "LPs", tot["lps"], "cuts", tot["cuts"], "vertices", live.sum(), "facets", len(Y), flush=True)
rng = np.random.default_rng(1)
a, b = sys.argv[3], sys.argv[4]
for p, q in ((a, b), (b, a)):
    X = res[p]["X"]; Y = res[q]["Y"]
    X = X[rng.choice(len(X), min(3000, len(X)), replace=False)]
    w = np.hstack([Y[:, :-1], 1 - Y[:, :-1].sum(axis=1, keepdims=True)])
    worst = 0.0
    for c0 in range(0, len(w), 20000):
        worst = min(worst, float((X @ w[c0:c0 + 20000].T - Y[c0:c0 + 20000, -1][None, :]).min()))
    print("vertices of %s against the cuts of %s: deepest violation %.3e" % (p, q, worst), flush=True)
