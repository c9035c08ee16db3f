#!/usr/bin/env python3
"""Poly-only microbenchmark of SURVEY.md 8d (no LP): N random tangent halfspaces d.y >= -1 in R^q through cone_polar, the
first q+3 queued and the rest cut one by one.  The reference's own polyhedron code (bslv_poly.c compiled into
oracle/_ref/libref_poly.so, one core) next to the GPU engine on the same sequence; the live-vertex counts must agree.
Figure of merit of SURVEY 8d K2: pair tests per second (pairs of the new facet examined by edge_test / the prune)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import poly_harness as ph
from bensolve_amd.poly import PolyEngine

out = []
for (q, N) in [(3, 2000), (5, 1000), (8, 60), (10, 40)]:
    vals = ph.tangent_halfspaces(q, N, 1)
    k0 = q + 3
    row = dict(q=q, N=N)
    if ph.ref_available() and "--no-ref" not in sys.argv:
        R = ph.FlatPoly("ref", q)
        t0 = time.perf_counter()
        ph.run_sequence(R, vals, init_after=k0)
        row["ref_cpu_s"] = round(time.perf_counter() - t0, 4)
        d = R.dump()
        row["ref_live_vertices"] = int(d["pu"].sum())
        R.close()
    for mode, name in ((0, "gpu_sequential"), (1, "gpu_rounds")):
        G = PolyEngine(q)
        G.set_batch_mode(mode)
        for k in range(k0):
            G.add(vals[k], 0)
        assert G.init() == 0
        t0 = time.perf_counter()
        G.add_cuts(vals[k0:])
        dt = time.perf_counter() - t0
        c = G.counts()
        d = G.dump()
        row[name + "_s"] = round(dt, 4)
        row[name + "_live_vertices"] = int(d["pu"].sum())
        row[name + "_pair_tests"] = int(c["pair_tests"])
        row[name + "_pair_tests_per_s"] = round(c["pair_tests"] / dt, 0)
        row[name + "_new_vertices_per_s"] = round(c["new_vertices"] / dt, 0)
        G.close()
    if "ref_cpu_s" in row:
        row["ref_pair_tests_per_s"] = round(row["gpu_sequential_pair_tests"] / row["ref_cpu_s"], 0)     # same pairs by construction
        row["speedup_rounds_over_ref"] = round(row["ref_cpu_s"] / row["gpu_rounds_s"], 1)
    print(json.dumps(row), flush=True)
    out.append(row)
if len(sys.argv) > 1 and not sys.argv[-1].startswith("--"):
    json.dump(out, open(sys.argv[-1], "w"), indent=1)
