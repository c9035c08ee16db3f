"""Probe: first steps of S-degenerate under the variants of the multi-kernel prune; hashes of the polyhedron per step."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
prob = synth.CONFIGS["S-degenerate"]()
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, fm_min in (("member lists", 4096), ("full scan", 1 << 30)):
    eng = BensonEngine(prob, eps=1e-7, pool_slots=4 * 64 + 64)
    eng.poly_call("debug_set", 4, fm_min)
    assert eng.start() == 0
    line = []
    for k in range(nsteps):
        nl, nt = eng.collect(64, 0, 1)
        rec, piv, ls = eng.solve_local(nl)
        st = eng.apply(rec)
        c = eng.poly_call("counts")
        line.append((nl, st["cuts"], c["nprimal"], c["nedges"]))
    d = eng.poly_dump()
    h = hashlib.sha256()
    for key in ("pu", "pi", "du", "di", "X", "Y", "E", "I"):
        h.update(np.ascontiguousarray(d[key]).tobytes())
    print(name, line, int(d["pu"].sum()), h.hexdigest()[:16], eng.poly_call("path_stats"), flush=True)
    eng.close()
