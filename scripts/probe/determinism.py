"""Is a batched run reproducible?  The same problem twice through the engine: dumps compared bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, hashlib
import bensolve_amd._lib as _l
if os.environ.get("BSLV_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["BSLV_LIB"])                 # A/B against another build of the library
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
def run(prob, batch):
    eng = BensonEngine(prob, eps=1e-9, pool_slots=max(4 * batch, 64))
    assert eng.start() == 0
    eng.run(batch)
    d = eng.poly_dump(); t = eng.totals(); r2 = eng.poly_call("rounds2_stats")
    eng.close()
    return d, t, r2
for (m, n, q, seed, batch) in ((40, 20, 4, 9, 32), (60, 30, 4, 3, 128)):
    prob = synth.covering_vlp(m, n, q, seed)
    a, ta, ra = run(prob, batch); b, tb, rb = run(prob, batch)
    same = all(np.array_equal(a[k], b[k]) for k in ("X", "Y", "E", "I", "pu", "pi"))
    hs = hashlib.sha256(b"".join(np.ascontiguousarray(a[k]).tobytes() for k in ("X", "Y", "E", "I", "pu", "pi"))).hexdigest()[:16]
    print((m, n, q, seed), "sha", hs, "nv", len(a["X"]), len(b["X"]), "identical" if same else "DIFFERENT", "lps", ta.get("lps"), tb.get("lps"), "rounds", ra.get("rounds"), rb.get("rounds"), flush=True)
