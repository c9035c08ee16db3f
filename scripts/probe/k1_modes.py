"""K1 on the matrix pipe against the scalar kernel on a REAL polyhedron (directions, dead slots, small nv), with and without touch counts"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bensolve_amd.poly import PolyEngine
gold = np.load("tests/golden/poly_ref_large.npz")
for name, k1 in (("tangent_q3_N2000", 300), ("tangent_q5_N200", 60), ("tangent_q3_N2000", 8)):
    q, v2h, apex, init_after = [int(x) for x in gold[name + "/in_meta"]]
    vals, ideals = gold[name + "/in_vals"], list(gold[name + "/in_ideals"])
    G = PolyEngine(q, v2h)
    if apex: G.dual0_apex()
    k0 = len(vals) if init_after < 0 else init_after
    for k in range(k0): G.add(vals[k], ideals[k])
    assert G.init() == 0
    for k in range(k0, k0 + k1): G.add(vals[k], ideals[k])
    d = G.dump()
    print(name, "nv", len(d["X"]), "ideal", int(d["pi"].sum()), "dead", int((d["pu"] == 0).sum()), flush=True)
    for B in (16, 40, 100):
        hps = vals[k0 + k1:k0 + k1 + B]
        res = {}
        for mode in (1, 0):
            G.debug_set(9, mode)
            w, anym, _ = G.classify_batch(hps)
            w2, tc, t1 = G.classify_batch_touch(hps)
            res[mode] = (w, anym, w2, tc, t1)
        G.debug_set(9, 1)
        for j, nm in enumerate(("words", "anyminus", "words(touch)", "tc", "t1")):
            a, b = res[0][j], res[1][j]
            if not np.array_equal(a, b):
                bad = np.argwhere(a != b)
                print("  B", B, nm, "DIFFERS at", len(bad), "first", bad[:5].tolist(), [hex(int(x)) if nm.startswith("w") else int(x) for x in a[tuple(bad[0])].ravel()[:1]], [hex(int(x)) if nm.startswith("w") else int(x) for x in b[tuple(bad[0])].ravel()[:1]])
            else:
                print("  B", B, nm, "equal")
    G.close()
