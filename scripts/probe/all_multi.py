import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bensolve_amd.poly import PolyEngine
path, n = sys.argv[1], int(sys.argv[2])
Z = np.load(path); Y, q, c = Z["Y"], int(Z["q"]), Z["c"]
res = {}
for mode in ("fused", "multi"):
    G = PolyEngine(q, 1, c); G.set_batch_mode(0)
    if mode == "multi": G.debug_set(0, 64)
    for k in range(1, q + 1): G.add(Y[k], 0)
    assert G.init() == 0
    rest = Y[q + 1:q + 1 + n]
    for b0 in range(0, n - 1, 256): G.add_cuts(rest[b0:min(b0 + 256, n - 1)], None)
    D0 = G.dump()
    G.add(rest[n - 1], 0)
    D1 = G.dump()
    res[mode] = (D0, D1, G.path_stats())
    G.close()
a0, a1, pa = res["fused"]; b0, b1, pb = res["multi"]
print(pa, pb)
print("state before equal:", all(np.array_equal(a0[k], b0[k]) for k in ("pu", "X", "E", "I")))
print("edges after: fused", len(a1["E"]), "multi", len(b1["E"]), "slots", len(a1["pu"]), len(b1["pu"]), "I equal", np.array_equal(a1["I"], b1["I"]))
sa = set(map(tuple, a1["E"])); sb = set(map(tuple, b1["E"]))
only_f = sorted(sa - sb); only_m = sorted(sb - sa)
print("only fused:", len(only_f), only_f[:30]); print("only multi:", len(only_m), only_m[:10])
I = a1["I"]; inc = {}
for v, f in I: inc.setdefault(int(v), []).append(int(f))
nv0 = len(a0["pu"])
newf = max(f for v, f in I)
mem = sorted(v for v in inc if newf in inc[v])
print("new facet", newf, "members", len(mem), "new vertices from", nv0)
for (u, v) in only_f[:8]:
    print(u, v, "len", len(inc.get(u, [])), len(inc.get(v, [])), "ideal", int(a1["pi"][u]), int(a1["pi"][v]), "mutual", sorted(set(inc.get(u, [])) & set(inc.get(v, []))))
