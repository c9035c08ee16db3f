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
    rcs = []
    for b0 in range(0, n, 256): rcs += list(G.add_cuts(rest[b0:min(b0 + 256, n)], None))
    res[mode] = (G.dump(), rcs)
    G.close()
(a, ra), (b, rb) = res["fused"], res["multi"]
print("rc equal", ra == rb, "redundant", sum(ra), sum(rb))
print("edges: fused", len(a["E"]), "multi", len(b["E"]), "slots", len(a["pu"]), len(b["pu"]), "I equal", np.array_equal(a["I"], b["I"]), "X equal", np.array_equal(a["X"], b["X"]))
sa = set(map(tuple, a["E"])); sb = set(map(tuple, b["E"]))
only_f = sorted(sa - sb); only_m = sorted(sb - sa)
print("only fused:", len(only_f), only_f[:24]); print("only multi:", len(only_m), only_m[:10])
inc = {}
for v, f in a["I"]: inc.setdefault(int(v), []).append(int(f))
for (u, v) in only_f[:10]:
    print(u, v, "len", len(inc.get(u, [])), len(inc.get(v, [])), "used", int(a["pu"][u]), int(a["pu"][v]), "ideal", int(a["pi"][u]), int(a["pi"][v]), "mutual", sorted(set(inc.get(u, [])) & set(inc.get(v, []))))
