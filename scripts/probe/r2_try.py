import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import poly_harness as ph
from bensolve_amd.poly import PolyEngine
for q, N, seed in [(3, 400, 1), (4, 200, 2), (5, 300, 3), (6, 60, 4), (3, 2000, 31), (5, 1000, 33)]:
    D = ph.tangent_halfspaces(q, N, seed)
    res = {}
    for mode in (0, 1):
        G = PolyEngine(q)
        G.debug_set(6, mode)
        for k in range(q + 3):
            G.add(D[k])
        assert G.init() == 0
        t = time.time()
        rc = G.add_cuts(D[q + 3:])
        dt = time.time() - t
        G.dual_adjacency()
        res[mode] = (ph.canonical(G.dump()), rc.copy(), G.rounds2_stats(), dt)
        G.close()
    a, b = res[0], res[1]
    print(q, N, "single %.3fs rounds2 %.3fs" % (a[3], b[3]), b[2], "rc equal", np.array_equal(a[1], b[1]), "sizes", len(a[0]["X"]), len(b[0]["X"]), flush=True)
    ph.assert_same(b[0], a[0])
print("OK")
