#!/usr/bin/env python3
"""zero_share_sim.py -- OFFLINE (CPU, numpy/scipy) on a snapshot of state_dump.py: how many rounds of mutually independent cuts a
batch of whole families needs (a) with the conflict relation of the rounds as built -- two cuts conflict when an element is non-PLUS
for both or an edge joins their non-PLUS regions -- and (b) if elements that are merely ON the plane of several cuts could be shared
(DESIGN.md 9 item 2): conflict only when a MINUS element / a MINUS end of an edge is involved."""
import sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from select_sim import halfspaces, dedupe, touch_matrix, rounds_mis

S = np.load(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
X, pu, pi, E = S["X"], S["pu"], S["pi"], S["E"]
q_par, lp_pos, rec, c = S["q_par"], S["lp_pos"], S["lp_rec"], S["c"]
q = X.shape[1]
live = np.nonzero(pu)[0]
remap = -np.ones(len(pu), np.int64); remap[live] = np.arange(len(live))
Xl, il, El = X[live], pi[live] != 0, remap[E]
lp_par = q_par[lp_pos]
rng = np.random.default_rng(5)
# whole families: parents in random order until the batch is full (the snapshot has no z of the parents)
fams = {}
for k, p in enumerate(lp_par):
    fams.setdefault(int(p), []).append(k)
order = rng.permutation(list(fams.keys()))
pick = []
for p in order:
    if len(pick) >= B: break
    pick += fams[int(p)]
pick = np.array(pick[:B])
r = rec[pick]
a = r[:, 2] != 0
ys = r[a][:, 4:4 + q]
first, inv = dedupe(ys)
hp = halfspaces(ys[first], c)
T = touch_matrix(Xl, il, hp)                     # 1 MINUS, 2 ZERO
n = T.shape[0]
A = sp.csr_matrix((np.ones(len(El), np.int32), (El[:, 0], El[:, 1])), shape=(n, n)); A = A + A.T
def graph(Ta, Tb):
    """cuts i, j conflict when an element is in Ta for i and Tb for j, or an edge joins Ta(i) and Tb(j) (symmetrised)"""
    Ta = Ta.astype(np.int32); Tb = Tb.astype(np.int32)
    C = (Ta.T @ Tb) + (Ta.T @ (A @ Tb))
    C = ((C + C.T) != 0).tolil(); C.setdiag(False)
    return C.tocsr()
NP, MI = (T != 0), (T == 1)
for name, C in (("as built (non-PLUS x non-PLUS)", graph(NP, NP)), ("shared on-plane elements (MINUS x non-PLUS only)", graph(MI, NP))):
    deg = np.diff(C.indptr)
    prio = rng.permutation(C.shape[0])
    t0 = time.time()
    rr, sizes = rounds_mis(C, prio)
    print("%-50s cuts %d, families %d | conflict degree mean %.1f max %d | rounds of maximal independent sets: %d (first %s)" % (
        name, C.shape[0], len(set(lp_par[pick].tolist())), deg.mean(), deg.max(), rr, sizes[:6]), flush=True)
zero_only = np.asarray(((T == 2).sum(1))).ravel()
print("elements ON the plane of k cuts of the batch: k>=2: %d, k>=8: %d, max %d" % ((zero_only >= 2).sum(), (zero_only >= 8).sum(), zero_only.max()))
