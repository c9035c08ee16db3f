"""Structure of the cuts of one S-mid batch (why do newest-first batches form a complete conflict graph?).

Runs the bench workload to a steady-state step, solves the LPs of the next batch, and -- before applying the cuts --
classifies the whole polyhedron against the distinct cuts on the host: per cut the MINUS / ZERO counts, per element the
number of cuts that touch it, and the conflict graph under three rules:
  (a) the engine's rule: a shared non-PLUS element or an edge between non-PLUS elements of two cuts
  (b) the same, but elements that are ZERO for both cuts do not count
  (c) the same as (b), and directions (ideal elements) never count
Prints the greedy independent-set sizes.  Measurement aid only.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine


def greedy(n, pairs):
    adj = [[] for _ in range(n)]
    for a, b in pairs:
        adj[a].append(b); adj[b].append(a)
    blocked = np.zeros(n, bool)
    sel = []
    for k in range(n):
        if blocked[k]:
            continue
        sel.append(k)
        for o in adj[k]:
            blocked[o] = True
    return len(sel)


def rounds(n, pairs):
    """number of rounds of greedy maximal independent sets until all cuts are applied (conflicts do not change: upper bound)"""
    adj = [set() for _ in range(n)]
    for a, b in pairs:
        adj[a].add(b); adj[b].add(a)
    left = list(range(n))
    r = 0
    while left:
        blocked = set()
        nxt = []
        for k in left:
            if k in blocked:
                nxt.append(k)
                continue
            blocked |= adj[k]
        left = nxt
        r += 1
    return r


def depth(n, pairs):
    """rounds when a round applies every cut that has no unapplied conflicting cut of lower index (local minima): DAG depth"""
    lower = [[] for _ in range(n)]
    for a, b in pairs:
        lower[max(a, b)].append(min(a, b))
    lvl = [0] * n
    for k in range(n):
        lvl[k] = 1 + max((lvl[o] for o in lower[k]), default=0)
    return max(lvl) if n else 0, lvl


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    policy = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    prob = synth.CONFIGS["S-mid"]()
    q = prob["q"]
    eng = BensonEngine(prob, eps=1e-7, pool_slots=4 * B + 64)
    if policy != 1:
        eng.set_policy(policy)
    assert eng.start() == 0
    for _ in range(12):
        eng.step(B)
    nl, nt = eng.collect(B)
    rec, piv, ls = eng.solve_local(nl)
    add = rec[:, 2] != 0
    ys = rec[add, 4:4 + q]
    key = np.round(ys / np.maximum(1.0, np.abs(ys).max(axis=1, keepdims=True)) * 1e11).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    first.sort()
    ys = ys[first]
    C = len(ys)
    print("batch %d LPs, %d with z > eps, %d distinct cuts, pivots %d" % (nl, add.sum(), C, piv))
    # lowerV2upperH with c = 1 (bslv_algs.c:287-305)
    hps = np.zeros((C, q + 1))
    hps[:, :q - 1] = ys[:, :q - 1]
    hps[:, q - 1] = 1.0 - ys[:, :q - 1].sum(axis=1)
    hps[:, q] = ys[:, q - 1]
    zero_w = (np.abs(hps[:, :q]) < 1e-9).sum(axis=1)
    print("cuts with k zero weights:", np.bincount(zero_w, minlength=q + 1))
    d = eng.poly_dump()
    X, used, ideal, E = d["X"], d["pu"].astype(bool), d["pi"].astype(bool), d["E"]
    s = X @ hps[:, :q].T                      # nv x C
    a = np.where(ideal[:, None], 0.0, hps[None, :, q])
    cls = np.where(s > a + 1e-9, 1, np.where(s > a - 1e-9, 0, -1)).astype(np.int8)
    cls[~used] = 1
    minus = (cls == -1).sum(axis=0); zero = (cls == 0).sum(axis=0)
    print("live %d (directions %d), edges %d" % (used.sum(), (used & ideal).sum(), len(E)))
    print("per cut: MINUS mean %.1f max %d | ZERO mean %.1f max %d | redundant now %d" % (minus.mean(), minus.max(), zero.mean(), zero.max(), (minus == 0).sum()))
    nonplus = cls != 1
    tc = nonplus.sum(axis=1)
    print("elements touched by >=1 cut: %d, by >=2: %d, max touch %d; touched directions: %d" % ((tc > 0).sum(), (tc > 1).sum(), tc.max(), ((tc > 0) & ideal).sum()))
    hot = np.nonzero(tc > 0)[0]
    for name, rule in (("a", 0), ("b", 1), ("c", 2)):
        pairs = set()
        for i in hot:
            if rule == 2 and ideal[i]:
                continue
            cs = np.nonzero(nonplus[i])[0]
            if len(cs) < 2:
                continue
            for x in range(len(cs)):
                for y in range(x + 1, len(cs)):
                    if rule >= 1 and cls[i, cs[x]] == 0 and cls[i, cs[y]] == 0:
                        continue
                    pairs.add((cs[x], cs[y]))
        hotmask = tc > 0
        Eh = E[hotmask[E[:, 0]] & hotmask[E[:, 1]]]
        for u, v in Eh:
            if rule == 2 and (ideal[u] or ideal[v]):
                continue
            cu, cv = np.nonzero(nonplus[u])[0], np.nonzero(nonplus[v])[0]
            for x in cu:
                for y in cv:
                    if x != y:
                        if rule >= 1 and cls[u, x] == 0 and cls[v, y] == 0:
                            continue
                        pairs.add((min(x, y), max(x, y)))
        dp, lvl = depth(C, pairs)
        print("rule (%s): %d conflict pairs of %d, greedy independent set %d, rounds (static upper bound) %d; local-minima rule: first set %d, rounds %d" % (
            name, len(pairs), C * (C - 1) // 2, greedy(C, pairs), rounds(C, pairs), sum(1 for x in lvl if x == 1), dp))
        if name == "a":
            rng = np.random.default_rng(1)
            nb = [[] for _ in range(C)]
            for x, y in pairs:
                nb[x].append(y); nb[y].append(x)
            deg = np.array([len(v) for v in nb])
            print("   degree: mean %.1f max %d" % (deg.mean(), deg.max()))
            for label, mk in (("fixed random priority", lambda r, alive: pri0), ("fresh random priority per round", lambda r, alive: rng.permutation(C)),
                              ("smallest remaining degree first (ties: index)", None)):
                pri0 = rng.permutation(C)
                alive = np.ones(C, bool)
                r = 0
                sizes = []
                while alive.any():
                    if mk is None:
                        dg = np.array([sum(1 for o in nb[k] if alive[o]) for k in range(C)])
                        pri = dg * C + np.arange(C)
                    else:
                        pri = mk(r, alive)
                    sel = [k for k in range(C) if alive[k] and all((not alive[o]) or pri[o] > pri[k] for o in nb[k])]
                    for k in sel:
                        alive[k] = False
                    sizes.append(len(sel))
                    r += 1
                print("   local minima under %s: %d rounds, first sets %s" % (label, r, sizes[:6]))
            tcs = tc[hot].astype(np.int64)
            ev = sum(int(tc[u]) * int(tc[v]) for u, v in Eh)
            print("   pair enumerations: elements %d, hot-hot edges %d (%d edges)" % ((tcs * (tcs - 1) // 2).sum(), ev, len(Eh)))
            for chunk in (256, 512):
                sub = [(a, b) for a, b in pairs if a < chunk and b < chunk]
                print("   first %d cuts only: greedy rounds %d, local-minima rounds %d" % (chunk, rounds(chunk, sub), depth(chunk, sub)[0]))
    eng.close()


if __name__ == "__main__":
    main()
