#!/usr/bin/env python3
"""select_sim.py -- OFFLINE study of batch selection rules on a snapshot written by state_dump.py (CPU only, numpy/scipy).

For a rule that picks B vertices of the unprocessed queue: how many distinct cuts their LPs return, how many of them would
be redundant once the others are applied (estimated: a cut whose MINUS set is covered ... not modelled), the conflict graph
of the cuts (two cuts conflict when an element is non-PLUS for both or an edge joins their non-PLUS regions), and how many
rounds of mutually independent cuts the batch needs under (a) the local-minima rule of round 2 and (b) a maximal
independent set per round (greedy by priority = Luby to convergence).
"""
import argparse
import sys
import time

import numpy as np
import scipy.sparse as sp

EPS = 1e-9


def halfspaces(ystar, c):
    q = ystar.shape[1]
    hp = np.zeros((len(ystar), q + 1))
    hp[:, :q - 1] = ystar[:, :q - 1]
    hp[:, q - 1] = 1.0 - (ystar[:, :q - 1] * c[:q - 1]).sum(1)
    hp[:, q] = ystar[:, q - 1]
    return hp


def dedupe(ystar):
    sc = np.maximum(1.0, np.abs(ystar).max(1, keepdims=True))
    key = np.round(ystar / sc * 1e11).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    return np.sort(first), inv


def touch_matrix(X, ideal, hp):
    """sparse (nlive x ncuts) int8: 1 = MINUS, 2 = ZERO (non-PLUS)"""
    q = X.shape[1]
    rows, cols, vals = [], [], []
    for b0 in range(0, len(hp), 256):
        H = hp[b0:b0 + 256]
        S = X @ H[:, :q].T                                   # nlive x nb
        alpha = np.where(ideal[:, None], 0.0, H[None, :, q])
        minus = S < alpha - EPS
        zero = (~minus) & (S <= alpha + EPS)
        r, c = np.nonzero(minus)
        rows.append(r); cols.append(c + b0); vals.append(np.ones(len(r), np.int8))
        r, c = np.nonzero(zero)
        rows.append(r); cols.append(c + b0); vals.append(np.full(len(r), 2, np.int8))
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(X), len(hp)))


def conflict_graph(T, E):
    """boolean (ncuts x ncuts) sparse: share a non-PLUS element, or an edge joins their non-PLUS regions"""
    Tb = (T != 0).astype(np.int32)
    C1 = (Tb.T @ Tb)
    n = T.shape[0]
    A = sp.csr_matrix((np.ones(len(E), np.int32), (E[:, 0], E[:, 1])), shape=(n, n))
    A = A + A.T
    # only rows that are touched matter
    NT = A @ Tb                                              # for element u: cuts touching a neighbour of u
    C2 = (Tb.T @ NT)
    C = ((C1 + C2) != 0).tolil()
    C.setdiag(False)
    return C.tocsr()


def rounds_local_minima(C, prio, redundant_free=True):
    """round-2 rule: a cut is selected when it has the highest priority (lowest prio value) among its alive neighbours"""
    n = C.shape[0]
    alive = np.ones(n, bool)
    rounds, sizes = 0, []
    indptr, indices = C.indptr, C.indices
    while alive.any():
        sel = []
        for k in np.nonzero(alive)[0]:
            nb = indices[indptr[k]:indptr[k + 1]]
            nb = nb[alive[nb]]
            if len(nb) == 0 or prio[k] < prio[nb].min():
                sel.append(k)
        alive[sel] = False
        rounds += 1
        sizes.append(len(sel))
    return rounds, sizes


def rounds_mis(C, prio):
    """a maximal independent set per round: greedy in priority order"""
    n = C.shape[0]
    alive = np.ones(n, bool)
    rounds, sizes = 0, []
    indptr, indices = C.indptr, C.indices
    order = np.argsort(prio)
    while alive.any():
        blocked = np.zeros(n, bool)
        sel = []
        for k in order:
            if not alive[k] or blocked[k]:
                continue
            sel.append(k)
            blocked[indices[indptr[k]:indptr[k + 1]]] = True
        alive[sel] = False
        rounds += 1
        sizes.append(len(sel))
    return rounds, sizes


def rounds_luby(C, prio, iters):
    """`iters` Luby iterations per round with the SAME priorities: iteration 1 = local minima among alive; later iterations =
    local minima among the alive cuts that are not adjacent to a cut selected in this round"""
    n = C.shape[0]
    alive = np.ones(n, bool)
    rounds, sizes = 0, []
    indptr, indices = C.indptr, C.indices
    while alive.any():
        cand = alive.copy()
        sel_all = []
        for it in range(iters):
            sel = []
            for k in np.nonzero(cand)[0]:
                nb = indices[indptr[k]:indptr[k + 1]]
                nb = nb[cand[nb]]
                if len(nb) == 0 or prio[k] < prio[nb].min():
                    sel.append(k)
            if not sel:
                break
            sel_all += sel
            cand[sel] = False
            for k in sel:
                cand[indices[indptr[k]:indptr[k + 1]]] = False
        alive[sel_all] = False
        rounds += 1
        sizes.append(len(sel_all))
    return rounds, sizes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("state")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--rules", default="newest,random,sib4,sib16")
    args = ap.parse_args()
    S = np.load(args.state)
    X, pu, pi, E = S["X"], S["pu"], S["pi"], S["E"]
    q_idx, q_par, lp_pos, rec, piv = S["q_idx"], S["q_par"], S["lp_pos"], S["lp_rec"], S["lp_piv"]
    c = S["c"]
    q = X.shape[1]
    live = np.nonzero(pu)[0]
    remap = -np.ones(len(pu), np.int64); remap[live] = np.arange(len(live))
    Xl, il = X[live], pi[live] != 0
    El = remap[E]
    assert (El >= 0).all()
    nq = len(q_idx)
    print("live %d, edges %d, queue %d, LPs solved %d (newest %d..)" % (len(live), len(E), nq, len(lp_pos), nq - 8192))
    add = rec[:, 2] != 0
    isnew = lp_pos >= nq - 8192
    print("newest 8192: add %.3f, z median %.2e, pivots %.2f | random sample of the rest: add %.3f, z median %.2e, pivots %.2f" % (
        add[isnew].mean(), np.median(rec[isnew, 3]), piv[isnew].mean(), add[~isnew].mean(), np.median(rec[~isnew, 3]), piv[~isnew].mean()))
    rng = np.random.default_rng(3)
    B = args.batch
    lp_par = q_par[lp_pos]
    for rule in args.rules.split(","):
        if rule == "newest":
            pick = np.nonzero(lp_pos >= nq - B)[0]
        elif rule == "random":
            pick = rng.choice(np.nonzero(~isnew)[0], B, replace=False)
        elif rule.startswith("sib"):
            cap = int(rule[3:])
            cnt = {}
            pick = []
            for k in np.argsort(-lp_pos):
                p = lp_par[k]
                if cnt.get(p, 0) >= cap:
                    continue
                cnt[p] = cnt.get(p, 0) + 1
                pick.append(k)
                if len(pick) >= B:
                    break
            pick = np.array(pick)
        elif rule.startswith("rsib"):                        # random sample, at most cap per parent
            cap = int(rule[4:])
            cnt = {}
            pick = []
            for k in rng.permutation(len(lp_pos)):
                p = lp_par[k]
                if cnt.get(p, 0) >= cap:
                    continue
                cnt[p] = cnt.get(p, 0) + 1
                pick.append(k)
                if len(pick) >= B:
                    break
            pick = np.array(pick)
        else:
            raise SystemExit("unknown rule " + rule)
        t0 = time.time()
        r = rec[pick]
        a = r[:, 2] != 0
        ys = r[a][:, 4:4 + q]
        first, inv = dedupe(ys)
        hp = halfspaces(ys[first], c)
        T = touch_matrix(Xl, il, hp)
        Tm = (T == 1)
        nminus = np.asarray(Tm.sum(0)).ravel()
        ntouch = np.asarray((T != 0).sum(0)).ravel()
        touched = np.asarray((T != 0).sum(1)).ravel()
        C = conflict_graph(T, El)
        deg = np.diff(C.indptr)
        prio = rng.permutation(C.shape[0])
        r_lm, s_lm = rounds_local_minima(C, prio)
        r_l2, s_l2 = rounds_luby(C, prio, 2)
        r_l3, s_l3 = rounds_luby(C, prio, 3)
        r_mis, s_mis = rounds_mis(C, prio)
        print("%-8s parents %4d | LPs %d add %d distinct %d (no MINUS now: %d) | MINUS/cut mean %.0f max %d | hot elements %d, max touch %d | conflict degree mean %.1f max %d | rounds: local minima %d (first %s), luby2 %d, luby3 %d, MIS %d (first %s) | %.1f s" % (
            rule, len(set(lp_par[pick].tolist())), len(pick), a.sum(), len(first), int((nminus == 0).sum()), nminus.mean(), nminus.max(), int((touched > 0).sum()), touched.max(),
            deg.mean(), deg.max(), r_lm, s_lm[:4], r_l2, r_l3, r_mis, s_mis[:4], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
