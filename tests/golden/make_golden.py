#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Runs ONLY in the build container (needs /root/reference
compiled into oracle/_ref by `make -C oracle ref`, and scipy).  Fixtures are DATA: inputs and
expected outputs, no reference text.

  poly_ref.npz   canonicalised results of the REFERENCE polyhedron engine (bslv_poly.c, unmodified,
                 via oracle/_ref/libref_poly.so) for fixed cut sequences
  lp_highs.json  optimal objective values of P2(v) instances from scipy/HiGHS (independent solver;
                 the reference's own LP solver, GLPK, is not installed: parity unpinned by the reference)
  hybrid_*.npz   outputs of oracle/_ref/bensolve_hybrid (reference driver + reference polyhedron
                 engine + oracle LP) on the ex/*.vlp suite and on small synthetic problems
"""
import itertools
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import poly_harness as ph          # noqa: E402
from bensolve_amd import synth     # noqa: E402
from bensolve_amd.lp import P2Model  # noqa: E402


def poly_cases():
    """name -> (dim, v2h, c, apex, vals, ideals, init_after)"""
    cases = {}
    for q, N, seed in [(2, 25, 9), (3, 300, 1), (4, 100, 3), (5, 120, 4), (6, 40, 6)]:
        cases["tangent_q%d_N%d" % (q, N)] = (q, 0, None, False, ph.tangent_halfspaces(q, N, seed), None, q + 3)
    for q in (3, 4, 5):
        cube = np.vstack([np.eye(q), -np.eye(q)])
        signs = np.array(list(itertools.product([-1, 1], repeat=q)), float)
        cases["cube_q%d" % q] = (q, 0, None, False, cube, None, None)
        cases["cube_trunc_q%d" % q] = (q, 0, None, False, np.vstack([cube, signs / (q - 2)]), None, None)
        cases["cube_support_q%d" % q] = (q, 0, None, False, np.vstack([cube, signs / q]), None, None)
        cases["cross_q%d" % q] = (q, 0, None, False, signs, None, None)
        cases["cross_trunc_q%d" % q] = (q, 0, None, False, np.vstack([signs, cube * 2.0]), None, None)
    # ordering cones of the example suite, given by generators (k lines of ex05 / ex08): cone_vertenum's sequence
    cases["cone_ex05"] = (3, 0, None, True, np.array([[2, 4, 0], [4, 0, 2], [2, 2, -1], [0, 2, 4.0]]).T[:0].reshape(0, 3), None, None)
    gens = read_cone("/root/reference/ex/ex05.vlp")
    cases["cone_ex05"] = (gens.shape[1], 0, None, True, gens, [1] * len(gens), None)
    gens = read_cone("/root/reference/ex/ex08.vlp")
    cases["cone_ex08"] = (gens.shape[1], 0, None, True, gens, [1] * len(gens), None)
    return cases


def read_cone(path):
    """generators (rows) from the k lines of a .vlp file (bslv_vlp.c:459-487); j=0 lines set c and are skipped"""
    q = ngen = None
    ent = []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "p":
            q, ngen = int(t[6]), int(t[9])
        elif t[0] == "k" and int(t[2]) > 0:
            ent.append((int(t[1]), int(t[2]), float(t[3])))
    G = np.zeros((ngen, q))
    for i, j, v in ent:
        G[j - 1, i - 1] = v
    return G


def run_case(kind, case):
    q, v2h, c, apex, vals, ideals, init_after = case
    P = ph.FlatPoly(kind, q, v2h, c)
    if apex:
        P.dual0_apex()
    rcs = ph.run_sequence(P, vals, ideals, init_after)
    P.dual_adjacency()
    can = ph.canonical(P.dump())
    P.close()
    return rcs, can


def pack(can):
    return dict(X=can["X"], pi=can["pi"], Y=can["Y"], di=can["di"],
                E=np.array(sorted(can["E"]), np.int64).reshape(-1, 2), I=np.array(sorted(can["I"]), np.int64).reshape(-1, 2),
                DE=np.array(sorted(can["DE"]), np.int64).reshape(-1, 2))


def make_poly():
    out = {}
    for name, case in poly_cases().items():
        rcs, can = run_case("ref", case)
        q, v2h, c, apex, vals, ideals, init_after = case
        out[name + "/in_vals"] = np.asarray(vals, float)
        out[name + "/in_ideals"] = np.asarray([0] * len(vals) if ideals is None else ideals)
        out[name + "/in_meta"] = np.array([q, v2h, int(apex), -1 if init_after is None else init_after])
        out[name + "/rc"] = np.asarray(rcs)
        for k, v in pack(can).items():
            out[name + "/" + k] = v
        print("poly", name, "primal", len(can["X"]), "dual", len(can["Y"]))
    np.savez_compressed(os.path.join(HERE, "poly_ref.npz"), **out)


def snap_cases():
    """Cut sequences with ONE crafted cut whose hyperplane passes delta above a live vertex that has a neighbour the cut removes
    (bslv_poly.c:666-674: for delta in (1e-11, 1e-9] poly__cut moves that vertex onto the hyperplane; below 1e-11 it does not), followed
    by ordinary cuts that meet the moved vertex.  The crafted cut is found on the REFERENCE's polyhedron, so only this script needs it."""
    cases = {}
    for q, N, seed in [(3, 40, 5), (4, 40, 6)]:
        D = ph.tangent_halfspaces(q, N, seed)
        P = ph.FlatPoly("ref", q, 0, None)
        ph.run_sequence(P, D, None, q + 3)
        d = P.dump()
        P.close()
        live = np.nonzero(d["pu"].astype(bool) & ~d["pi"].astype(bool))[0]
        tail = ph.tangent_halfspaces(q, 12, seed + 100) * 0.97        # (a little deeper than the first N: they cut near the moved vertex too)
        for delta in (5e-10, 5e-11, 5e-12):
            rng = np.random.default_rng(seed)
            v = None
            for i0 in live:
                nb = [b if a == i0 else a for a, b in d["E"] if i0 in (a, b)]
                x0 = d["X"][i0]
                for _ in range(200):
                    w = rng.normal(size=q)
                    if w @ x0 >= -1e-3:
                        continue
                    cand = w * ((-1 + delta) / (w @ x0))
                    if any(d["pu"][n] and d["X"][n] @ cand < -1 - 1e-3 for n in nb):
                        v = cand
                        break
                if v is not None:
                    break
            assert v is not None
            assert abs((d["X"][i0] @ v + 1) - delta) < 1e-13
            cases["snap_q%d_delta%.0e" % (q, delta)] = (q, 0, None, False, np.vstack([D, v[None], tail]), None, q + 3)
    return cases


def snap_dir_cases():
    """As snap_cases, for a DIRECTION (an ideal element: alpha = 0 in poly__cut's comparisons, bslv_poly.c:596,666): halfspaces with
    positive normals leave an unbounded polyhedron with extreme directions; the crafted cut has w.r = delta for one of them and removes a
    neighbour of it."""
    cases = {}
    q, N, seed = 3, 30, 7
    rng = np.random.default_rng(seed)
    D = np.abs(rng.normal(size=(N, q)))
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    P = ph.FlatPoly("ref", q, 0, None)
    ph.run_sequence(P, D, None, q + 3)
    d = P.dump()
    P.close()
    pu, pi = d["pu"].astype(bool), d["pi"].astype(bool)
    tail = np.abs(np.random.default_rng(seed + 100).normal(size=(8, q)))
    tail = tail / np.linalg.norm(tail, axis=1, keepdims=True) * 0.97
    for delta in (5e-10, 5e-11, 5e-12):
        rng = np.random.default_rng(seed + 1)
        w = None
        for i0 in np.nonzero(pu & pi)[0]:
            r0 = d["X"][i0]
            nb = [b if a == i0 else a for a, b in d["E"] if i0 in (a, b)]
            for _ in range(500):
                w0 = rng.normal(size=q)
                cand = w0 - ((w0 @ r0 - delta) / (r0 @ r0)) * r0
                removes = any(pu[n] and ((pi[n] and d["X"][n] @ cand < -1e-3) or (not pi[n] and d["X"][n] @ cand < -1 - 1e-3)) for n in nb)
                if removes and abs(cand @ r0 - delta) < 1e-15:
                    w = cand
                    break
            if w is not None:
                break
        assert w is not None
        cases["snapdir_q%d_delta%.0e" % (q, delta)] = (q, 0, None, False, np.vstack([D, w[None], tail]), None, q + 3)
    return cases


def make_poly_snap():
    out = {}
    for name, case in snap_dir_cases().items():
        rcs, can = run_case("ref", case)
        q, v2h, c, apex, vals, ideals, init_after = case
        out[name + "/in_vals"] = np.asarray(vals, float)
        out[name + "/in_ideals"] = np.asarray([0] * len(vals))
        out[name + "/in_meta"] = np.array([q, v2h, int(apex), init_after])
        out[name + "/rc"] = np.asarray(rcs)
        for k, v in pack(can).items():
            out[name + "/" + k] = v
        print("poly snap (direction)", name, "primal", len(can["X"]), "dual", len(can["Y"]))
    np.savez_compressed(os.path.join(HERE, "poly_ref_snap_dirs.npz"), **out)
    out = {}
    for name, case in snap_cases().items():
        rcs, can = run_case("ref", case)
        q, v2h, c, apex, vals, ideals, init_after = case
        out[name + "/in_vals"] = np.asarray(vals, float)
        out[name + "/in_ideals"] = np.asarray([0] * len(vals))
        out[name + "/in_meta"] = np.array([q, v2h, int(apex), init_after])
        out[name + "/rc"] = np.asarray(rcs)
        for k, v in pack(can).items():
            out[name + "/" + k] = v
        print("poly snap", name, "primal", len(can["X"]), "dual", len(can["Y"]))
    np.savez_compressed(os.path.join(HERE, "poly_ref_snap.npz"), **out)


def read_objective_columns(path):
    """columns of P (the o lines of a .vlp file, bslv_vlp.c:434-457) as rows"""
    q = n = None
    ent = []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "p":
            n, q = int(t[4]), int(t[6])
        elif t[0] == "o":
            ent.append((int(t[1]), int(t[2]), float(t[3])))
    P = np.zeros((n, q))
    for k, j, v in ent:
        P[j - 1, k - 1] = v
    return P


def poly_cases_large():
    """SURVEY.md 8c's list: the random-tangent synthetics at the sizes of BASELINE.md section 2, the dual cone of ex06 and the
    lattice directions of ex10 (its 343 objective columns, integers in {-3..3}^3: many parallel and cohyperplanar cuts)."""
    cases = {}
    for q, N, seed in [(3, 2000, 31), (5, 200, 32), (5, 1000, 33), (8, 60, 34)]:
        cases["tangent_q%d_N%d" % (q, N)] = (q, 0, None, False, ph.tangent_halfspaces(q, N, seed), None, q + 3)
    gens = read_cone("/root/reference/ex/ex06.vlp")
    cases["cone_ex06"] = (gens.shape[1], 0, None, True, gens, [1] * len(gens), None)
    L = read_objective_columns("/root/reference/ex/ex10.vlp")
    L = L[np.abs(L).sum(axis=1) > 0]
    cases["lattice_ex10"] = (3, 0, None, False, L / 3.0, None, None)
    return cases


def digest_pairs(pairs):
    """SHA-256 of a canonical index set (sorted pairs, int64 little endian)"""
    import hashlib
    a = np.array(sorted(pairs), np.int64).reshape(-1, 2)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(), len(a)


def make_poly_large():
    """coordinates in full, index sets as (SHA-256, count) of the canonical relabelling: see tests/test_oracle_poly.py"""
    out, meta = {}, {}
    for name, case in poly_cases_large().items():
        rcs, can = run_case("ref", case)
        q, v2h, c, apex, vals, ideals, init_after = case
        out[name + "/in_vals"] = np.asarray(vals, float)
        out[name + "/in_ideals"] = np.asarray([0] * len(vals) if ideals is None else ideals)
        out[name + "/in_meta"] = np.array([q, v2h, int(apex), -1 if init_after is None else init_after])
        out[name + "/rc"] = np.asarray(rcs)
        for k in ("X", "pi", "Y", "di"):
            out[name + "/" + k] = can[k]
        meta[name] = {k: digest_pairs(can[k]) for k in ("E", "I", "DE")}
        print("poly large", name, "primal", len(can["X"]), "dual", len(can["Y"]), meta[name])
    np.savez_compressed(os.path.join(HERE, "poly_ref_large.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "poly_ref_large.json"), "w"), indent=1)


def make_lp():
    from scipy.optimize import linprog
    rec = []
    for (m, n, q, seed, B) in [(12, 6, 2, 3, 6), (30, 15, 3, 5, 8), (60, 30, 3, 7, 8), (40, 20, 4, 9, 8)]:
        prob = synth.covering_vlp(m, n, q, seed)
        model = P2Model(prob)
        rng = np.random.default_rng(seed)
        X = rng.random((B, n)) * (3.0 / n) + 1.0 / n
        V = (X @ prob["P"].T) * rng.uniform(0.2, 1.2, size=(B, 1))
        for v in V:
            # min z  s.t. A x >= 1, x >= 0, P x - z <= v
            c = np.zeros(n + 1); c[n] = 1
            Aub = np.vstack([np.hstack([-prob["A"], np.zeros((m, 1))]), np.hstack([prob["P"], -np.ones((q, 1))])])
            bub = np.concatenate([-np.ones(m), v])
            res = linprog(c, A_ub=Aub, b_ub=bub, bounds=[(0, None)] * n + [(None, None)], method="highs")
            assert res.status == 0
            rec.append(dict(m=m, n=n, q=q, seed=seed, v=list(map(float, v)), obj=float(res.fun)))
    json.dump(rec, open(os.path.join(HERE, "lp_highs.json"), "w"), indent=0)
    print("lp goldens", len(rec))


def run_hybrid(vlp_path, args):
    exe = os.path.join(ROOT, "oracle", "_ref", "bensolve_hybrid")
    with tempfile.TemporaryDirectory() as td:
        base = os.path.join(td, "out")
        p = subprocess.run([exe, vlp_path, "-m", "0", "-o", base] + args, capture_output=True, text=True, timeout=600)
        res = dict(rc=p.returncode, stdout=p.stdout[-300:])
        for suf in ("img_p", "img_d", "adj_p", "inc_p"):
            f = base + "_" + suf + ".sol"
            if os.path.exists(f):
                res[suf] = open(f).read()
        return res


def parse_img(txt):
    a = np.array([[float(x) for x in l.split()] for l in txt.strip().splitlines()])
    return a[:, 0].astype(int), a[:, 1:]


def make_hybrid():
    out = {}
    status = {}
    for ex in ("ex01", "ex02", "ex03", "ex04", "ex05", "ex06", "ex08", "ex11"):
        r = run_hybrid("/root/reference/ex/%s.vlp" % ex, [])
        status[ex] = dict(rc=r["rc"], msg=r["stdout"].strip().splitlines()[-1] if r["stdout"].strip() else "")
        if "img_p" in r:
            t, X = parse_img(r["img_p"]); out[ex + "/p_type"] = t; out[ex + "/p"] = X
            t, Y = parse_img(r["img_d"]); out[ex + "/d_type"] = t; out[ex + "/d"] = Y
        print("hybrid", ex, status[ex])
    with tempfile.TemporaryDirectory() as td:
        for name, prob in (("syn_30x15_q3_s5", synth.covering_vlp(30, 15, 3, 5)), ("syn_60x30_q3_s7", synth.covering_vlp(60, 30, 3, 7)),
                           ("syn_40x20_q4_s9", synth.covering_vlp(40, 20, 4, 9))):
            path = os.path.join(td, name + ".vlp")
            synth.write_vlp(prob, path)
            r = run_hybrid(path, ["-b"])
            t, X = parse_img(r["img_p"]); out[name + "/p_type"] = t; out[name + "/p"] = X
            t, Y = parse_img(r["img_d"]); out[name + "/d_type"] = t; out[name + "/d"] = Y
            print("hybrid", name, X.shape, Y.shape)
    np.savez_compressed(os.path.join(HERE, "hybrid.npz"), **out)
    json.dump(status, open(os.path.join(HERE, "hybrid_status.json"), "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        make_poly_large()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "snap":
        make_poly_snap()
        sys.exit(0)
    make_poly()
    make_poly_snap()
    make_poly_large()
    make_lp()
    make_hybrid()
