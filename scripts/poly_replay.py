#!/usr/bin/env python3
"""Replays the cut sequence of a batched Benson run (S-mid) through the polyhedron engine alone, one cut at a time, under
different execution modes, and through the CPU oracle; reports where the dumps first differ.
usage: poly_replay.py record <steps> <file.npz> | compare <file.npz> <ncuts> [oracle]"""
import os, sys, json, hashlib, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

KEYS = ("pu", "pi", "X", "du", "di", "E", "I")


def digest(D):
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(D[k]).tobytes())
    return h.hexdigest()[:16]


def record(steps, path):
    from bensolve_amd import synth
    from bensolve_amd.benson import BensonEngine
    prob = synth.CONFIGS["S-mid"]()
    eng = BensonEngine(prob, eps=1e-7, pool_slots=4160)
    assert eng.start() == 0
    for _ in range(steps):
        nl, nt = eng.collect(1024, 0, 1)
        rec, piv, ls = eng.solve_local(nl)
        eng.apply(rec)
    D = eng.poly_dump()
    np.savez_compressed(path, Y=D["Y"], q=prob["q"], c=np.asarray(prob.get("c", np.ones(prob["q"]))))
    print("recorded", len(D["Y"]), "dual vertices")


def replay_gpu(path, ncuts):
    from bensolve_amd.poly import PolyEngine
    Z = np.load(path)
    Y, q, c = Z["Y"], int(Z["q"]), Z["c"]
    G = PolyEngine(q, 1, c)
    G.set_batch_mode(0)
    for k in range(1, q + 1):
        G.add(Y[k], 0)
    assert G.init() == 0
    rest = Y[q + 1:q + 1 + ncuts]
    rcs = []
    for b0 in range(0, len(rest), 256):
        rcs += list(G.add_cuts(rest[b0:b0 + 256], None))
    D = G.dump()
    out = dict(digest=digest(D), live=int(D["pu"].sum()), slots=int(len(D["pu"])), edges=int(len(D["E"])), applied=int(len(rcs) - sum(rcs)), paths=G.path_stats())
    G.close()
    return out, D


def replay_oracle(path, ncuts):
    import poly_harness as ph
    Z = np.load(path)
    Y, q, c = Z["Y"], int(Z["q"]), Z["c"]
    O = ph.FlatPoly("oracle", q, 1, c)
    for k in range(1, q + 1):
        O.add(Y[k], 0)
    assert O.init() == 0
    for y in Y[q + 1:q + 1 + ncuts]:
        O.add(y, 0)
    D = O.dump()
    O.close()
    return dict(digest=digest(D), live=int(D["pu"].sum()), slots=int(len(D["pu"])), edges=int(len(D["E"]))), D


if __name__ == "__main__":
    if sys.argv[1] == "record":
        record(int(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == "child":
        out, _ = replay_gpu(sys.argv[2], int(sys.argv[3]))
        print(json.dumps(out))
    else:
        path, n = sys.argv[2], int(sys.argv[3])
        for name, env in (("fused", {}), ("multi", {"BSLV_K2_LDS": "64"})):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path, str(n)], env=e, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            print(name, r.stdout.strip().splitlines()[-1], flush=True)
        if len(sys.argv) > 4:
            out, _ = replay_oracle(path, n)
            print("oracle", json.dumps(out))
