#!/usr/bin/env python3
"""state_dump.py -- a snapshot of an S-mid run for OFFLINE study of batch selection rules (scripts/probe/select_sim.py):
the live polyhedron (coordinates, edges), the whole unprocessed queue with parent facets, and the cut that P2(v) returns
for a large sample of the queue (the newest ones and a random sample of the rest), solved but NOT applied.

    python scripts/probe/state_dump.py --out gpurun_out/state.npz [--steps 10] [--newest 8192] [--sample 24576]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/state.npz")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--newest", type=int, default=8192)
    ap.add_argument("--sample", type=int, default=24576)
    ap.add_argument("--pool", type=int, default=45056)
    ap.add_argument("--workload", default="S-mid")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available()
    from bensolve_amd import synth
    from bensolve_amd.benson import BensonEngine
    from front_probe import window
    prob = synth.CONFIGS[args.workload]()
    eng = BensonEngine(prob, eps=1e-7, pool_slots=args.pool)
    assert eng.start() == 0
    lib = eng.lib
    lib.bslv_benson_collect_given.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.bslv_benson_last_local.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
    B = args.batch
    for _ in range(200):
        eng.step(B)
        if eng.poly_call("unprocessed", 0)[3] >= B:
            break
    for _ in range(args.steps):
        eng.step(B)
    idx, val, ideal, par, cnt = window(eng, 1 << 30)
    keep = ideal == 0
    idx, val, par = idx[keep], val[keep], par[keep]
    n = len(idx)
    rng = np.random.default_rng(11)
    newest = np.arange(max(0, n - args.newest), n)
    rest = rng.choice(max(1, n - args.newest), size=min(args.sample, max(0, n - args.newest)), replace=False) if n > args.newest else np.zeros(0, np.int64)
    pos = np.unique(np.concatenate([newest, rest]))
    recs, pivs, srcs = [], [], []
    for b0 in range(0, len(pos), 4096):
        p = pos[b0:b0 + 4096]
        bi = np.ascontiguousarray(idx[p]); bv = np.ascontiguousarray(val[p]); bp = np.ascontiguousarray(par[p])
        nl, nt = ctypes.c_int(), ctypes.c_int()
        rc = lib.bslv_benson_collect_given(eng.h, len(bi), bi.ctypes.data, bv.ctypes.data, bp.ctypes.data, 0, 1, ctypes.byref(nl), ctypes.byref(nt))
        assert rc == 0
        rec, piv, ls = eng.solve_local(nl.value)
        src = np.zeros(nl.value, np.int32); pv = np.zeros(nl.value, np.int32); gen = np.zeros(nl.value, np.int32)
        lib.bslv_benson_last_local(eng.h, nl.value, src.ctypes.data, pv.ctypes.data, gen.ctypes.data)
        recs.append(rec.copy()); pivs.append(pv); srcs.append(src)
        print("solved %d LPs, pivots/LP %.2f, from the root %d" % (nl.value, pv.mean(), int((src == 0).sum())), flush=True)
    rec = np.concatenate(recs); piv = np.concatenate(pivs); src = np.concatenate(srcs)
    d = eng.poly_dump()
    live = d["pu"] != 0
    print("live %d of %d slots, edges %d, queue %d, LPs %d" % (live.sum(), len(live), len(d["E"]), n, len(rec)), flush=True)
    np.savez_compressed(args.out, X=d["X"], pu=d["pu"], pi=d["pi"], ps=d["ps"], E=d["E"], Y=d["Y"], du=d["du"],
                        q_idx=idx, q_par=par, lp_pos=pos, lp_rec=rec, lp_piv=piv, lp_src=src, c=np.ones(eng.q))
    eng.close()


if __name__ == "__main__":
    main()
