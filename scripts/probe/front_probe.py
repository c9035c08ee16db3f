#!/usr/bin/env python3
"""front_probe.py -- what a batch selection rule does to the two halves of an S-mid step.

The caller-chosen batch (bslv_benson_collect_given) lets selection rules be tried in numpy before one is built into the
driver: newest first (the round-2 default), a cap on the children of one cut, K fronts (regions of the image space, newest
first inside each).  Per rule: pivots per LP (mean / median / p90 / max, share started from the root tableau), redundant
share, cuts per pass, time of the two phases.

    python scripts/probe/front_probe.py [--steps 12] [--warm 6] [--rules newest,sib8,...]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def window(eng, W):
    lib, ph, q = eng.lib, eng._poly_h, eng.q
    lib.bslv_poly_unprocessed2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    cnt = ctypes.c_int()
    lib.bslv_poly_unprocessed2(ph, 0, 1, None, None, None, None, ctypes.byref(cnt))
    n = min(cnt.value, W)
    idx = np.empty(n, np.int32); val = np.empty((n, q)); ideal = np.empty(n, np.int32); par = np.empty(n, np.int32)
    if n:
        rc = lib.bslv_poly_unprocessed2(ph, n, 1, idx.ctypes.data, val.ctypes.data, ideal.ctypes.data, par.ctypes.data, ctypes.byref(cnt))
        assert rc == 0
    return idx, val, ideal, par, cnt.value


def pick(rule, idx, val, par, B, G):
    """positions (ascending) of the batch inside the window (ascending slot order, newest last)"""
    n = len(idx)
    if rule["kind"] == "newest":
        return np.arange(max(0, n - B), n)
    cap = rule.get("cap", 1 << 30)
    K = rule.get("K", 1)
    if K > 1:
        vn = val / np.maximum(np.abs(val).sum(1, keepdims=True), 1e-300)
        reg = np.argmax(vn @ G[:K].T, axis=1)
    else:
        reg = np.zeros(n, np.int32)
    quota = [B // K + (1 if k < B % K else 0) for k in range(K)]
    taken = [0] * K
    per_parent = {}
    out = []
    spill = []
    for k in range(n - 1, -1, -1):
        r = reg[k]
        p = par[k]
        c = per_parent.get(p, 0)
        if c >= cap:
            continue
        if taken[r] >= quota[r]:
            if len(spill) < B:
                spill.append(k)
            continue
        per_parent[p] = c + 1
        taken[r] += 1
        out.append(k)
        if len(out) >= B:
            break
    for k in spill:                      # regions that ran dry: fill up from the others (newest first)
        if len(out) >= B:
            break
        out.append(k)
    return np.array(sorted(out), np.int64)


def run(rule, args, prob):
    import torch
    from bensolve_amd.benson import BensonEngine
    B = args.batch
    eng = BensonEngine(prob, eps=1e-7, pool_slots=args.pool)
    assert eng.start() == 0
    lib = eng.lib
    lib.bslv_benson_collect_given.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.bslv_benson_last_local.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
    rng = np.random.default_rng(7)
    G = rng.normal(size=(64, eng.q))
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    # ramp as bench.py does: newest first until a full batch is there
    for _ in range(200):
        eng.step(B)
        if eng.poly_call("unprocessed", 0)[3] >= B:
            break
    rows = []
    tot = dict(lps=0, cuts=0, red=0, conf=0, piv=0, t_sel=0.0, t_lp=0.0, t_cut=0.0, rounds=0, root=0)
    allpiv = []
    r2s0 = ps0 = None
    for step in range(args.warm + args.steps):
        if step == args.warm:
            r2s0, ps0 = eng.poly_call("rounds2_stats"), eng.poly_call("path_stats")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, val, ideal, par, cnt = window(eng, args.window)
        if ideal.any():
            eng.poly_call("mark", idx[ideal != 0])
            keep = ideal == 0
            idx, val, par = idx[keep], val[keep], par[keep]
        pos = pick(rule, idx, val, par, B, G)
        bi = np.ascontiguousarray(idx[pos]); bv = np.ascontiguousarray(val[pos]); bp = np.ascontiguousarray(par[pos])
        nl, nt = ctypes.c_int(), ctypes.c_int()
        rc = lib.bslv_benson_collect_given(eng.h, len(bi), bi.ctypes.data, bv.ctypes.data, bp.ctypes.data, 0, 1, ctypes.byref(nl), ctypes.byref(nt))
        assert rc == 0, rc
        t1 = time.perf_counter()
        rec, piv, ls = eng.solve_local(nl.value)
        t2 = time.perf_counter()
        r0 = eng.poly_call("rounds_run")
        s = eng.apply(rec)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        r1 = eng.poly_call("rounds_run")
        src = np.zeros(nl.value, np.int32); pv = np.zeros(nl.value, np.int32); gen = np.zeros(nl.value, np.int32)
        lib.bslv_benson_last_local(eng.h, nl.value, src.ctypes.data, pv.ctypes.data, gen.ctypes.data)
        if step >= args.warm:
            tot["lps"] += s["lps"]; tot["cuts"] += s["cuts"]; tot["red"] += s["redundant"]; tot["conf"] += s["confirmed"]
            tot["piv"] += int(pv.sum()); tot["t_sel"] += t1 - t0; tot["t_lp"] += t2 - t1; tot["t_cut"] += t3 - t2; tot["rounds"] += r1 - r0
            tot["root"] += int((src == 0).sum())
            allpiv.append(pv)
            rows.append(dict(step=step, lps=s["lps"], parents=int(len(set(bp.tolist()))), cuts=s["cuts"], red=s["redundant"], conf=s["confirmed"],
                             piv_mean=round(float(pv.mean()), 2), piv_max=int(pv.max()), root=int((src == 0).sum()), gen_max=int(gen.max()),
                             rounds=int(r1 - r0), ms_lp=round((t2 - t1) * 1e3, 2), ms_cut=round((t3 - t2) * 1e3, 2), queue=cnt,
                             z_med=round(float(np.median(rec[:, 3])), 6)))
    ap = np.concatenate(allpiv) if allpiv else np.zeros(1)
    r2s1, ps1 = eng.poly_call("rounds2_stats"), eng.poly_call("path_stats")
    T = tot["t_lp"] + tot["t_cut"]            # (the numpy selection is not part of the product: left out of the rate)
    res = dict(rule=rule["name"], lps=tot["lps"], useful=tot["cuts"] + tot["conf"], redundant_frac=round(1 - (tot["cuts"] + tot["conf"]) / max(tot["lps"], 1), 3),
               lps_per_s=round(tot["lps"] / T, 0), useful_per_s=round((tot["cuts"] + tot["conf"]) / T, 0),
               piv_mean=round(float(ap.mean()), 2), piv_p50=float(np.median(ap)), piv_p90=float(np.percentile(ap, 90)), piv_p99=float(np.percentile(ap, 99)), piv_max=int(ap.max()),
               root_frac=round(tot["root"] / max(tot["lps"], 1), 4), cuts_per_pass=round(tot["cuts"] / max(tot["rounds"], 1), 2),
               ms_lp=round(tot["t_lp"] * 1e3 / args.steps, 2), ms_cut=round(tot["t_cut"] * 1e3 / args.steps, 2), us_per_cut=round(tot["t_cut"] * 1e6 / max(tot["cuts"], 1), 2),
               ms_select_numpy=round(tot["t_sel"] * 1e3 / args.steps, 2), pool=eng.pool_stats(), r2={k: r2s1[k] - r2s0[k] for k in r2s1}, paths={k: ps1[k] - ps0[k] for k in ps1}, health=eng.poly_call("rounds2_health"))
    eng.close()
    return res, rows


RULES = {
    "newest": dict(kind="newest"),
    "sib1": dict(kind="cap", cap=1), "sib2": dict(kind="cap", cap=2), "sib4": dict(kind="cap", cap=4), "sib8": dict(kind="cap", cap=8), "sib16": dict(kind="cap", cap=16), "sib32": dict(kind="cap", cap=32), "sib64": dict(kind="cap", cap=64),
    "f4": dict(kind="cap", K=4), "f8": dict(kind="cap", K=8), "f16": dict(kind="cap", K=16), "f32": dict(kind="cap", K=32),
    "f8s16": dict(kind="cap", K=8, cap=16), "f8s32": dict(kind="cap", K=8, cap=32), "f16s16": dict(kind="cap", K=16, cap=16), "f16s32": dict(kind="cap", K=16, cap=32),
    "f16s8": dict(kind="cap", K=16, cap=8), "f32s16": dict(kind="cap", K=32, cap=16), "f32s8": dict(kind="cap", K=32, cap=8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warm", type=int, default=6)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--pool", type=int, default=24576)
    ap.add_argument("--window", type=int, default=262144)
    ap.add_argument("--workload", default="S-mid")
    ap.add_argument("--rules", default="newest,sib16,f8,f8s16,f16s16")
    ap.add_argument("--rows", action="store_true")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available()
    from bensolve_amd import synth
    prob = synth.CONFIGS[args.workload]()
    for name in args.rules.split(","):
        rule = dict(RULES[name]); rule["name"] = name
        res, rows = run(rule, args, prob)
        print(json.dumps(res), flush=True)
        if args.rows:
            for r in rows:
                print("   ", json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
