#!/usr/bin/env python3
"""Soak test of the execution modes of the cut pipeline on the bench workload: K batched Benson steps on S-mid in the
default mode and with every fast path turned off; the two polyhedra must be identical slot by slot.
usage: mode_soak.py [steps [workload [batch]]]"""
import os, sys, json, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps, workload="S-mid", B=1024):
    import numpy as np
    from bensolve_amd import synth
    from bensolve_amd.benson import BensonEngine
    eng = BensonEngine(synth.CONFIGS[workload](), eps=1e-7, pool_slots=4 * B + 64)
    assert eng.start() == 0
    tot = dict(lps=0, cuts=0)
    trace = []
    for _ in range(steps):
        nl, nt = eng.collect(B, 0, 1)
        rec, piv, ls = eng.solve_local(nl)
        s = eng.apply(rec)
        tot["lps"] += s["lps"]; tot["cuts"] += s["cuts"]
        c = eng.poly_call("counts")
        trace.append((s["lps"], s["cuts"], c["nprimal"], c["nedges"], c["ndual"]))
    D = eng.poly_dump()
    h = hashlib.sha256()
    for k in ("pu", "pi", "ps", "X", "du", "di", "Y", "E", "I"):
        h.update(np.ascontiguousarray(D[k]).tobytes())
    out = dict(tot, slots=int(len(D["pu"])), live=int(D["pu"].sum()), edges=int(len(D["E"])), facets=int(D["du"].sum()), sha256=h.hexdigest(),
               paths=eng.poly_call("path_stats"), trace=trace)
    eng.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        print(json.dumps(run(int(sys.argv[1]), sys.argv[3], int(sys.argv[4]))))
        sys.exit(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    workload = sys.argv[2] if len(sys.argv) > 2 else "S-mid"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    res = {}
    for name, env in (("default", {}), ("conservative", {"BSLV_NO_SPEC": "1", "BSLV_NO_HOT": "1"}), ("fallback_prune", {"BSLV_K2_LDS": "64"}),
                      ("decline", {"BSLV_CROSS_UB": "100"}), ("decline_fallback_nohot", {"BSLV_CROSS_UB": "100", "BSLV_K2_LDS": "64", "BSLV_NO_HOT": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(steps), "child", workload, str(B)], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
        print(name, {k: v for k, v in res[name].items() if k != "trace"}, flush=True)
    for k in res:
        for i, (a, b) in enumerate(zip(res["default"]["trace"], res[k]["trace"])):
            if a != b:
                print("first difference default vs", k, "at step", i, a, b)
                break
    ok = all(res[k]["sha256"] == res["default"]["sha256"] for k in res)
    print("IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
