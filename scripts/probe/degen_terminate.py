"""S-degenerate family (unit cube + integer cover rows, integer lattice objectives): which members run to TERMINATION on one GPU?
usage: degen_terminate.py "q,n,m,batch,eps;q,n,m,batch,eps;..." [time cap per member in s]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
cap = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
rows = []
os.makedirs("gpurun_out", exist_ok=True)
for spec in sys.argv[1].split(";"):
    q, n, m, batch = [int(x) for x in spec.split(",")[:4]]
    eps = float(spec.split(",")[4]) if len(spec.split(",")) > 4 else 1e-9
    prob = synth.degenerate_vlp(m, n, q, 3)
    eng = BensonEngine(prob, eps=eps, pool_slots=max(4 * batch, 64))
    t0 = time.time()
    assert eng.start() == 0
    done, steps, s = False, 0, {}
    err = None
    try:
        while time.time() - t0 < cap:
            s = eng.step(batch); steps += 1
            if s["lps"] == 0 and s["left"] == 0:
                done = True; break
    except Exception as e:                      # capacity / memory: reported, the next member still runs
        err = str(e)[:160]
    t = time.time() - t0
    tot = eng.totals(); c = eng.poly_call("counts")
    row = dict(q=q, n=n, m=m, batch=batch, eps=eps, finished=done, seconds=round(t, 2), steps=steps, lps=tot["lps"], cuts=tot["cuts"], pivots=tot["pivots"],
               vertex_slots=c["nprimal"], facets=c["ndual"], edges=c["nedges"], left=s.get("left"), error=err)
    rows.append(row); print(json.dumps(row), flush=True)
    json.dump(rows, open("gpurun_out/degen_terminate.json", "w"), indent=1)
    eng.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/degen_terminate.json", "w"), indent=1)
