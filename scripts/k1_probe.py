"""K1 incidence kernel (k_classify_batch) roofline probe: Nv synthetic points x B halfspaces.
algorithmic bytes: 8 q Nv + Nv (flags, 1 B here; SURVEY counts Nv/8) + 8 (q+1) B + Nv B / 4"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bensolve_amd.poly import PolyEngine
q = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
G = PolyEngine(q)
G.bench_fill(nv)
rng = np.random.default_rng(0)
rows = []
for B in (8, 16, 32, 64, 128, 256, 512):
    hps = np.hstack([rng.normal(size=(B, q)), rng.normal(size=(B, 1))])
    _, anym, ms = G.classify_batch(hps, repeats=20, fetch=False)
    alg = 8.0 * q * nv + nv / 8.0 + 8.0 * (q + 1) * B + nv * B / 4.0
    flops = 2.0 * q * nv * B
    rows.append(dict(q=q, nv=nv, B=B, ms=round(ms, 4), alg_GBps=round(alg / ms / 1e6, 1), frac_hbm=round(alg / ms / 1e6 / 8000, 4),
                     TFLOPs=round(flops / ms / 1e9, 2)))
    print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/k1_probe_q%d.json" % q, "w"), indent=1)
