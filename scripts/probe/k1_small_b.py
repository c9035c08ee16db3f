"""K1 at B = 1..8 (the arithmetic vanishes: what the loads and the store alone cost) and the copy-rate yardstick of the same bytes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bensolve_amd.poly import PolyEngine
q, nv = 5, 8_000_000
G = PolyEngine(q); G.bench_fill(nv)
rng = np.random.default_rng(0)
for B in [int(b) for b in os.environ.get("BS", "2,4,8,12,16,24,32").split(",")]:
    hps = np.hstack([rng.normal(size=(B, q)), rng.normal(size=(B, 1))])
    _, _, ms = G.classify_batch(hps, repeats=20, fetch=False)
    alg = 8.0 * q * nv + nv / 8.0 + 8.0 * (q + 1) * B + nv * B / 4.0
    print(dict(B=B, ms=round(ms, 4), alg_GBps=round(alg / ms / 1e6, 1), real_GBps=round((8.0 * q * nv + nv + 8 * nv) / ms / 1e6, 1)), flush=True)
# yardstick: torch copy of 49 B per element (read 41, write 8 -> here simply a 24.5 B/elt copy = same total traffic)
a = torch.empty(int(nv * 24.5) // 8, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(dict(copy_ms=round(ms, 4), GBps=round(2 * a.numel() * 8 / ms / 1e6, 1)))
