"""dev probe: LP engine throughput on a synthetic config (not a test, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bensolve_amd import synth
from bensolve_amd.lp import P2Model, LpEngine

cfg = sys.argv[1] if len(sys.argv) > 1 else "S-mid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prob = synth.CONFIGS[cfg]()
model = P2Model(prob)
eng = LpEngine.from_model(model, pool_slots=B + 1)
import ctypes
eng.lib.bslv_lpq_rows_folded.argtypes = [ctypes.c_void_p]
folded = eng.lib.bslv_lpq_rows_folded(eng.h)
print(cfg, "LP", model.M - folded, "x", model.N, "(tableau; model as given %d x %d, %d rows folded into column bounds by the LP layer)" % (model.M, model.N, folded), flush=True)
print("slot bytes", eng.slot_bytes(), flush=True)
eng.set_profile(True)
rng = np.random.default_rng(0)
n, q = prob["n"], prob["q"]
v0 = np.zeros((1, q))
eng.reset_slot(0)
t = time.time()
st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), model.ub_for(v0))
ls0 = eng.last_stats()
print("cold start: status", st, "pivots", it, "passes %d" % ls0["passes"], "time %.3f s" % (time.time() - t), ls0, flush=True)
y0 = eng.primal([0], model.y_first, q)[0]
z0 = eng.obj([0])[0]
print("y0", y0, "z0", z0)
# vertices near the found boundary point: v = y0 - delta
for scale in (0.01, 0.1, 1.0):
    V = y0[None, :] - np.abs(rng.normal(scale=scale * np.abs(y0).mean(), size=(B, q)))
    src = np.zeros(B, np.int32); dst = np.arange(1, B + 1, dtype=np.int32)
    t = time.time()
    st, it = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), model.ub_for(V))
    dt = time.time() - t
    s = eng.last_stats()
    piv = s["pivots"]
    bytes_pass = s["passes"] * 16.0 * (model.M - folded + 1) * (model.N + 1)         # one read + one write of the tableau per (LP, pass)
    print("scale %.2f: B=%d ok=%d pivots/LP mean %.1f max %d rounds %d passes %d pivots %d total %.1f ms flush %.1f ms -> %.0f LPs/s, k_flush %.0f GB/s (tableau passes), %.0f GB/s per-pivot equivalent"
          % (scale, B, int((st == 4).sum()), it.mean(), it.max(), s["lockstep_iters"], s["passes"], piv, dt * 1e3, s["update_ms"],
             B / dt, bytes_pass / (s["update_ms"] * 1e-3) / 1e9, bytes_pass / max(s["passes"], 1) * piv / (s["update_ms"] * 1e-3) / 1e9), flush=True)
