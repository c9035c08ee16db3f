"""Does the distributed step cost anything by itself?  S-mid, one rank: bslv_benson_step against bslv_benson_step_dist (RCCL with one rank)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine
from bensolve_amd._lib import load_library, check
lib = load_library()
prob = synth.CONFIGS["S-mid"]()
for use_dist in (False, True):
    if use_dist:
        buf = (ctypes.c_ubyte * 128)()
        check(lib.bslv_dist_unique_id(buf, 128)); check(lib.bslv_dist_init(0, 1, buf, 128))
    eng = BensonEngine(prob, eps=1e-7, pool_slots=4 * 2048 + 64)
    assert eng.start() == 0
    f = lib.bslv_benson_step_dist if use_dist else lib.bslv_benson_step
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    stats = (ctypes.c_long * 8)(); ms = (ctypes.c_double * 3)()
    for _ in range(8): check(f(eng.h, 2048, stats, ms))
    torch.cuda.synchronize(); t0 = time.perf_counter(); lps = 0
    for _ in range(8):
        check(f(eng.h, 2048, stats, ms)); lps += stats[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("dist" if use_dist else "plain", "ms/step %.2f" % (dt / 8 * 1e3), "LPs/s %.0f" % (lps / dt), flush=True)
    eng.close()
lib.bslv_dist_finalize()
