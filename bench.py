#!/usr/bin/env python3
"""bench.py -- scalar LPs/s (+ vertices enumerated/s) of the batched Benson hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched under
torch.distributed.run, one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.

  step      = one outer iteration of the re-shaped phase2_primal loop: collect the newest B*N
              unprocessed vertices -> B LPs per GPU (batched dual simplex on HBM-resident tableaux)
              -> one all_gather of the cut records -> every rank applies the cuts (GPU double
              description).  Weak scaling: B LPs per GPU per step.
  workload  = S-mid: synthetic VLP q=5, n=500, m=1000 (BASELINE.json configs[2]/[3]).
  setup (untimed): problem generation, upload, cold start + the r weighted-sum LPs, and the ramp
              of the vertex queue up to one full batch -- so every timed step has full batches and
              all inputs (constraint matrix, tableau pool, polyhedron) are resident in HBM.
  roofline  = the tableau pass (k_flush): algorithmic bytes = (LP, pass) pairs * 16*(m+r+1)*(n+2)
              (one read + one write of the eliminated tableau per pass; SURVEY 8d K3 charges that per pivot,
              the engine applies up to 6 pivots per pass) / its HIP-event time, measured live inside the
              timed region on the engine's stream.
  cpu_baseline = the CPU oracle (oracle/benson_cpu.c: sequential loop, warm-started dense dual
              simplex, 1 core) on the first LPs of the same workload, rank 0, N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def bench_ex09_lp(args):
    """--workload ex09-lp: the scalar LP of ex09 of the reference's suite (ex/ex09.vlp: 4608 x 36 939, 185 856 non-zeros, q = 3, ordering
    cone with 6 generators -- SURVEY 8f rank 4, the problem the reference hands to GLPK as COO, bslv_lp.c:60-70) on the REVISED form
    of the LP engine: basis inverse per LP (171 MB) instead of the tableau (1.36 GB), A once as CSC / CSR.  One step = a batch of
    B P2(v) solves warm-started from one solved LP (B different right-hand sides); value = LPs / s over the K timed steps, the roofline
    is the tableau-pass kernel on B^-1 (k_flush: 16 M ldt bytes per (LP, pass)).  One GPU."""
    import numpy as np
    import torch
    from bensolve_amd.synth import read_vlp
    from bensolve_amd.lp import P2Model, LpEngine
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(0)
    steps = args.steps or 1
    warm = 0 if args.warmup is None else args.warmup
    B = args.batch or 8
    prob = read_vlp(os.path.join(ROOT, "tests", "golden", "ex", "ex09.vlp"))
    model = P2Model(prob)
    os.environ.setdefault("BSLV_LP_REV", "1")          # (this workload IS the revised form; BSLV_LP_REV=0 python bench.py --workload ex09-lp rates the tableau form on the same LPs)
    eng = LpEngine.from_model(model, pool_slots=B + 1)
    dims = dict(M=model.M, N=model.N)
    eng.reset_slot(0)
    v0 = np.full((1, prob["q"]), 1e3)
    t0 = time.perf_counter()
    st, it = eng.solve_batch([0], [0], np.full((1, model.r), -np.inf), model.ub_for(v0)[:1])
    cold = dict(status=int(st[0]), pivots=int(it[0]), secs=round(time.perf_counter() - t0, 2), **{k: eng.last_stats()[k] for k in ("passes", "lockstep_iters")})
    print("bench: ex09 cold LP done: %s" % cold, file=sys.stderr, flush=True)
    if st[0] != 4:
        raise SystemExit("ex09-lp: the cold solve failed: status %d" % st[0])
    rng = np.random.default_rng(9)
    src = np.zeros(B, np.int32)
    dst = np.arange(1, B + 1, dtype=np.int32)

    bad = [0]

    def one_step():
        # every coordinate scaled by itself: a few thousand pivots away from the LP the batch starts from (points on the ray through v0 keep
        # its basis: 0 pivots).  About one such LP in ten runs into a primal clean-up that stalls for 4 (M + N) = 166 000 pivots until
        # Bland's rule ends it: the step then rates the stall, which is what a caller of this form meets today
        V = v0 * rng.uniform(0.6, 0.98, size=(B, prob["q"]))
        stv, itv = eng.solve_batch(src, dst, np.full((B, model.r), -np.inf), model.ub_for(V))
        bad[0] += int((stv != 4).sum())        # (revised form: an LP whose inverse drifted comes back UNDEFINED for the caller's retry)
        return int(itv.sum()), eng.last_stats()
    for _ in range(warm):
        one_step()
    eng.set_profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    piv = passes = rounds = 0
    upd_ms = 0.0
    for _ in range(steps):
        p, ls = one_step()
        piv += p; passes += ls["passes"]; rounds += ls["lockstep_iters"]; upd_ms += ls["update_ms"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.set_profile(False)
    rev = int(eng.lib.bslv_lpq_is_revised(eng.h))
    eng.close()
    ldt = (model.M + 15) // 16 * 16
    alg = 16.0 * model.M * ldt if rev else 16.0 * (model.M + 1) * ((model.N + 15) // 16 * 16)
    achieved = passes * alg / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    out = {"metric": "scalar LPs/sec (P2(v) of ex/ex09.vlp: %d x %d, 185856 non-zeros, q=3), batches of %d warm-started LPs" % (model.M, model.N, B),
           "value": round(steps * B / dt, 2), "unit": "LPs/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": round(dt * 1e3 / steps, 2), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "ex/ex09.vlp of the reference's suite (tests/golden/ex), right-hand sides synthetic",
           "config": {"workload": "ex09-lp", "lp_rows_cols": [dims["M"], dims["N"]], "form": "revised (basis inverse per LP, A as CSC/CSR)" if rev else "tableau", "batch": B,
                      "slot_bytes": 8 * model.M * ldt if rev else 8 * (model.M + 1) * model.N},
           "pivots_per_lp": round(piv / max(steps * B, 1), 1), "passes": passes, "lockstep_rounds": rounds, "cold_start": cold, "lps_not_optimal": bad[0],
           "roofline": {"bound": "hbm", "kernel": "k_flush on B^-1 (pending pivots applied in one pass)", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                        "traffic": None, "launches": rounds, "avg_launch_us": round(upd_ms * 1e3 / max(rounds, 1), 1), "alg_bytes_per_pass": alg,
                        "note": "the pass is NOT what bounds this workload: the selection (k_select, one workgroup per LP: sparse row / column products and passes over 37 000 columns) is -- profiles/r04_ex09_lp_kernel_stats.csv"},
           "cpu_baseline": {"value": None, "unit": "LPs/s", "cores": 1, "kind": "port", "sample": "none: oracle/lp_dense.c needs a dense 4618 x 36943 tableau per LP and minutes per cold solve -- not a bounded sample"}}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed outer iterations (default 20; S-degenerate: 3, its polyhedron outgrows 32-bit indices after ~6)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed outer iterations before them (default 4; S-degenerate: 0)")
    ap.add_argument("--workload", default="S-mid")
    ap.add_argument("--batch", type=int, default=0, help="LPs per GPU per step (default by workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="N=1 only: overlap the LPs of batch k with the cuts of batch k-1 (measured slower on S-mid: the one-batch lag costs more cuts per LP than the overlap saves)")
    ap.add_argument("--policy", type=int, default=0, help="batch selection: 1 newest first, 2 spread, 3 newest first with at most --sib-cap children of one cut (default by workload)")
    ap.add_argument("--sib-cap", type=int, default=1)
    ap.add_argument("--sib-window", type=int, default=8)
    ap.add_argument("--fronts", type=int, default=8, help="policy 4: number of depth-first fronts")
    ap.add_argument("--fam-mode", type=int, default=3, help="policy 6: parents newest first (0), pseudo-random (1), far apart (2)")
    ap.add_argument("--fam-batches", type=int, default=1, help="policy 6: parents among the cuts of the last N outer iterations")
    ap.add_argument("--pool", type=int, default=0, help="tableau slots (default 4*batch+64)")
    ap.add_argument("--no-long-window", action="store_true", help="skip the continuation of the run to 5x the steps (reported as long_window, not part of value)")
    ap.add_argument("--no-pair", action="store_true", help="skip the S-small whole-run GPU/CPU pair of cpu_baseline")
    ap.add_argument("--cpu-lps", type=int, default=0, help="LPs of the CPU sample (default by workload)")
    args = ap.parse_args()
    if args.workload == "ex09-lp":
        return bench_ex09_lp(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from bensolve_amd import synth
    from bensolve_amd.benson import BensonEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # REHEARSAL of the N > 1 path on a box with fewer GPUs than ranks (BSLV_BENCH_REHEARSAL=1): the ranks share the devices there
    # are, torch.distributed runs on gloo and the exchange step on the callback transport -- RCCL itself refuses two ranks on one
    # device.  Everything else (dealing the batch, the C step, barrier and max over ranks, the JSON line) is the N-GPU code.
    rehearsal = bool(os.environ.get("BSLV_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    from bensolve_amd._lib import load_library as _load_lib
    _rc = _load_lib().bslv_set_device(local_rank)      # the engine allocates on the thread's current HIP device: make it explicit
    if _rc != 0:
        raise SystemExit("bslv_set_device(%d) failed: %d" % (local_rank, _rc))
    device = torch.device("cuda", local_rank)
    transport = "none (one rank)"
    if world > 1 and rehearsal:
        dist.init_process_group("gloo")
        from bensolve_amd.benson import dist_init_callback
        dist_init_callback(dist)
        transport = "REHEARSAL: gloo all_gather through bslv_dist_init_callback, ranks share a device"
    elif world > 1:
        dist.init_process_group("nccl", device_id=device)
        # the data-path collective is the library's own: direct ncclAllGather (RCCL over xGMI) inside bslv_benson_step_dist;
        # torch.distributed hands out the communicator id and does the barrier / max-over-ranks of the timing
        from bensolve_amd.benson import dist_init_rccl
        transport = dist_init_rccl(dist, device)
        if not transport.startswith("ncclAllGather"):
            # LOUD: the data-path collective is meant to be RCCL inside the library; anything else is a different measurement
            print("bench.py: rank %d: RCCL transport NOT in use: %s" % (rank, transport), file=sys.stderr)
            if not os.environ.get("BSLV_BENCH_ALLOW_FALLBACK"):
                raise SystemExit("bench.py --gpus %d: the library's RCCL communicator could not be created on every rank (%s); "
                                 "set BSLV_BENCH_ALLOW_FALLBACK=1 to measure the torch.distributed fallback instead" % (world, transport))

    if args.steps is None:
        args.steps = 20     # (the rate moves with the window: waves of redundant LPs -- 5-step windows gave 62 k .. 108 k LPs/s on S-mid; 20 steps average over them.  S-degenerate: as far as its polyhedron lets it, see `stopped`)
    if args.warmup is None:
        args.warmup = 0 if args.workload == "S-degenerate" else 4
    defaults = {"S-small": 2048, "S-mid": 2048, "S-degenerate": 64}
    B = args.batch or defaults.get(args.workload, 256)
    # CPU sample: (warm-up, rated) LPs in the reference's vertex order, then newest first
    cpu_plan = {"S-small": (1000, 8000, 1000, 8000), "S-mid": (100, 200, 200, 1500), "S-degenerate": (2, 2, 2, 2)}.get(args.workload, (50, 200, 50, 200))
    if args.cpu_lps:
        cpu_plan = (cpu_plan[0], args.cpu_lps, cpu_plan[2], args.cpu_lps)
    prob = synth.CONFIGS[args.workload]()
    m, n, q = prob["m"], prob["n"], prob["q"]
    r = q
    pool_slots = args.pool or 4 * B + 64
    eng = BensonEngine(prob, eps=1e-7, pool_slots=pool_slots)
    slot_bytes = eng.lp_call("slot_bytes")
    if args.policy:
        eng.set_policy(args.policy)
        if args.policy == 3:
            eng.set_sibling_rule(args.sib_cap, args.sib_window)
        if args.policy == 4:
            eng.set_fronts(args.fronts, args.sib_cap)
        if args.policy == 6:
            eng.set_families(args.fam_mode, args.fam_batches)
    st = eng.start()
    if st != 0:
        raise SystemExit("phase 2 start failed: vlp status %d" % st)

    pipe = None
    phase_ms = [0.0, 0.0, 0.0]          # host wall clock of collect / LP batch / cut application (N=1, unpipelined)
    dist_ms = [0.0, 0.0, 0.0, 0.0]      # N > 1, this rank: collect / its LPs / all-gather (waits for the slowest rank) / application of ALL ranks' cuts

    def one_step():
        if world > 1:
            s = eng.step(B * world)              # (bslv_benson_step_dist: collect -> this rank's LPs -> all-gather -> apply)
            s.update(n_total=s["lps"])
            ph4 = (ctypes.c_double * 4)()
            eng.lib.bslv_dist_last_phases(ph4)
            for k in range(4):
                dist_ms[k] += ph4[k]
            return s
        if pipe is not None:
            s = pipe.step()
            s["lps"] = s["lps_solved"]
            return s
        t0 = time.perf_counter()
        nl, nt = eng.collect(B, 0, 1)
        t1 = time.perf_counter()
        rec, piv, ls = eng.solve_local(nl)
        t2 = time.perf_counter()
        s = eng.apply(rec)
        t3 = time.perf_counter()
        phase_ms[0] += (t1 - t0) * 1e3; phase_ms[1] += (t2 - t1) * 1e3; phase_ms[2] += (t3 - t2) * 1e3
        s.update(n_local=nl, n_total=nt, pivots=piv, lockstep=ls)
        return s

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ramp (untimed setup): grow the vertex queue to one full global batch
    ramp_steps = 0
    while True:
        s = one_step()
        ramp_steps += 1
        left = eng.lib.bslv_benson_unprocessed_left(eng.h)
        cnt = eng.poly_call("unprocessed", 0)[3]
        if cnt >= B * world or ramp_steps > 200 or (s["n_total"] == 0):
            break
    if world == 1 and args.pipeline:
        from bensolve_amd.benson import PipelinedStepper
        pipe = PipelinedStepper(eng, B)
        one_step()                       # fills the pipeline (untimed)
    for _ in range(args.warmup):
        one_step()

    eng.lp_call("set_profile", True)
    phase_ms[:] = [0.0, 0.0, 0.0]
    dist_ms[:] = [0.0, 0.0, 0.0, 0.0]
    rounds0 = eng.poly_call("rounds_run")
    ps0 = eng.poly_call("path_stats")
    r2s0 = eng.poly_call("rounds2_stats")
    nv0 = eng.poly_call("counts")["new_vertices"]
    pt0 = eng.poly_call("counts")["pair_tests"]
    def lazy_stats():
        o = (ctypes.c_long * 3)()
        eng.lib.bslv_lpq_lazy_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        eng.lib.bslv_lpq_lazy_stats(eng._lp_h, o)
        return [int(o[0]), int(o[1]), o[2] / 1000.0]
    lz0 = lazy_stats()
    lps = cuts = pivots = lockstep = redundant = confirmed = passes = flush_launches = 0
    upd_ms = 0.0
    lp_ms = 0.0
    sync()
    step_t = []                          # host clock at the end of every timed step (for the spread of the rate over windows)
    step_lps = []
    t0 = time.perf_counter()
    stopped = None                       # S-degenerate at q = 10 outgrows the engine's 32-bit block / edge indices after a number of steps: the line then rates the steps that completed
    steps_asked = args.steps
    for k_step in range(args.steps):
        try:
            s = one_step()
        except Exception as e:
            if args.workload != "S-degenerate" or world > 1 or k_step == 0:
                raise
            stopped = {"after_steps": k_step, "asked": steps_asked, "reason": str(e)[-300:]}
            print("bench: stopped after %d of %d steps: %s" % (k_step, steps_asked, e), file=sys.stderr, flush=True)
            args.steps = k_step
            break
        if args.workload == "S-degenerate" and rank == 0:
            print("bench: step %d: %d LPs, %d cuts, %.1f s so far" % (k_step + 1, s["lps"], s["cuts"], time.perf_counter() - t0), file=sys.stderr, flush=True)
        step_t.append(time.perf_counter()); step_lps.append(s["lps"])
        lps += s["lps"]
        cuts += s["cuts"]
        redundant += s.get("redundant", 0)
        confirmed += s.get("confirmed", 0)
        pivots += s["pivots"]
        lockstep += s["lockstep"]
        ls = eng.lp_call("last_stats")
        upd_ms += ls["update_ms"]
        lp_ms += ls["total_ms"]
        passes += ls["passes"]
        flush_launches += ls["launches"]
    sync()
    dt = time.perf_counter() - t0
    eng.lp_call("set_profile", False)
    # the counters of the timed region, before anything else runs on this engine
    snap = {"counts": eng.poly_call("counts"), "rounds_run": eng.poly_call("rounds_run"), "path_stats": eng.poly_call("path_stats"),
            "health": eng.poly_call("rounds2_health"), "starts": eng.start_stats(), "totals": eng.totals(), "r2": eng.poly_call("rounds2_stats"), "defer": eng.defer_stats()}
    snap["live"] = int(eng.poly_dump()["pu"].sum()) if snap["counts"]["nprimal"] < 5_000_000 and stopped is None else -1
    eng.lib.bslv_poly_largest_facet.argtypes = [ctypes.c_void_p]
    snap["largest_facet"] = int(eng.lib.bslv_poly_largest_facet(eng._poly_h))
    if rank == 0: print("bench: timed region done (%d steps, %.3f s)" % (args.steps, dt), file=sys.stderr, flush=True)
    phase_timed = list(phase_ms)         # (a copy: one_step() keeps adding to phase_ms if the run goes on below)
    lz1 = lazy_stats()
    mat_ms = lz1[2] - lz0[2]             # tableau passes made on request at the end of apply() (lazy tableaux): LP work, booked under "lp_tableaux_on_request", not under the cuts
    dist_timed = list(dist_ms)
    # Everything the headline needs is measured.  Two more figures for the reader, outside the timed region (rank 0, one GPU):
    #  * long_window: the SAME run continued to 5x the steps -- S-mid never terminates, the polyhedron keeps growing, and the rate
    #    of the first steps after the ramp is not the rate of a long run;
    #  * below, after the CPU baseline: the batch selection rule of rounds 1-2 (newest vertices first) on a fresh engine.
    long_window = None
    if world == 1 and pipe is None and args.workload == "S-mid" and not args.no_long_window:
        try:
            extra = 4 * args.steps
            l2 = c2 = f2 = 0
            tl = time.perf_counter()
            for _ in range(extra):
                s = one_step()
                l2 += s["lps"]; c2 += s["cuts"]; f2 += s.get("confirmed", 0)
            sync()
            dtl = time.perf_counter() - tl
            print("bench: long window done (%d more steps, %.3f s)" % (extra, dtl), file=sys.stderr, flush=True)
            long_window = {"steps": args.steps + extra, "lps_per_sec": round((lps + l2) / (dt + dtl), 1), "useful_lps_per_sec": round((cuts + confirmed + c2 + f2) / (dt + dtl), 1),
                           "continuation_only": {"lps_per_sec": round(l2 / dtl, 1), "useful_lps_per_sec": round((c2 + f2) / dtl, 1)},
                           "note": "the timed region plus 4x as many steps of the same run (not part of `value`)"}
        except Exception as e:      # (a figure beside the headline must not cost the headline)
            long_window = {"error": str(e)}
            print("bench: long window failed: %s" % e, file=sys.stderr, flush=True)

    c1 = snap["counts"]
    new_vertices = c1["new_vertices"] - nv0
    pair_tests = c1["pair_tests"] - pt0
    live = snap["live"]

    # spread of the headline: the rate over consecutive windows of the timed region (>= 3 windows; rank 0's clock)
    nwin = min(5, args.steps) if args.steps >= 3 else 1
    win_rates = []
    for wdx in range(nwin):
        a, b = wdx * args.steps // nwin, (wdx + 1) * args.steps // nwin
        if b > a:
            ta = t0 if a == 0 else step_t[a - 1]
            win_rates.append(sum(step_lps[a:b]) / max(step_t[b - 1] - ta, 1e-9))
    win_rates.sort()
    starts = snap["starts"]
    # max over ranks of the elapsed time; sums of the per-rank pivot counts
    if world > 1:
        cdev = torch.device("cpu") if rehearsal else device       # (gloo: host tensors)
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        agg = torch.tensor([pivots, upd_ms, snap["totals"]["lps"], starts["root"], starts["nearest"]] + dist_timed + [lp_ms], dtype=torch.float64, device=cdev)
        allp = [torch.zeros_like(agg) for _ in range(world)]
        dist.all_gather(allp, agg)
        pivots_all = sum(float(a[0]) for a in allp)
        per_rank = [{"rank": r, "pivots": int(a[0]), "starts_from_root_tableau": int(a[3]), "starts_from_nearest_tableau": int(a[4]),
                     "phase_ms_per_step": {"collect": round(float(a[5]) / args.steps, 3), "lp": round(float(a[6]) / args.steps, 3), "allgather_incl_wait": round(float(a[7]) / args.steps, 3),
                                           "cuts_of_all_ranks": round(float(a[8]) / args.steps, 3)}} for r, a in enumerate(allp)]
    else:
        pivots_all = pivots
        per_rank = [{"rank": 0, "pivots": int(pivots), "starts_from_root_tableau": starts["root"], "starts_from_nearest_tableau": starts["nearest"]}]

    # Dominant kernel: k_flush, one pass over the tableau of every LP that has pivots pending (delayed update: up to 6 pivots
    # are selected on vectors, then applied together).  Unit = one (LP, pass); algorithmic bytes per unit = one read + one
    # write of the eliminated tableau, 16*(m+r+1)*(n+2) -- SURVEY 8d K3's figure, which the reference algorithm pays per PIVOT.
    dims = eng.lp_dims()                 # (rows of A that were single-variable bounds are not in the LP)
    m_lp = dims["M"] - dims["rows_folded"] - q - r - 1     # rows of A in the TABLEAU (the LP layer folds rows with one non-zero into column bounds)
    alg_bytes_per_pass = 16.0 * (m_lp + r + 1) * (n + 2)
    launches = max(flush_launches, 1)    # (every k_flush launch the HIP events bracketed: the lock-step rounds AND the passes made on request for the kept LPs -- what a kernel trace of the timed region counts)
    achieved = (passes * alg_bytes_per_pass) / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    per_pivot_equiv = (pivots * alg_bytes_per_pass) / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    # HBM traffic from the PMC counters: collected in separate rocprofv3 --pmc passes (profiles/r04_pmc_k_flush.json,
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), per tableau pass; scaled to this run's passes per launch
    traffic = None
    pmc_file = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_pmc_k_flush.json" % k) for k in (4, 3, 1)) if os.path.exists(p)), "")
    if args.workload == "S-mid" and os.path.exists(pmc_file):
        per_pass = json.load(open(pmc_file))["k_flush"]["traffic_bytes_per_pass"]
        traffic = round(per_pass * passes / launches, 0)
    roofline = {"bound": "hbm", "kernel": "k_flush (tableau pass applying the pending pivots)", "achieved": round(achieved, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "launches": launches, "avg_launch_us": round(upd_ms * 1e3 / launches, 2),
                "alg_bytes_per_launch": round(passes * alg_bytes_per_pass / launches, 0),
                "tableau_passes_rank0": passes, "pivots_rank0": pivots, "pivots_per_pass": round(pivots / max(passes, 1), 2),
                "per_pivot_equivalent_GBps": round(per_pivot_equiv, 1),
                "note": "per_pivot_equivalent = what one read+write of the tableau PER PIVOT (SURVEY 8d K3, the reference's GLPK-style update) would need to move in the same time"}

    # second figure of merit: the cut phase (bslv_poly's half of the path).  Not HBM-bound (integer / LDS /
    # latency): reported as time per cut, cuts per pass over the polyhedron, and pair tests per second next to the
    # reference's own bslv_poly.c on one core (6.9e6/s at q=5, N=1000).
    cut_ms = (phase_timed[2] if world == 1 else dist_timed[3]) - mat_ms       # (N > 1: every rank applies the cuts of ALL ranks: rank 0's clock; without the tableau passes apply() makes for the kept LPs)
    passes_poly = snap["rounds_run"] - rounds0
    ps = snap["path_stats"]
    roofline_cuts = {"bound": "latency/integer (not hbm)", "kernels": "per round of independent cuts: k_r2_minit (conflict matrix, LDS) + k_r2_select3 (maximal independent set) + k_r2_assign3 + k_flags2 + k_r2_emit + k_r2_classify3 + k2_fused_t<true> (one prune per selected cut) + k_r2_k2emit; k_flags2+k_emit2+k2_fused per single cut",
                     "cuts_applied": cuts, "cuts_per_step": round(cuts / max(args.steps, 1), 1),
                     "us_per_cut": round(cut_ms * 1e3 / max(cuts, 1), 2) if pipe is None else None,
                     "lps_per_cut": round(lps / max(cuts, 1), 3),
                     "scaling_ceiling_lps_per_sec": round((lps / max(cuts, 1)) / (cut_ms * 1e-3 / max(cuts, 1)), 1) if pipe is None and cut_ms > 0 else None,
                     "scaling_ceiling_note": "the cut phase is replicated: every rank applies the cuts of ALL ranks, so the whole-job rate at ANY number of GPUs stays below LPs per cut / seconds per cut of one replica (this figure) -- the LP phase is the only part that shards",
                     "passes_over_polyhedron": passes_poly, "cuts_per_pass": round(cuts / max(passes_poly, 1), 2),
                     "single_cut_pipeline_cuts": ps["single_cuts"] - ps0["single_cuts"], "hot_chunks": ps["hot_chunks"] - ps0["hot_chunks"],
                     "pair_tests_per_sec": round(pair_tests / dt, 1), "reference_pair_tests_per_sec_1core": 6.9e6,
                     "mailbox": snap["health"],
                     "multi_kernel_prunes": (snap["r2"]["fallback_prunes"] - r2s0["fallback_prunes"]) + (ps["prune_fallbacks"] - ps0["prune_fallbacks"]),
                     "largest_new_facet_elements": snap["largest_facet"],
                     "cuts_handed_back": snap["defer"]["handed_back"], "cuts_waiting_at_end": snap["defer"]["waiting"], "one_family_batches": snap["defer"]["one_family_batches"],
                     "note": "a round applies a maximal set of mutually independent cuts of the chunk (<= 1024 cuts) in the passes of one cut: ~250 us of dependent small launches, latency-bound (profiles/r03_cut_phase_counters.json: 64-92 % of wave cycles waiting)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload.startswith("S-degenerate") and not args.cpu_lps:
        # the oracle's plain dual simplex does not finish ONE cold LP of this family at full size in minutes (45 k pivots at
        # q=6, n=500): no CPU figure instead of a bench that never returns.  --cpu-lps N forces the attempt.
        cpu = {"value": None, "unit": "LPs/s", "cores": 1, "kind": "port",
               "sample": "not measured: oracle/lp_dense.c needs > 10 minutes for the cold start of one 4000 x 2000 LP of the degenerate family (no bound flipping, no perturbation); --cpu-lps N forces it"}
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        # same stage on both sides: the CPU oracle first works through `warm` LPs untimed-by-us (its cold solve and the first,
        # far-apart vertices), then the LPs behind the warm-up are rated -- (a) in the reference's own vertex order (lowest slot
        # first, poly__get_vrtx) and (b) newest first, the order the batched driver uses (consecutive vertices are neighbours:
        # few pivots per warm-started LP).  In the sequential loop every LP is useful (one cut or one confirmation).
        warm_a, n_a, warm_b, n_b = cpu_plan
        rc, fp, ca = oracle_api.benson_phase2_primal(prob, eps=1e-7, max_lps=warm_a + n_a, order=0, warm_lps=warm_a)
        fp.close()
        rc, fp, cb = oracle_api.benson_phase2_primal(prob, eps=1e-7, max_lps=warm_b + n_b, order=1, warm_lps=warm_b)
        fp.close()

        def rate(c):
            w = c.warm_lps if c.warm_lps else 0
            secs = c.secs_total - (c.warm_secs if w else 0.0)
            lp = c.lps - w
            return lp / max(secs, 1e-9), (c.pivots - (c.warm_pivots if w else 0)) / max(lp, 1), lp, secs, (c.new_vertices - (c.warm_new_vertices if w else 0)) / max(secs, 1e-9)
        ra, rb = rate(ca), rate(cb)
        cpu = {"value": round(ra[0], 3), "unit": "LPs/s", "cores": 1, "kind": "port",
               "sample": "oracle/benson_cpu.c (sequential phase2_primal loop: one vertex -> one warm-started dense dual simplex LP -> one cut; "
                         "GLPK is not installed, so the reference's own LP half cannot run), %s, reference vertex order (lowest slot first), "
                         "%d LPs after a warm-up of %d LPs, %.1f s" % (args.workload, ra[2], ca.warm_lps, ra[3]),
               "pivots_per_lp": round(ra[1], 1), "vertices_per_sec": round(ra[4], 1),
               "newest_first": {"value": round(rb[0], 3), "pivots_per_lp": round(rb[1], 1), "lps": rb[2], "warmup_lps": cb.warm_lps, "secs": round(rb[3], 2),
                                "vertices_per_sec": round(rb[4], 1),
                                "note": "same oracle, vertices taken newest first as the batched driver does: the best sequential order for warm starts"},
               "every_lp_useful": True}
        if args.workload == "S-mid" and not args.policy and not args.no_long_window:
            # the batch selection rule of rounds 1-2 on a fresh engine, same ramp / warm-up / steps: what BENCH_r01 / r02 measured
            print("bench: CPU baseline done; rounds-1-2 rule on a fresh engine ...", file=sys.stderr, flush=True)
            try:
                e3 = BensonEngine(prob, eps=1e-7, pool_slots=pool_slots)
                e3.set_policy(1)
                e3.poly_call("debug_set", 11, 0); e3.poly_call("debug_set", 7, 512)      # (rounds by local minima, chunks of 512: round 2's cut phase)
                e3.start()
                for _ in range(200):
                    e3.step(B)
                    if e3.poly_call("unprocessed", 0)[3] >= B:
                        break
                for _ in range(args.warmup):
                    e3.step(B)
                torch.cuda.synchronize(); t3 = time.perf_counter()
                l3 = c3 = 0
                for _ in range(args.steps):
                    s3 = e3.step(B)
                    l3 += s3["lps"]; c3 += s3["cuts"] + s3["confirmed"]
                torch.cuda.synchronize(); t3 = time.perf_counter() - t3
                e3.close()
                print("bench: rounds-1-2 rule done", file=sys.stderr, flush=True)
                cpu["policy_newest_first_rounds_of_local_minima"] = {"lps_per_sec": round(l3 / t3, 1), "useful_lps_per_sec": round(c3 / t3, 1), "lps_redundant_frac": round(1 - c3 / max(l3, 1), 4),
                                                                      "note": "rounds 1-2's selection (newest vertices first) and cut phase (local minima of one order, chunks of 512) on a fresh engine over the same window; not part of `value`"}
            except Exception as e:      # (a figure beside the headline must not cost the headline)
                cpu["policy_newest_first_rounds_of_local_minima"] = {"error": str(e)}
                print("bench: rounds-1-2 rule failed: %s" % e, file=sys.stderr, flush=True)
        if not args.no_pair and args.workload != "S-small":
            # like-for-like pair: S-small to termination on both sides (same problem, same eps, whole phase 2)
            sp = synth.CONFIGS["S-small"]()
            print("bench: S-small pair ...", file=sys.stderr, flush=True)
            try:
                e2 = BensonEngine(sp, eps=1e-7, pool_slots=4 * 2048 + 64)
                e2.start()
                torch.cuda.synchronize(); tg = time.perf_counter()
                e2.run(2048)
                torch.cuda.synchronize(); tg = time.perf_counter() - tg
                tot = e2.totals()
                e2.close()
                rc, fp, cs = oracle_api.benson_phase2_primal(sp, eps=1e-7)
                fp.close()
                cpu["whole_run_pair"] = {"workload": "S-small (q=3, n=100, m=200) phase 2 to termination", "gpu_secs": round(tg, 4), "gpu_lps": tot["lps"],
                                         "cpu_secs": round(cs.secs_total, 4), "cpu_lps": cs.lps, "speedup_time_to_termination": round(cs.secs_total / max(tg, 1e-9), 2)}
            except Exception as e:      # (a figure beside the headline must not cost the headline)
                cpu["whole_run_pair"] = {"error": str(e)}
                print("bench: S-small pair failed: %s" % e, file=sys.stderr, flush=True)

    if rank == 0:
        out = {
            "metric": "scalar LPs/sec (Benson phase 2, batched P2(v) solves incl. cut application), synthetic VLP q=%d n=%d m=%d" % (q, n, m),
            "value": round(lps / dt, 2), "unit": "LPs/s",
            "value_min": round(win_rates[0], 2) if win_rates else None, "value_median": round(win_rates[len(win_rates) // 2], 2) if win_rates else None,
            "value_max": round(win_rates[-1], 2) if win_rates else None, "value_windows": len(win_rates),
            "useful_lps_per_sec": round((cuts + confirmed) / dt, 2), "lps_redundant_frac": round(1.0 - (cuts + confirmed) / max(lps, 1), 4),
            "useful_note": "useful = LPs whose outcome changed the state (cut applied or vertex confirmed); the rest returned a cut that an earlier LP of the same batch had already delivered.  The reference's sequential loop solves only useful LPs (bslv_algs.c:1030-1080): compare useful_lps_per_sec with cpu_baseline.value", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s (q=%d, n=%d, m=%d dense covering VLP, seed per SURVEY 8d)" % (args.workload, q, n, m),
                       "lp_rows_cols": [dims["M"] - dims["rows_folded"], dims["N"]], "rows_folded_by_presolve": dims["rows_folded"], "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "vertex batch sharded over %d GPU(s), one all-gather of cut records per step (in the library), cut application replicated" % world,
                       "transport": transport,
                       "lp_poly_overlap": pipe is not None,
                       "tableau_slot_bytes": slot_bytes, "pool_slots": pool_slots, "batch_policy": args.policy or "6 (library default: whole families, children of the shallowest cuts of the last batch first)", "ramp_steps_untimed": ramp_steps},
            "vertices_per_sec": round(new_vertices / dt, 1), "new_vertices": new_vertices, "cuts_applied": cuts, "cuts_redundant": redundant, "vertices_confirmed": confirmed,
            "lazy_tableaux": {"lp_passes_skipped": lz1[0] - lz0[0], "passes_made_on_request": lz1[1] - lz0[1], "note": "an LP that is finished when its tableau pass would be due keeps its pending pivots; only the LPs whose cut is applied (the parents of later LPs) get their tableau, at the end of apply() -- bslv_lpq_set_lazy / bslv_lpq_materialise"},
            "lps": lps, "pivots_per_lp": round(pivots_all / max(lps, 1), 2), "pair_tests_per_sec": round(pair_tests / dt, 1),
            "live_vertices": live, "poly_rounds": snap["rounds_run"] - rounds0, "lp_ms_rank0": round(lp_ms, 2), "phase_ms_per_step": ({"collect": round(phase_timed[0] / args.steps, 2), "lp": round(phase_timed[1] / args.steps, 2), "lp_tableaux_on_request": round(mat_ms / args.steps, 2), "cuts": round((phase_timed[2] - mat_ms) / args.steps, 2)} if world == 1 else
                                   {"collect": round(dist_timed[0] / args.steps, 2), "lp": round(dist_timed[1] / args.steps, 2), "allgather_incl_wait": round(dist_timed[2] / args.steps, 2), "cuts": round(dist_timed[3] / args.steps, 2), "note": "rank 0; every rank in warm_starts.per_rank"}), "update_kernel_ms_rank0": round(upd_ms, 2),
            "warm_starts": {"per_rank": per_rank, "note": "LPs whose parent's tableau was not resident on the rank that solved them start from the nearest resident tableau (else from the root tableau): the hit rate of the dealing rule"},
            "useful_vs_cpu_baseline": round(((cuts + confirmed) / dt) / cpu["value"], 1) if cpu and cpu.get("value") else None,
            "long_window": long_window, "stopped": stopped,
            "roofline": roofline, "roofline_cuts": roofline_cuts, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    eng.close()
    if world > 1:
        from bensolve_amd.benson import dist_finalize
        dist_finalize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
