"""The exchange step of the multi-GPU path inside the library (dist.hip, include/bslv_hip.h section 4a): one process per GPU,
the batch dealt to the ranks, ONE all-gather of fixed-size record blocks, every rank applies all records.
  * RCCL transport with a single rank on the one GPU of the test box (ncclCommInitRank + ncclAllGather are exercised;
    RCCL refuses two ranks on one device, so the two-rank runs below use the callback transport);
  * two ranks (two processes sharing the GPU) over the callback transport (torch.distributed / gloo): replicas bit-identical,
    result equal to the single-process run;
The command-line driver joins the communicator the same way (TCP hand-out of the id to MASTER_ADDR); with RCCL only, so it
cannot be run with two ranks on this box."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch                      # (before the library loads RCCL: PyTorch brings its own copy, and one process must not hold two)

import poly_harness as ph
from bensolve_amd import synth
from bensolve_amd.benson import BensonEngine, dist_init_callback, dist_finalize
from bensolve_amd._lib import load_library, check

pytestmark = pytest.mark.gpu


def test_rccl_transport_single_rank():
    lib = load_library()
    buf = (ctypes.c_ubyte * 128)()
    check(lib.bslv_dist_unique_id(buf, 128))
    assert any(buf)
    check(lib.bslv_dist_init(0, 1, buf, 128))
    try:
        assert lib.bslv_dist_world() == 1 and lib.bslv_dist_rank() == 0
        send = np.arange(1000, dtype=np.float64)
        recv = np.zeros(1000)
        check(lib.bslv_dist_allgather(send.ctypes.data_as(ctypes.c_void_p), recv.ctypes.data_as(ctypes.c_void_p), 1000))
        assert np.array_equal(send, recv)                      # through device memory and ncclAllGather
        # the distributed step with one rank equals the plain step
        prob = synth.covering_vlp(30, 15, 3, 5)
        res = []
        for use_dist in (True, False):
            eng = BensonEngine(prob, eps=1e-9, pool_slots=512)
            assert eng.start() == 0
            stats = (ctypes.c_long * 8)(); ms = (ctypes.c_double * 3)()
            for _ in range(10000):
                f = lib.bslv_benson_step_dist if use_dist else lib.bslv_benson_step
                f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
                check(f(eng.h, 32, stats, ms))
                if stats[0] == 0 and stats[7] == 0:
                    break
            d = eng.poly_dump()
            res.append(d)
            eng.close()
        for k in ("pu", "pi", "E", "I", "X", "Y", "du"):
            assert np.array_equal(res[0][k], res[1][k]), k
    finally:
        dist_finalize()


def _problem(args):
    return synth.degenerate_vlp(*args[1:5]) if args[0] == "degenerate" else synth.covering_vlp(*args)


def _force_large_facet_path(eng, shard):
    """every adjacency prune through the multi-kernel path with the row-tiled pair kernel (as for facets of 10^4..10^5 elements),
    and with `shard` its pair space dealt to the ranks"""
    eng.poly_call("debug_set", 0, 64)        # k2_fused's LDS too small for any facet: multi-kernel prune
    eng.poly_call("debug_set", 4, 2)         # every facet 'large': tiled pair kernel, member lists
    eng.poly_call("debug_set", 10, 2 if shard else 1 << 30)


def _rank_worker(rank, world, port, args, batch, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dist_init_callback(dist)
    prob = _problem(args)
    eng = BensonEngine(prob, eps=1e-9, pool_slots=512)
    if args[0] == "degenerate":
        _force_large_facet_path(eng, len(args) < 6 or args[5])
    assert eng.start() == 0
    eng.run(batch)                      # bslv_benson_step routes to bslv_benson_step_dist once a transport is set
    eng.poly_call("dual_adjacency")
    d = eng.poly_dump()
    tot = eng.totals()
    out[rank] = dict(d, lps=tot["lps"], local_pivots=tot["pivots"], sharded=eng.poly_call("sharded_prunes"), fallbacks=eng.poly_call("path_stats")["prune_fallbacks"])
    eng.close()
    dist_finalize()
    dist.destroy_process_group()


@pytest.mark.parametrize("args,batch", [((30, 15, 3, 5), 32), ((40, 20, 4, 9), 128), (("degenerate", 60, 30, 5, 3), 128)])
def test_two_ranks_through_the_c_step(args, batch):
    """(the third case: a member of the degenerate family with every adjacency prune forced through the large-facet path and its
    PAIR SPACE dealt to the two ranks -- SURVEY 8e -- : the adjacent pairs are all-gathered and appended in rank order, so the
    edge list, and with it everything else, is bit-identical to the run in which every rank tests all pairs)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_worker, args=(2, port, args, batch, out), nprocs=2, join=True)
    d0, d1 = out[0], out[1]
    for k in ("pu", "pi", "ps", "E", "I", "X", "Y", "du", "DE"):
        assert np.array_equal(d0[k], d1[k]), "replicas diverged in " + k
    assert d0["lps"] == d1["lps"] and d0["local_pivots"] > 0 and d1["local_pivots"] > 0        # both ranks solved LPs
    prob = _problem(args)
    eng = BensonEngine(prob, eps=1e-9, pool_slots=512)
    if args[0] == "degenerate":
        _force_large_facet_path(eng, False)
    assert eng.start() == 0
    eng.run(batch)
    eng.poly_call("dual_adjacency")
    sd = eng.poly_dump()
    single = ph.canonical(sd, decimals=6)
    eng.close()
    if args[0] == "degenerate":
        assert d0["sharded"] > 50 and d0["sharded"] == d1["sharded"] >= d0["fallbacks"], (d0["sharded"], d0["fallbacks"])     # (+ the prunes of the rounds)
        # the same two-rank run with every rank testing the whole pair space: bit-identical polyhedron, slot by slot and edge by edge
        out2 = mgr.dict()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_rank_worker, args=(2, port, tuple(args) + (False,), batch, out2), nprocs=2, join=True)
        assert out2[0]["sharded"] == 0
        for k in ("pu", "pi", "ps", "E", "I", "X", "Y", "du", "DE"):
            assert np.array_equal(d0[k], out2[0][k]), "sharded prune differs from the replicated prune in " + k


def _steps_worker(rank, world, port, args, batch, steps, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dist_init_callback(dist)
    eng = BensonEngine(synth.covering_vlp(*args), eps=1e-7, pool_slots=4 * batch + 64)
    assert eng.start() == 0
    lps = piv = cuts = 0
    r0 = None
    for k in range(steps):
        if k == steps // 2:                       # (the first half is the ramp: few vertices, cold tableaux)
            r0 = (eng.poly_call("rounds_run"), cuts, lps, piv)
        s = eng.step(batch)
        lps += s["lps"]; piv += s["pivots"]; cuts += s["cuts"]
    out[rank] = dict(lps=lps - r0[2], local_pivots=piv - r0[3], cuts=cuts - r0[1], passes=eng.poly_call("rounds_run") - r0[0], starts=eng.start_stats())
    eng.close()
    dist_finalize()
    dist.destroy_process_group()


def test_two_ranks_keep_their_warm_starts():
    """a two-rank rehearsal showed 73 pivots per LP against 5 with one rank -- vertices dealt to the rank that does not
    hold their parent's tableau started from the root tableau.  They now start from the nearest resident tableau of the rank that
    gets them, and the dealing rule keeps families with their owner: two ranks must not need more than twice the pivots per LP of
    one rank on the same global batch, nor fewer cuts per pass (the cut phase is replicated)."""
    import torch.multiprocessing as mp
    args, batch, steps = (300, 150, 4, 11), 512, 24
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_steps_worker, args=(2, port, args, batch, steps, out), nprocs=2, join=True)
    two_lps = out[0]["lps"]
    two_piv = out[0]["local_pivots"] + out[1]["local_pivots"]
    assert out[0]["lps"] == out[1]["lps"] and out[0]["cuts"] == out[1]["cuts"] and out[0]["passes"] == out[1]["passes"]
    eng = BensonEngine(synth.covering_vlp(*args), eps=1e-7, pool_slots=4 * batch + 64)
    assert eng.start() == 0
    lps = piv = cuts = 0
    for k in range(steps):
        if k == steps // 2:
            r0 = (eng.poly_call("rounds_run"), cuts, lps, piv)
        st = eng.step(batch)
        lps += st["lps"]; piv += st["pivots"]; cuts += st["cuts"]
    one = dict(lps=lps - r0[2], piv=piv - r0[3], cuts=cuts - r0[1], passes=eng.poly_call("rounds_run") - r0[0])
    eng.close()
    assert one["lps"] > 2000 and two_lps > 2000
    ppl1, ppl2 = one["piv"] / one["lps"], two_piv / two_lps
    cpp1, cpp2 = one["cuts"] / max(one["passes"], 1), out[0]["cuts"] / max(out[0]["passes"], 1)
    print("pivots per LP: one rank %.2f, two ranks %.2f | cuts per pass %.2f / %.2f | starts of the two ranks: %s %s" % (ppl1, ppl2, cpp1, cpp2, out[0]["starts"], out[1]["starts"]))
    assert ppl2 <= 2.0 * ppl1 + 1.0, (ppl1, ppl2)
    assert cpp2 >= 0.8 * cpp1, (cpp1, cpp2)


def _smid_worker(rank, world, port, batch, steps, out):
    import hashlib
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dist_init_callback(dist)
    eng = BensonEngine(synth.CONFIGS["S-mid"](), eps=1e-7, pool_slots=2 * batch + 64)
    assert eng.start() == 0
    lps = cuts = piv = 0
    for _ in range(steps):
        s = eng.step(batch * world)
        lps += s["lps"]; cuts += s["cuts"]; piv += s["pivots"]
    d = eng.poly_dump()
    h = hashlib.sha256()
    for k in ("pu", "pi", "ps", "E", "I", "X", "du", "Y"):
        h.update(k.encode()); h.update(np.ascontiguousarray(d[k]).tobytes())
    ph4 = (ctypes.c_double * 4)()
    eng.lib.bslv_dist_last_phases(ph4)
    out[rank] = dict(sha=h.hexdigest(), lps=lps, cuts=cuts, local_pivots=piv, nprimal=int(len(d["pu"])), live=int(d["pu"].sum()), nedges=int(len(d["E"])), phases=list(ph4),
                     starts=eng.start_stats())
    eng.close()
    dist_finalize()
    dist.destroy_process_group()


def test_two_ranks_at_the_size_of_the_headline_stay_bit_identical():
    """BASELINE configs[3] in small: S-mid (q = 5, n = 500, m = 1000) as stated, two ranks (two processes sharing the test box's GPU,
    callback transport), 1024 LPs per rank and step, 10 steps: the replicas of the polyhedron are the same bit for bit -- slots,
    flags, coordinates, incidence pairs and the edge list in order (SHA-256 over the raw dump) -- both ranks solved LPs, and the
    phase clocks of the distributed step (bslv_dist_last_phases, what bench.py --gpus N reports per rank) are filled in."""
    import torch.multiprocessing as mp
    batch, steps = 1024, 10
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_smid_worker, args=(2, port, batch, steps, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert a["sha"] == b["sha"], (a, b)
    assert a["lps"] == b["lps"] > 5 * batch and a["cuts"] == b["cuts"] > 0 and a["nprimal"] == b["nprimal"] and a["nedges"] == b["nedges"]
    assert a["local_pivots"] > 0 and b["local_pivots"] > 0
    for r in (a, b):
        assert len(r["phases"]) == 4 and r["phases"][1] > 0 and r["phases"][3] > 0, r["phases"]
