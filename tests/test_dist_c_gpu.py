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


def _rank_worker(rank, world, port, args, batch, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dist_init_callback(dist)
    prob = synth.covering_vlp(*args)
    eng = BensonEngine(prob, eps=1e-9, pool_slots=512)
    assert eng.start() == 0
    eng.run(batch)                      # bslv_benson_step routes to bslv_benson_step_dist once a transport is set
    eng.poly_call("dual_adjacency")
    d = eng.poly_dump()
    tot = eng.totals()
    out[rank] = dict(d, lps=tot["lps"], local_pivots=tot["pivots"])
    eng.close()
    dist_finalize()
    dist.destroy_process_group()


@pytest.mark.parametrize("args,batch", [((30, 15, 3, 5), 32), ((40, 20, 4, 9), 128)])
def test_two_ranks_through_the_c_step(args, batch):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_worker, args=(2, port, args, batch, out), nprocs=2, join=True)
    d0, d1 = out[0], out[1]
    for k in ("pu", "pi", "ps", "E", "I", "X", "Y", "du", "DE"):
        assert np.array_equal(d0[k], d1[k]), "replicas diverged in " + k
    assert d0["lps"] == d1["lps"] and d0["local_pivots"] > 0 and d1["local_pivots"] > 0        # both ranks solved LPs
    prob = synth.covering_vlp(*args)
    eng = BensonEngine(prob, eps=1e-9, pool_slots=512)
    assert eng.start() == 0
    eng.run(batch)
    eng.poly_call("dual_adjacency")
    single = ph.canonical(eng.poly_dump(), decimals=6)
    eng.close()
    ph.assert_benson_results_agree(ph.canonical(dict((k, d0[k]) for k in ("d", "pu", "pi", "ps", "X", "du", "di", "Y", "E", "I", "DE")), decimals=6), single)
