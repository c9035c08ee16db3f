"""CPU, world_size 2, gloo: the one collective of the multi-GPU path (gather of cut records) and the
dealing rule's shard bound.  The engines themselves need a GPU; the N>1 end-to-end run is
tests/test_benson_gpu.py::test_two_ranks_match_single (gloo, two processes on one GPU)."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bensolve_amd.benson import gather_records, shard_capacity


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec_len = q + 5
    n_total = 11
    # rank r owns the records whose source slot is congruent r mod world
    mine = [k for k in range(n_total) if k % world == rank]
    rec = np.zeros((len(mine), rec_len))
    for i, k in enumerate(mine):
        rec[i, 0] = 100 + k
        rec[i, 1] = 4
        rec[i, 2] = k % 2
        rec[i, 3] = 0.5 * k
        rec[i, 4:4 + q] = np.arange(q) + k
        rec[i, 4 + q] = rank
    allrec = gather_records(dist, rec, n_total, rec_len, torch.device("cpu"))
    out[rank] = allrec
    # an empty shard on one rank must work too
    rec2 = rec[:3] if rank == 0 else np.zeros((0, rec_len))
    all2 = gather_records(dist, rec2, 3, rec_len, torch.device("cpu"))
    out[world + rank] = all2
    dist.destroy_process_group()


def test_gather_records_world2():
    world, q = 2, 3
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, q, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert np.array_equal(a, b)                          # every rank sees the same records
    assert sorted(a[:, 0]) == [100 + k for k in range(11)]
    for row in a:
        k = int(row[0]) - 100
        assert row[4 + q] == k % world and row[3] == 0.5 * k and np.array_equal(row[4:4 + q], np.arange(q) + k)
    assert np.array_equal(out[2], out[3]) and len(out[2]) == 3


def test_shard_capacity_bounds_the_dealing_rule():
    for n_total in (0, 1, 7, 64, 1000):
        for world in (1, 2, 4, 8):
            cap = (n_total + world - 1) // world + max(1, n_total // (4 * world))   # bslv_benson_collect
            assert shard_capacity(n_total, world) >= cap
