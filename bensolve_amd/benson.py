"""Host-side mirror of the reference's phase-2 driver (bslv_algs.c:958-1082) over the batched engine.

Single process: `BensonEngine.step()`.  Multi-GPU: one process per GPU, `step_distributed()` does
collect -> solve_local -> ONE all_gather of the fixed-size cut records (torch.distributed; backend
"nccl" = RCCL on GPU, "gloo" in the CPU tests) -> apply on every rank."""
import ctypes
import numpy as np
from ._lib import load_library, check
from .poly import PolyEngine, _bind as _bind_poly
from .lp import LpEngine

_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    vp, i, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.bslv_benson_create.argtypes = [ctypes.POINTER(vp), i, i, i] + [vp] * 8 + [vp, i, vp, d, i]
    lib.bslv_benson_destroy.argtypes = [vp]
    lib.bslv_benson_destroy.restype = None
    lib.bslv_benson_start.argtypes = [vp, vp]
    lib.bslv_benson_collect.argtypes = [vp, i, i, i, vp, vp]
    lib.bslv_benson_record_len.argtypes = [vp]
    lib.bslv_benson_solve_local.argtypes = [vp, vp, vp, vp]
    lib.bslv_benson_apply.argtypes = [vp, i, vp, vp]
    lib.bslv_benson_step.argtypes = [vp, i, vp, vp]
    lib.bslv_benson_unprocessed_left.argtypes = [vp]
    lib.bslv_benson_set_pipelined.argtypes = [vp, i]
    lib.bslv_benson_collect_ctx.argtypes = [vp, i, i, i, i, vp, vp]
    lib.bslv_benson_solve_local_ctx.argtypes = [vp, i, vp, vp, vp]
    lib.bslv_benson_apply_ctx.argtypes = [vp, i, i, vp, vp]
    lib.bslv_benson_set_policy.argtypes = [vp, i]
    lib.bslv_benson_totals.argtypes = [vp, vp, vp, vp]
    lib.bslv_benson_poly.argtypes = [vp]
    lib.bslv_benson_poly.restype = vp
    lib.bslv_benson_lp.argtypes = [vp]
    lib.bslv_benson_lp.restype = vp
    _bound = True


class BensonEngine:
    def __init__(self, prob, R=None, c=None, eps=1e-7, pool_slots=256):
        self.lib = load_library()
        _bind(self.lib)
        _bind_poly(self.lib)
        m, n, q = prob["m"], prob["n"], prob["q"]
        self.q = q
        f8 = lambda a: np.ascontiguousarray(a, np.float64)
        A, P = f8(prob["A"]), f8(prob["P"])
        R = np.eye(q) if R is None else f8(R)
        c = np.ones(q) if c is None else f8(c)
        rt = np.ascontiguousarray(prob["rtype"], np.uint8)
        ct = np.ascontiguousarray(prob["ctype"], np.uint8)
        rlb, rub, clb, cub = f8(prob["rlb"]), f8(prob["rub"]), f8(prob["clb"]), f8(prob["cub"])
        h = ctypes.c_void_p()
        check(self.lib.bslv_benson_create(ctypes.byref(h), m, n, q, A.ctypes.data, P.ctypes.data, rt.ctypes.data,
                                          rlb.ctypes.data, rub.ctypes.data, ct.ctypes.data, clb.ctypes.data, cub.ctypes.data,
                                          R.ctypes.data, R.shape[1], c.ctypes.data, eps, pool_slots))
        self.h = h
        self.rec_len = self.lib.bslv_benson_record_len(h)
        # non-owning views of the two engines (for dumps / stats)
        self.poly = PolyEngine.__new__(PolyEngine)
        self.poly.lib, self.poly.d, self.poly.h = self.lib, q, None
        self._poly_h = ctypes.c_void_p(self.lib.bslv_benson_poly(h))
        self.lp = LpEngine.__new__(LpEngine)
        self.lp.lib, self.lp.h = self.lib, None
        self._lp_h = ctypes.c_void_p(self.lib.bslv_benson_lp(h))

    # views that do not destroy the borrowed handles
    def poly_dump(self):
        self.poly.h = self._poly_h
        try:
            return self.poly.dump()
        finally:
            self.poly.h = None

    def poly_call(self, name, *a):
        self.poly.h = self._poly_h
        try:
            return getattr(self.poly, name)(*a)
        finally:
            self.poly.h = None

    def lp_call(self, name, *a):
        self.lp.h = self._lp_h
        try:
            return getattr(self.lp, name)(*a)
        finally:
            self.lp.h = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.bslv_benson_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_policy(self, policy):
        check(self.lib.bslv_benson_set_policy(self.h, int(policy)))

    def set_sibling_rule(self, cap, window):
        self.lib.bslv_benson_set_sibling_rule.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        check(self.lib.bslv_benson_set_sibling_rule(self.h, int(cap), int(window)))

    def set_fronts(self, nfronts, sib_cap=1 << 20):
        self.lib.bslv_benson_set_fronts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        check(self.lib.bslv_benson_set_fronts(self.h, int(nfronts), int(sib_cap)))

    def set_families(self, mode, batches):
        self.lib.bslv_benson_set_families.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        check(self.lib.bslv_benson_set_families(self.h, int(mode), int(batches)))

    def pool_stats(self):
        out = (ctypes.c_long * 4)()
        self.lib.bslv_benson_pool_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(self.lib.bslv_benson_pool_stats(self.h, out))
        return dict(zip(("free", "resident", "held", "pool"), list(out)))

    def start(self):
        st = ctypes.c_int()
        check(self.lib.bslv_benson_start(self.h, ctypes.byref(st)))
        return st.value

    def step(self, max_batch):
        stats = (ctypes.c_long * 8)()
        ms = (ctypes.c_double * 3)()
        check(self.lib.bslv_benson_step(self.h, max_batch, stats, ms))
        keys = ("lps", "cuts", "redundant", "confirmed", "failed", "pivots", "lockstep", "left")
        out = dict(zip(keys, list(stats)))
        out.update(ms_lp=ms[0], ms_poly=ms[1], ms_total=ms[2])
        return out

    def set_pipelined(self, on):
        check(self.lib.bslv_benson_set_pipelined(self.h, int(on)))

    def collect(self, max_batch, rank=0, world=1, ctx=0):
        nl, nt = ctypes.c_int(), ctypes.c_int()
        check(self.lib.bslv_benson_collect_ctx(self.h, ctx, max_batch, rank, world, ctypes.byref(nl), ctypes.byref(nt)))
        return nl.value, nt.value

    def solve_local(self, n_local, ctx=0):
        rec = np.zeros((max(n_local, 1), self.rec_len))
        piv, ls = ctypes.c_int(), ctypes.c_int()
        check(self.lib.bslv_benson_solve_local_ctx(self.h, ctx, rec.ctypes.data, ctypes.byref(piv), ctypes.byref(ls)))
        return rec[:n_local], piv.value, ls.value

    def apply(self, records, ctx=0):
        records = np.ascontiguousarray(records, np.float64).reshape(-1, self.rec_len)
        stats = (ctypes.c_long * 5)()
        check(self.lib.bslv_benson_apply_ctx(self.h, ctx, len(records), records.ctypes.data, stats))
        return dict(zip(("lps", "cuts", "redundant", "confirmed", "failed"), list(stats)))

    def lp_dims(self):
        M, N, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.lib.bslv_benson_lp_dims.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
        check(self.lib.bslv_benson_lp_dims(self.h, ctypes.byref(M), ctypes.byref(N), ctypes.byref(f)))
        return dict(M=M.value, N=N.value, rows_folded=f.value)

    def totals(self):
        a, b, c = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
        check(self.lib.bslv_benson_totals(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(lps=a.value, cuts=b.value, pivots=c.value)

    def start_stats(self):
        a, b = ctypes.c_long(), ctypes.c_long()
        self.lib.bslv_benson_start_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        check(self.lib.bslv_benson_start_stats(self.h, ctypes.byref(a), ctypes.byref(b)))
        return dict(root=a.value, nearest=b.value)

    def set_defer(self, min_cuts):
        self.lib.bslv_benson_set_defer.argtypes = [ctypes.c_void_p, ctypes.c_int]
        check(self.lib.bslv_benson_set_defer(self.h, int(min_cuts)))

    def defer_stats(self):
        out = (ctypes.c_long * 4)()
        self.lib.bslv_benson_defer_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(self.lib.bslv_benson_defer_stats(self.h, out))
        return dict(handed_back=out[0], waiting=out[1], flushes=out[2], one_family_batches=out[3])

    def run(self, max_batch, max_steps=None):
        """run to termination (poly__get_vrtx returns 'none left', bslv_algs.c:1032-1035)"""
        steps = 0
        while max_steps is None or steps < max_steps:
            s = self.step(max_batch)
            steps += 1
            if s["lps"] == 0 and s["left"] == 0:
                break
        return steps

    def step_distributed(self, max_batch, dist, device):
        """one outer iteration over all ranks of `dist` (torch.distributed): the batch is dealt to
        ranks, each solves its shard, ONE all_gather of the padded record blocks, all apply."""
        rank, world = dist.get_rank(), dist.get_world_size()
        n_local, n_total = self.collect(max_batch, rank, world)
        rec, piv, ls = self.solve_local(n_local)
        allrec = gather_records(dist, rec, n_total, self.rec_len, device)
        st = self.apply(allrec)
        st.update(n_local=n_local, n_total=n_total, pivots=piv, lockstep=ls)
        return st


def dist_init_rccl(dist, device):
    """Join the library's own RCCL communicator (include/bslv_hip.h section 4a): rank 0 creates the ncclUniqueId, ONE
    broadcast over `dist` (torch.distributed, any backend) hands it to the others; from then on BensonEngine.step() runs the
    exchange step inside libbslv_hip.so (bslv_benson_step_dist: direct ncclAllGather over xGMI).
    Every rank first checks that it can load RCCL and make an id at all; only if ALL can is the communicator created -- a rank
    that cannot would leave the others waiting in ncclCommInitRank.  Otherwise every rank falls back to the callback transport
    (the same C step, all-gather through torch.distributed on device tensors).  Returns the name of the transport in use."""
    import torch
    lib = load_library()
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = (ctypes.c_ubyte * 128)()
    ok, why = 1, ""
    try:
        check(lib.bslv_dist_unique_id(buf, 128))
    except Exception as e:                               # RCCL not loadable on this rank
        ok, why = 0, str(e)
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        dist_init_callback(dist, device)
        return "torch.distributed all_gather through bslv_dist_init_callback (RCCL not loadable on some rank%s)" % ((": " + why[:120]) if why else "")
    t = torch.tensor(list(buf), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    idb = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
    check(lib.bslv_dist_init(rank, world, idb, 128))
    return "ncclAllGather (RCCL) inside libbslv_hip.so"


_cb_keep = []


def dist_init_callback(dist, device=None):
    """The same exchange step over a caller-supplied all-gather (here: torch.distributed, e.g. gloo): for tests on a one-GPU
    box, where RCCL refuses two ranks on one device.  device: stage the buffers on that device (needed when the process group's
    backend is nccl, which does not take host tensors)."""
    import torch
    lib = load_library()
    rank, world = dist.get_rank(), dist.get_world_size()
    FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p)

    def gather(send, recv, count, ctx):
        try:
            a = torch.from_numpy(np.ctypeslib.as_array(send, shape=(count,)).copy())
            if device is not None:
                a = a.to(device)
            out = [torch.empty_like(a) for _ in range(world)]
            dist.all_gather(out, a)
            np.ctypeslib.as_array(recv, shape=(count * world,))[:] = torch.cat(out).cpu().numpy()
            return 0
        except Exception:
            return 1
    cb = FN(gather)
    _cb_keep.append(cb)
    lib.bslv_dist_init_callback.argtypes = [ctypes.c_int, ctypes.c_int, FN, ctypes.c_void_p]
    check(lib.bslv_dist_init_callback(rank, world, cb, None))


def dist_finalize():
    lib = load_library()
    lib.bslv_dist_finalize.restype = None
    lib.bslv_dist_finalize()


class PipelinedStepper:
    """Single-process software pipeline: the LPs of batch k (LP engine, second host thread; ctypes releases
    the GIL) overlap with the cut application of batch k-1 (polyhedron engine).  Batch members are marked
    when collected, so batch k never contains a vertex of batch k-1."""

    def __init__(self, eng, max_batch):
        self.eng, self.B, self.k, self.pending = eng, max_batch, 0, None
        eng.set_pipelined(True)

    def step(self):
        import threading
        eng, ctx = self.eng, self.k & 1
        self.k += 1
        nl, nt = eng.collect(self.B, 0, 1, ctx=ctx)
        box = {}

        def work():
            try:
                box["res"] = eng.solve_local(nl, ctx=ctx)
            except Exception as e:          # re-raised on the main thread
                box["err"] = e
        th = threading.Thread(target=work)
        th.start()
        st = dict(lps=0, cuts=0, redundant=0, confirmed=0, failed=0)
        applied = self.pending is not None
        try:
            if self.pending is not None:
                st = eng.apply(self.pending[1], ctx=self.pending[0])
        finally:
            th.join()
        if "err" in box:
            raise box["err"]
        rec, piv, ls = box["res"]
        self.pending = (ctx, rec) if nl else None
        st.update(n_local=nl, n_total=nt, pivots=piv, lockstep=ls, lps_solved=nl, applied=applied)
        return st

    def flush(self):
        st = None
        if self.pending is not None:
            st = self.eng.apply(self.pending[1], ctx=self.pending[0])
            self.pending = None
        return st

    def run(self, max_steps=100000):
        """to termination: nothing left to collect and nothing pending"""
        steps = 0
        while steps < max_steps:
            s = self.step()
            steps += 1
            if s["n_total"] == 0 and not s["applied"]:      # nothing collected and no cuts applied since: done
                break
        return steps


def shard_capacity(n_total, world):
    """upper bound of one rank's shard (the dealing rule of bslv_benson_collect)"""
    return (n_total + world - 1) // world + max(1, n_total // (4 * world)) + 1


def gather_records(dist, rec, n_total, rec_len, device):
    """The ONE collective of an outer iteration: every rank contributes a fixed-size
    block [count ; records ; zero padding] and receives everybody's.  Returns the concatenated
    records of all ranks (rank order; apply() sorts by source slot)."""
    import torch
    world = dist.get_world_size()
    rec = np.asarray(rec, np.float64).reshape(-1, rec_len)
    n_local = len(rec)
    cap = shard_capacity(n_total, world)
    assert n_local <= cap, "shard larger than the dealing rule allows"
    block = torch.zeros((cap + 1, rec_len), dtype=torch.float64)
    block[0, 0] = n_local
    if n_local:
        block[1:1 + n_local] = torch.from_numpy(rec)
    block = block.to(device)
    gathered = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(gathered, block)
    parts = []
    for g in gathered:
        g = g.cpu().numpy()
        parts.append(g[1:1 + int(g[0, 0])])
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, rec_len))
