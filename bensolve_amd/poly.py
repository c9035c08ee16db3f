"""Host-side mirror of the reference's polyhedron interface (bslv_poly.h:90-118) over the HIP engine."""
import ctypes
import numpy as np
from ._lib import load_library, check

CONE_POLAR, LOWER2UPPER, UPPER2LOWER = 0, 1, 2
_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.bslv_poly_create.argtypes = [ctypes.POINTER(vp), i, i, vp]
    lib.bslv_poly_destroy.argtypes = [vp]
    lib.bslv_poly_destroy.restype = None
    lib.bslv_poly_dual0_apex.argtypes = [vp]
    lib.bslv_poly_add.argtypes = [vp, vp, i, vp]
    lib.bslv_poly_add_cuts.argtypes = [vp, i, vp, vp, vp]
    lib.bslv_poly_init.argtypes = [vp, vp]
    lib.bslv_poly_set_batch_mode.argtypes = [vp, i]
    lib.bslv_poly_rounds_run.argtypes = [vp]
    lib.bslv_poly_rounds_run.restype = ctypes.c_long
    lib.bslv_poly_debug_set.argtypes = [vp, ctypes.c_int, ctypes.c_long]
    lib.bslv_poly_debug_set.restype = ctypes.c_int
    lib.bslv_poly_path_stats.argtypes = [vp, vp]
    lib.bslv_poly_path_stats.restype = ctypes.c_int
    lib.bslv_poly_next.argtypes = [vp, vp, vp, vp, vp]
    lib.bslv_poly_unprocessed.argtypes = [vp, i, vp, vp, vp, vp]
    lib.bslv_poly_mark.argtypes = [vp, i, vp]
    lib.bslv_poly_dual_adjacency.argtypes = [vp]
    lib.bslv_poly_classify_batch.argtypes = [vp, i, vp, vp, vp, i, vp]
    if hasattr(lib, "bslv_poly_classify_batch_touch"):                  # (TEST entry; absent from older builds used in A/B probes)
        lib.bslv_poly_classify_batch_touch.argtypes = [vp, i, vp, vp, vp, vp]
    lib.bslv_poly_bench_fill.argtypes = [vp, i, ctypes.c_ulonglong]
    for n in ("dim", "nprimal", "ndual"):
        getattr(lib, "bslv_poly_" + n).argtypes = [vp]
    for n in ("nedges", "ninc", "ndual_edges", "pair_tests", "new_vertices"):
        f = getattr(lib, "bslv_poly_" + n)
        f.argtypes = [vp]
        f.restype = ctypes.c_long
    lib.bslv_poly_get_primal.argtypes = [vp, vp, vp, vp, vp]
    lib.bslv_poly_get_dual.argtypes = [vp, vp, vp, vp]
    lib.bslv_poly_get_edges.argtypes = [vp, vp]
    lib.bslv_poly_get_inc.argtypes = [vp, vp]
    lib.bslv_poly_get_dual_edges.argtypes = [vp, vp]
    _bound = True


class PolyEngine:
    """poly_args-like object: primal polyhedron + dual, living in HBM."""

    def __init__(self, dim, v2h=CONE_POLAR, c=None):
        self.lib = load_library()
        _bind(self.lib)
        self.d = dim
        cc = None if c is None else np.ascontiguousarray(c, np.float64)
        h = ctypes.c_void_p()
        check(self.lib.bslv_poly_create(ctypes.byref(h), dim, v2h, None if cc is None else cc.ctypes.data))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.bslv_poly_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dual0_apex(self):
        check(self.lib.bslv_poly_dual0_apex(self.h))

    def add(self, val, ideal=0):
        """poly__add_vrtx: 0 = added/queued, 1 = redundant"""
        v = np.ascontiguousarray(val, np.float64)
        rc = ctypes.c_int()
        check(self.lib.bslv_poly_add(self.h, v.ctypes.data, int(ideal), ctypes.byref(rc)))
        return rc.value

    def add_cuts(self, vals, ideals=None):
        vals = np.ascontiguousarray(vals, np.float64).reshape(-1, self.d)
        B = len(vals)
        rc = np.zeros(B, np.int32)
        idl = None if ideals is None else np.ascontiguousarray(ideals, np.int32)
        check(self.lib.bslv_poly_add_cuts(self.h, B, vals.ctypes.data, None if idl is None else idl.ctypes.data, rc.ctypes.data))
        return rc

    def set_batch_mode(self, mode):
        check(self.lib.bslv_poly_set_batch_mode(self.h, int(mode)))

    def rounds_run(self):
        return self.lib.bslv_poly_rounds_run(self.h)

    def debug_set(self, key, value):
        check(self.lib.bslv_poly_debug_set(self.h, int(key), int(value)))

    def rounds2_stats(self):
        out = (ctypes.c_long * 5)()
        self.lib.bslv_poly_rounds2_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(self.lib.bslv_poly_rounds2_stats(self.h, out))
        return dict(rounds=out[0], cuts=out[1], chunks=out[2], fallback_prunes=out[3], declined=out[4])

    def rounds2_health(self):
        """(cuts handed back late, mailbox reads repeated because the sequence number arrived before the state)"""
        for f in ("bslv_poly_rounds2_late_left", "bslv_poly_rounds2_torn_reads"):
            getattr(self.lib, f).restype = ctypes.c_long
            getattr(self.lib, f).argtypes = [ctypes.c_void_p]
        return dict(late_left=self.lib.bslv_poly_rounds2_late_left(self.h), torn_reads=self.lib.bslv_poly_rounds2_torn_reads(self.h))

    def sharded_prunes(self):
        """multi-GPU: adjacency prunes whose pair space was dealt to the ranks"""
        self.lib.bslv_poly_sharded_prunes.restype = ctypes.c_long
        self.lib.bslv_poly_sharded_prunes.argtypes = [ctypes.c_void_p]
        return self.lib.bslv_poly_sharded_prunes(self.h)

    def reserve(self, elements=0, edges=0, pool_words=0):
        """capacity ahead of need (bslv_poly_reserve): the arrays otherwise double when they fill up, in the middle of a batch of cuts"""
        self.lib.bslv_poly_reserve.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long]
        self.lib.bslv_poly_reserve.restype = ctypes.c_int
        check(self.lib.bslv_poly_reserve(self.h, elements, edges, pool_words))

    def set_snap(self, on=1):
        """the projection sub-band of poly__cut (bslv_poly.c:666-674); cuts are then applied one at a time"""
        self.lib.bslv_poly_set_snap.argtypes = [ctypes.c_void_p, ctypes.c_int]
        check(self.lib.bslv_poly_set_snap(self.h, on))

    def snapped(self):
        out = ctypes.c_long()
        self.lib.bslv_poly_snapped.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(self.lib.bslv_poly_snapped(self.h, ctypes.byref(out)))
        return out.value

    def path_stats(self):
        out = (ctypes.c_long * 6)()
        check(self.lib.bslv_poly_path_stats(self.h, out))
        return dict(hot_chunks=out[0], speculative=out[1], declined=out[2], prune_fallbacks=out[3], single_cuts=out[4], member_list_prunes=out[5])

    def init(self):
        rc = ctypes.c_int()
        check(self.lib.bslv_poly_init(self.h, ctypes.byref(rc)))
        return rc.value

    def next(self):
        v = np.empty(self.d)
        ideal, idx, rc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self.lib.bslv_poly_next(self.h, v.ctypes.data, ctypes.byref(ideal), ctypes.byref(idx), ctypes.byref(rc)))
        return None if rc.value else (v, ideal.value, idx.value)

    def unprocessed(self, max_out=None):
        cnt = ctypes.c_int()
        check(self.lib.bslv_poly_unprocessed(self.h, 0, None, None, None, ctypes.byref(cnt)))
        n = cnt.value if max_out is None else min(cnt.value, max_out)
        idx = np.empty(n, np.int32)
        val = np.empty((n, self.d))
        ideal = np.empty(n, np.int32)
        if n:
            check(self.lib.bslv_poly_unprocessed(self.h, n, idx.ctypes.data, val.ctypes.data, ideal.ctypes.data, ctypes.byref(cnt)))
        return idx, val, ideal, cnt.value

    def mark(self, idx):
        idx = np.ascontiguousarray(np.atleast_1d(idx), np.int32)
        check(self.lib.bslv_poly_mark(self.h, len(idx), idx.ctypes.data))

    def dual_adjacency(self):
        check(self.lib.bslv_poly_dual_adjacency(self.h))

    def classify_batch(self, hps, repeats=1, fetch=True):
        hps = np.ascontiguousarray(hps, np.float64).reshape(-1, self.d + 1)
        B = len(hps)
        nv = self.lib.bslv_poly_nprimal(self.h)
        words = np.zeros(((B + 31) // 32, nv), np.uint64) if fetch else None
        anym = np.zeros(B, np.int32)
        ms = ctypes.c_float()
        check(self.lib.bslv_poly_classify_batch(self.h, B, hps.ctypes.data, None if words is None else words.ctypes.data,
                                                anym.ctypes.data, repeats, ctypes.byref(ms)))
        return words, anym, ms.value

    def classify_batch_touch(self, hps):
        """TEST: K1 as the chunked cut application launches it -> (words, touch counts, first touched halfspace)"""
        hps = np.ascontiguousarray(hps, np.float64).reshape(-1, self.d + 1)
        B = len(hps)
        nv = self.lib.bslv_poly_nprimal(self.h)
        words = np.zeros(((B + 31) // 32, nv), np.uint64)
        tc, t1 = np.zeros(nv, np.int32), np.zeros(nv, np.int32)
        check(self.lib.bslv_poly_classify_batch_touch(self.h, B, hps.ctypes.data, words.ctypes.data, tc.ctypes.data, t1.ctypes.data))
        return words, tc, t1

    def bench_fill(self, nv, seed=1):
        check(self.lib.bslv_poly_bench_fill(self.h, int(nv), int(seed)))

    def counts(self):
        L, h = self.lib, self.h
        return dict(nprimal=L.bslv_poly_nprimal(h), ndual=L.bslv_poly_ndual(h), nedges=L.bslv_poly_nedges(h),
                    pair_tests=L.bslv_poly_pair_tests(h), new_vertices=L.bslv_poly_new_vertices(h))

    def dump(self):
        L, h, d = self.lib, self.h, self.d
        nv, nf = L.bslv_poly_nprimal(h), L.bslv_poly_ndual(h)
        pu, pi, ps = (np.zeros(nv, np.uint8) for _ in range(3))
        X = np.zeros((nv, d))
        check(L.bslv_poly_get_primal(h, pu.ctypes.data, pi.ctypes.data, ps.ctypes.data, X.ctypes.data))
        du, di = np.zeros(nf, np.uint8), np.zeros(nf, np.uint8)
        Y = np.zeros((nf, d))
        check(L.bslv_poly_get_dual(h, du.ctypes.data, di.ctypes.data, Y.ctypes.data))
        E = np.zeros((L.bslv_poly_nedges(h), 2), np.int32)
        check(L.bslv_poly_get_edges(h, E.ctypes.data))
        I = np.zeros((L.bslv_poly_ninc(h), 2), np.int32)
        check(L.bslv_poly_get_inc(h, I.ctypes.data))
        DE = np.zeros((L.bslv_poly_ndual_edges(h), 2), np.int32)
        check(L.bslv_poly_get_dual_edges(h, DE.ctypes.data))
        return dict(d=d, pu=pu, pi=pi, ps=ps, X=X, du=du, di=di, Y=Y, E=E, I=I, DE=DE)
