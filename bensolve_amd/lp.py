"""Host-side mirror of the reference's scalar-LP interface for the hot path.

`P2Model` builds the LP  (P_2(v))  of bslv_algs.c:562-664 (init_P2) in GLPK's row/column
model; `LpEngine` is a thin ctypes view of the batched engine of include/bslv_hip.h section 1.
"""
import ctypes
import numpy as np
from ._lib import load_library, check

INF = float("inf")


def bounds_from_types(types, lb, ub):
    """'f','l','u','d','s' (bslv_lists.h:41-48, bslv_lp.c:34-43) -> (lo, up) arrays."""
    types = np.asarray(types)
    t = np.array([chr(c) if not isinstance(c, str) else c for c in types])
    lo = np.where(np.isin(t, ["l", "d", "s"]), lb, -INF).astype(np.float64)
    up = np.where(np.isin(t, ["u", "d"]), ub, INF).astype(np.float64)
    up = np.where(t == "s", lb, up)
    return lo, up


class P2Model:
    """min z  s.t.  vlp rows/cols,  -P x + y = 0,  R_j.y - z <= ub_j (j<r),  eta.y free.

    Index map (0-based variable ids; aux = rows, structural = cols), Appendix C.2 of SURVEY.md:
      rows 0..m-1 : A x             cols 0..n-1   : x
      rows m..m+q-1 : -P x + y = 0  cols n..n+q-1 : y (free)
      rows m+q..m+q+r-1 : R_j.y - z cols n+q      : z (free, cost 1)
      row  m+q+r : eta.y (free in the inhomogeneous problem)
    `R` is q x r with generators as COLUMNS (ZR[j*p+i], bslv_algs.c:599)."""

    def __init__(self, prob, R=None, eta=None):
        m, n, q = prob["m"], prob["n"], prob["q"]
        A, P = np.asarray(prob["A"], np.float64), np.asarray(prob["P"], np.float64)
        if R is None:
            R = np.eye(q)      # default cone, c = (1..1): Z = I scaled so Z'c = 1 (bslv_vlp.c:661-672,775-792)
        R = np.asarray(R, np.float64)
        r = R.shape[1]
        M, N = m + q + r + 1, n + q + 1
        L = np.zeros((M, N))
        L[:m, :n] = A
        L[m:m + q, :n] = -P
        L[m:m + q, n:n + q] = np.eye(q)
        L[m + q:m + q + r, n:n + q] = R.T
        L[m + q:m + q + r, n + q] = -1.0
        if eta is not None:
            L[m + q + r, n:n + q] = eta
        rlo, rup = bounds_from_types(prob["rtype"], prob["rlb"], prob["rub"])
        clo, cup = bounds_from_types(prob["ctype"], prob["clb"], prob["cub"])
        lo = np.concatenate([rlo, np.zeros(q), np.full(r, -INF), [-INF], clo, np.full(q + 1, -INF)])
        up = np.concatenate([rup, np.zeros(q), np.zeros(r), [INF], cup, np.full(q + 1, INF)])
        cost = np.zeros(N + 1)
        cost[N] = 1.0
        self.m, self.n, self.q, self.r, self.M, self.N = m, n, q, r, M, N
        self.L, self.lo, self.up, self.cost, self.R = L, lo, up, cost, R
        self.var_first = m + q        # aux variables of the r rows
        self.w_first = m              # duals of rows m..m+q-1  (bslv_algs.c:1050)
        self.y_first = M + n          # primals of cols n..n+q-1 (bslv_algs.c:1055)

    def ub_for(self, V):
        """rows->ub[j] = R_j . v (bslv_algs.c:1041-1046); V is B x q."""
        return np.asarray(V, np.float64) @ self.R


class LpEngine:
    def __init__(self, M, N, A, lo, up, cost, var_first, var_cnt, pool_slots):
        self.lib = load_library()
        self.M, self.N, self.vcnt = M, N, var_cnt
        A = np.ascontiguousarray(A, np.float64)
        lo = np.ascontiguousarray(lo, np.float64)
        up = np.ascontiguousarray(up, np.float64)
        cost = np.ascontiguousarray(cost, np.float64)
        assert A.shape == (M, N) and lo.shape == (M + N,) and up.shape == (M + N,) and cost.shape == (N + 1,)
        h = ctypes.c_void_p()
        check(self.lib.bslv_lpq_create(ctypes.byref(h), M, N, A.ctypes.data, lo.ctypes.data, up.ctypes.data,
                                       cost.ctypes.data, var_first, var_cnt, pool_slots))
        self.h = h
        self.pool_slots = pool_slots

    @classmethod
    def from_model(cls, model, pool_slots):
        return cls(model.M, model.N, model.L, model.lo, model.up, model.cost, model.var_first, model.r, pool_slots)

    def close(self):
        if getattr(self, "h", None):
            self.lib.bslv_lpq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot_bytes(self):
        return self.lib.bslv_lpq_slot_bytes(self.h)

    def set_profile(self, on):
        check(self.lib.bslv_lpq_set_profile(self.h, int(on)))

    def reset_slot(self, slot):
        check(self.lib.bslv_lpq_reset_slot(self.h, slot))

    def rows_folded(self):
        import ctypes
        self.lib.bslv_lpq_rows_folded.argtypes = [ctypes.c_void_p]
        return int(self.lib.bslv_lpq_rows_folded(self.h))

    def set_lazy(self, on):
        import ctypes
        self.lib.bslv_lpq_set_lazy.argtypes = [ctypes.c_void_p, ctypes.c_int]
        check(self.lib.bslv_lpq_set_lazy(self.h, int(on)))

    def materialise(self, slots):
        import ctypes
        slots = np.ascontiguousarray(slots, np.int32)
        self.lib.bslv_lpq_materialise.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        check(self.lib.bslv_lpq_materialise(self.h, len(slots), slots.ctypes.data))

    def discard_pending(self):
        import ctypes
        self.lib.bslv_lpq_discard_pending.argtypes = [ctypes.c_void_p]
        check(self.lib.bslv_lpq_discard_pending(self.h))

    def lazy_stats(self):
        import ctypes
        o = (ctypes.c_long * 3)()
        self.lib.bslv_lpq_lazy_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(self.lib.bslv_lpq_lazy_stats(self.h, o))
        return dict(skipped=int(o[0]), on_request=int(o[1]), ms=o[2] / 1000.0)

    def set_bounds(self, lo, up):
        lo = np.ascontiguousarray(lo, np.float64)
        up = np.ascontiguousarray(up, np.float64)
        check(self.lib.bslv_lpq_set_bounds(self.h, lo.ctypes.data, up.ctypes.data))

    def solve_batch(self, src, dst, vlo, vup):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.ascontiguousarray(dst, np.int32)
        B = len(src)
        vlo = np.ascontiguousarray(vlo, np.float64).reshape(B, self.vcnt)
        vup = np.ascontiguousarray(vup, np.float64).reshape(B, self.vcnt)
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        check(self.lib.bslv_lpq_solve_batch(self.h, B, src.ctypes.data, dst.ctypes.data, vlo.ctypes.data,
                                            vup.ctypes.data, status.ctypes.data, iters.ctypes.data))
        return status, iters

    def solve_batch_obj(self, src, dst, cost_first, costs, vlo=None, vup=None):
        """LPs that differ in the objective: costs[b, t] on variable cost_first + t (bslv_lpq_solve_batch_obj)."""
        src = np.ascontiguousarray(src, np.int32)
        dst = np.ascontiguousarray(dst, np.int32)
        B = len(src)
        costs = np.ascontiguousarray(costs, np.float64).reshape(B, -1)
        vlo = np.ascontiguousarray(np.zeros((B, self.vcnt)) if vlo is None else vlo, np.float64).reshape(B, self.vcnt)
        vup = np.ascontiguousarray(np.zeros((B, self.vcnt)) if vup is None else vup, np.float64).reshape(B, self.vcnt)
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        self.lib.bslv_lpq_solve_batch_obj.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
        check(self.lib.bslv_lpq_solve_batch_obj(self.h, B, src.ctypes.data, dst.ctypes.data, vlo.ctypes.data, vup.ctypes.data,
                                                cost_first, costs.shape[1], costs.ctypes.data, status.ctypes.data, iters.ctypes.data))
        return status, iters

    def _get(self, fn, slots, first, cnt):
        slots = np.ascontiguousarray(slots, np.int32)
        out = np.empty((len(slots), cnt), np.float64)
        check(fn(self.h, len(slots), slots.ctypes.data, first, cnt, out.ctypes.data))
        return out

    def primal(self, slots, first, cnt):
        return self._get(self.lib.bslv_lpq_get_primal, slots, first, cnt)

    def dual(self, slots, first, cnt):
        return self._get(self.lib.bslv_lpq_get_dual, slots, first, cnt)

    def obj(self, slots):
        slots = np.ascontiguousarray(slots, np.int32)
        out = np.empty(len(slots), np.float64)
        check(self.lib.bslv_lpq_get_obj(self.h, len(slots), slots.ctypes.data, out.ctypes.data))
        return out

    def set_extended(self, on):
        """the extended selection (perturbation, primal clean-up) for every later solve of this engine (include/bslv_hip.h)"""
        self.lib.bslv_lpq_set_extended.argtypes = [ctypes.c_void_p, ctypes.c_int]
        check(self.lib.bslv_lpq_set_extended(self.h, int(bool(on))))

    def last_stats(self):
        it = ctypes.c_int()
        piv = ctypes.c_long()
        ums = ctypes.c_double()
        tms = ctypes.c_double()
        check(self.lib.bslv_lpq_last_stats(self.h, ctypes.byref(it), ctypes.byref(piv), ctypes.byref(ums), ctypes.byref(tms)))
        ext = (ctypes.c_long * 4)()
        self.lib.bslv_lpq_last_ext_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.lib.bslv_lpq_last_ext_stats(self.h, ext)
        self.lib.bslv_lpq_last_passes.restype = ctypes.c_long
        self.lib.bslv_lpq_last_passes.argtypes = [ctypes.c_void_p]
        self.lib.bslv_lpq_last_launches.restype = ctypes.c_long
        self.lib.bslv_lpq_last_launches.argtypes = [ctypes.c_void_p]
        self.lib.bslv_lpq_last_flip_updates.restype = ctypes.c_long
        self.lib.bslv_lpq_last_flip_updates.argtypes = [ctypes.c_void_p]
        return dict(lockstep_iters=it.value, pivots=piv.value, update_ms=ums.value, total_ms=tms.value, passes=self.lib.bslv_lpq_last_passes(self.h), launches=self.lib.bslv_lpq_last_launches(self.h),
                    flip_iterations=ext[0], perturbations=ext[1], primal_steps=ext[2], wrong_sign_removals=ext[3],
                    flip_vector_updates=self.lib.bslv_lpq_last_flip_updates(self.h))
