"""ctypes view of oracle/liboracle.so (CPU restatement; TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    L = ctypes.CDLL(LIB)
    vp, i, d, c = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char
    L.olp_create.restype = vp
    L.olp_create.argtypes = [i, i]
    L.olp_free.argtypes = [vp]
    L.olp_load_coo.argtypes = [vp, i, vp, vp, vp]
    L.olp_resize_extra.argtypes = [vp, i, i, i, i]
    L.olp_set_mat_row.argtypes = [vp, i, i, vp, vp]
    L.olp_set_row_bnds.argtypes = [vp, i, c, d, d]
    L.olp_set_col_bnds.argtypes = [vp, i, c, d, d]
    L.olp_set_obj.argtypes = [vp, i, d]
    L.olp_std_basis.argtypes = [vp]
    L.olp_solve.argtypes = [vp, i]
    L.olp_obj_val.argtypes = [vp]
    L.olp_obj_val.restype = d
    for f in ("olp_row_prim", "olp_col_prim", "olp_row_dual", "olp_col_dual"):
        getattr(L, f).restype = d
        getattr(L, f).argtypes = [vp, i]
    L.olp_iterations.restype = ctypes.c_long
    L.olp_iterations.argtypes = [vp]
    L.olp_pivots.restype = ctypes.c_long
    L.olp_pivots.argtypes = [vp]
    _lib = L
    return L


def _btype(lo, up):
    if np.isinf(lo) and np.isinf(up):
        return b"f"
    if np.isinf(up):
        return b"l"
    if np.isinf(lo):
        return b"u"
    return b"s" if lo == up else b"d"


class OracleLP:
    """Dense LP in GLPK's row/col model on the CPU oracle (oracle/lp_dense.c)."""

    def __init__(self, A, lo, up, cost):
        self.L = load()
        M, N = A.shape
        self.M, self.N = M, N
        self.h = self.L.olp_create(M, N)
        ri, ci = np.nonzero(A)
        v = np.ascontiguousarray(A[ri, ci], np.float64)
        ri = np.ascontiguousarray(ri + 1, np.int32)
        ci = np.ascontiguousarray(ci + 1, np.int32)
        self.L.olp_load_coo(self.h, len(v), ri.ctypes.data, ci.ctypes.data, v.ctypes.data)
        for k in range(M):
            self.set_bound(k, lo[k], up[k])
        for k in range(N):
            self.set_bound(M + k, lo[M + k], up[M + k])
        for j in range(N + 1):
            self.L.olp_set_obj(self.h, j, float(cost[j]))

    def set_bound(self, var, lo, up):
        t = _btype(lo, up)
        lo_, up_ = (0.0 if np.isinf(lo) else float(lo)), (0.0 if np.isinf(up) else float(up))
        if var < self.M:
            self.L.olp_set_row_bnds(self.h, var + 1, t, lo_, up_)
        else:
            self.L.olp_set_col_bnds(self.h, var - self.M + 1, t, lo_, up_)

    def solve(self, method=1):
        return self.L.olp_solve(self.h, method)

    def obj(self):
        return self.L.olp_obj_val(self.h)

    def primal(self, first, cnt):
        return np.array([self.L.olp_row_prim(self.h, k + 1) if k < self.M else self.L.olp_col_prim(self.h, k - self.M + 1)
                         for k in range(first, first + cnt)])

    def dual(self, first, cnt):
        return np.array([self.L.olp_row_dual(self.h, k + 1) if k < self.M else self.L.olp_col_dual(self.h, k - self.M + 1)
                         for k in range(first, first + cnt)])

    def pivots(self):
        return self.L.olp_pivots(self.h)

    def close(self):
        if self.h:
            self.L.olp_free(self.h)
            self.h = None


class BensonStats(ctypes.Structure):
    _fields_ = [("lps", ctypes.c_long), ("cuts", ctypes.c_long), ("pivots", ctypes.c_long),
                ("new_vertices", ctypes.c_long), ("secs_total", ctypes.c_double), ("secs_lp", ctypes.c_double),
                ("secs_poly", ctypes.c_double), ("status", ctypes.c_int),
                ("warm_lps", ctypes.c_long), ("warm_cuts", ctypes.c_long), ("warm_pivots", ctypes.c_long),
                ("warm_new_vertices", ctypes.c_long), ("warm_secs", ctypes.c_double)]


def benson_phase2_primal(prob, R=None, c=None, eps=1e-7, max_lps=0, order=0, warm_lps=0):
    """Sequential CPU Benson phase 2 (oracle/benson_cpu.c).  Returns (rc, FlatPoly-like dump, stats).
    order 0 = the reference's vertex order (lowest slot), 1 = newest first; warm_lps: snapshot of the counters after that
    many LPs in stats.warm_* (bench.py rates the LPs behind the warm-up)."""
    import poly_harness as ph
    L = load()
    m, n, q = prob["m"], prob["n"], prob["q"]
    A = np.ascontiguousarray(prob["A"], np.float64)
    P = np.ascontiguousarray(prob["P"], np.float64)
    R = np.eye(q) if R is None else np.ascontiguousarray(R, np.float64)
    c = np.ones(q) if c is None else np.ascontiguousarray(c, np.float64)
    r = R.shape[1]
    f8 = lambda a: np.ascontiguousarray(a, np.float64)
    rt, ct = np.ascontiguousarray(prob["rtype"], np.uint8), np.ascontiguousarray(prob["ctype"], np.uint8)
    rlb, rub, clb, cub = f8(prob["rlb"]), f8(prob["rub"]), f8(prob["clb"]), f8(prob["cub"])
    out = ctypes.c_void_p()
    st = BensonStats()
    L.obenson_phase2_primal_ex.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 8 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                                                     ctypes.c_double, ctypes.c_long, ctypes.c_int, ctypes.c_long,
                                                                                     ctypes.c_void_p, ctypes.c_void_p]
    rc = L.obenson_phase2_primal_ex(m, n, q, A.ctypes.data, P.ctypes.data, rt.ctypes.data, rlb.ctypes.data, rub.ctypes.data,
                                    ct.ctypes.data, clb.ctypes.data, cub.ctypes.data, R.ctypes.data, r, c.ctypes.data,
                                    eps, max_lps, order, warm_lps, ctypes.byref(out), ctypes.byref(st))
    fp = ph.FlatPoly.__new__(ph.FlatPoly)
    fp.L, fp.pre, fp.d, fp.h = L, "opoly_", q, out
    if not getattr(L, "_poly_bound", False):
        ph._bind(L, "opoly_")
        L._poly_bound = True
    return rc, fp, st
