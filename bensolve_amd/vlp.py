"""ctypes mirror of the callers around phase 2: bslv_vlp_solve_primal (sol_init + phases 0/1/2, bslv_main.c:236-345)
and bslv_cone_vertenum (bslv_algs.c:331-407).  Generator matrices are q x k, generators as columns."""
import ctypes
import numpy as np

from ._lib import load_library, check
from .poly import PolyEngine, _bind as _bind_poly


class VlpInfo(ctypes.Structure):
    _fields_ = [("q", ctypes.c_int), ("o", ctypes.c_int), ("p", ctypes.c_int), ("r", ctypes.c_int), ("h", ctypes.c_int),
                ("c_dir", ctypes.c_int), ("negate_primal", ctypes.c_int), ("negate_dual_last", ctypes.c_int),
                ("lps", ctypes.c_long), ("steps", ctypes.c_long)] + \
               [(k, ctypes.POINTER(ctypes.c_double)) for k in ("c", "eta", "R", "H", "Y", "Z")] + [("message", ctypes.c_char * 160)]


STATUS = {1: "infeasible", 2: "unbounded", 3: "no vertex", 4: "optimal", 5: "input error"}


def cone_vertenum(gen):
    """gen: dim x n_in.  Returns (prim dim x n_prim, dual dim x n_dual) or None when the cone has no interior."""
    lib = load_library()
    gen = np.ascontiguousarray(gen, np.float64)
    dim, n_in = gen.shape
    prim, dual = ctypes.POINTER(ctypes.c_double)(), ctypes.POINTER(ctypes.c_double)()
    npr, ndu, rc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.bslv_cone_vertenum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    lib.bslv_free.argtypes = [ctypes.c_void_p]
    lib.bslv_free.restype = None
    check(lib.bslv_cone_vertenum(gen.ctypes.data, n_in, dim, ctypes.byref(prim), ctypes.byref(npr), ctypes.byref(dual), ctypes.byref(ndu), ctypes.byref(rc)))
    if rc.value:
        return None
    a = np.ctypeslib.as_array(prim, shape=(dim * max(npr.value, 1),))[: dim * npr.value].reshape(dim, npr.value).copy()
    b = np.ctypeslib.as_array(dual, shape=(dim * max(ndu.value, 1),))[: dim * ndu.value].reshape(dim, ndu.value).copy()
    lib.bslv_free(prim); lib.bslv_free(dual)
    return a, b


def solve_primal(prob, cone_kind=0, gen=None, c=None, bounded=False, eps_phase0=1e-8, eps_phase1=1e-8, eps_benson_phase1=1e-7,
                 eps_benson_phase2=1e-7, batch=256, alg_phase2="primal", alg_phase1="primal", preimages=False):
    """prob: dict as bensolve_amd.synth builds them (P as in the file: not negated).  Returns dict(status, message, info...,
    dump = slot-indexed dump of the result polyhedron with the sign changes of poly_trans_primal applied).  alg_phase2 = "dual":
    phase2_dual; the dump is then given with the sides swapped back, so that X / pi are the upper image in both cases."""
    dual2 = alg_phase2 == "dual"
    lib = load_library()
    vp = ctypes.c_void_p
    A = np.ascontiguousarray(prob["A"], np.float64); P = np.ascontiguousarray(prob["P"], np.float64)
    rt = np.ascontiguousarray(prob["rtype"], np.uint8); ct = np.ascontiguousarray(prob["ctype"], np.uint8)
    arrs = [np.ascontiguousarray(prob[k], np.float64) for k in ("rlb", "rub", "clb", "cub")]
    g = None if gen is None else np.ascontiguousarray(gen, np.float64)
    cc = None if c is None else np.ascontiguousarray(c, np.float64)
    h = vp()
    st = ctypes.c_int()
    info = VlpInfo()
    lib.bslv_vlp_solve_primal.argtypes = [ctypes.c_int] * 3 + [vp] * 8 + [ctypes.c_int] * 2 + [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int] + [ctypes.c_double] * 4 + [ctypes.c_int, vp, vp, vp]
    lib.bslv_vlp_info_free.argtypes = [vp]
    lib.bslv_vlp_info_free.restype = None
    lib.bslv_vlp_solve_dual2.argtypes = lib.bslv_vlp_solve_primal.argtypes
    check((lib.bslv_vlp_solve_dual2 if dual2 else lib.bslv_vlp_solve_primal)(prob["m"], prob["n"], prob["q"], A.ctypes.data, P.ctypes.data, rt.ctypes.data, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                    ct.ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, int(prob.get("optdir", 1)), cone_kind,
                                    None if g is None else g.ctypes.data, 0 if g is None else g.shape[1], None if cc is None else cc.ctypes.data,
                                    int(bounded), (1 if alg_phase1 == "dual" else 0) | (2 if preimages else 0), eps_phase0, eps_phase1, eps_benson_phase1, eps_benson_phase2, batch,
                                    ctypes.byref(h), ctypes.byref(st), ctypes.byref(info)))
    q = prob["q"]
    out = dict(status=STATUS.get(st.value, st.value), message=info.message.decode(), lps=info.lps, steps=info.steps, c_dir=info.c_dir)

    def mat(ptr, k):
        return np.ctypeslib.as_array(ptr, shape=(q * k,)).reshape(q, k).copy() if ptr and k > 0 else None
    if st.value == 4:
        out.update(c=mat(info.c, 1).ravel(), eta=mat(info.eta, 1).ravel(), R=mat(info.R, info.r), H=mat(info.H, info.h), Y=mat(info.Y, info.o), Z=mat(info.Z, info.p))
        lib.bslv_benson_poly.restype = vp
        lib.bslv_benson_poly.argtypes = [vp]
        _bind_poly(lib)
        pe = PolyEngine.__new__(PolyEngine)
        pe.lib, pe.h, pe.d = lib, (h if dual2 else vp(lib.bslv_benson_poly(h))), q
        pe.dual_adjacency()
        d = pe.dump()
        pe.h = None                       # (destroyed below)
        if dual2:                         # primal side = lower image: swap the sides of the dump
            d = dict(d=q, pu=d["du"], pi=d["di"], ps=np.zeros_like(d["du"]), X=d["Y"], du=d["pu"], di=d["pi"], Y=d["X"],
                     E=d["DE"], DE=d["E"], I=d["I"][:, ::-1].copy())
        if info.negate_primal:
            d["X"] = -d["X"]
        if info.negate_dual_last:
            d["Y"][:, -1] = -d["Y"][:, -1]
        out["dump"] = d
        lib.bslv_benson_destroy.argtypes = [vp]
        lib.bslv_benson_destroy.restype = None
        lib.bslv_poly_destroy.argtypes = [vp]
        lib.bslv_poly_destroy.restype = None
        (lib.bslv_poly_destroy if dual2 else lib.bslv_benson_destroy)(h)
    lib.bslv_vlp_info_free(ctypes.byref(info))
    return out
