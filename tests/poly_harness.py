"""Run cut sequences through a polyhedron engine (oracle restatement, compiled reference, or the
HIP engine) behind one flat API and canonicalise the result for set-wise comparison
(SURVEY.md section 8c 'comparison rule')."""
import ctypes
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_POLY = os.path.join(ROOT, "oracle", "_ref", "libref_poly.so")
REF_POLY_HIP = os.path.join(ROOT, "oracle", "_ref", "libref_poly_hip.so")

CONE_POLAR, LOWER2UPPER, UPPER2LOWER = 0, 1, 2


def _bind(L, pre):
    vp, i = ctypes.c_void_p, ctypes.c_int
    g = lambda n: getattr(L, pre + n)
    g("create").restype = vp
    g("create").argtypes = [i, i, vp]
    g("free").argtypes = [vp]
    g("dual0_apex").argtypes = [vp]
    g("add").argtypes = [vp, vp, i]
    g("init").argtypes = [vp]
    g("next").argtypes = [vp, vp, vp, vp]
    g("mark").argtypes = [vp, i]
    g("dual_adjacency").argtypes = [vp]
    for n in ("dim", "nprimal", "ndual"):
        g(n).argtypes = [vp]
    for n in ("nedges", "ninc", "ndual_edges"):
        g(n).argtypes = [vp]
        g(n).restype = ctypes.c_long
    g("get_primal").argtypes = [vp, vp, vp, vp, vp]
    g("get_dual").argtypes = [vp, vp, vp, vp]
    g("get_edges").argtypes = [vp, vp]
    g("get_inc").argtypes = [vp, vp]
    g("get_dual_edges").argtypes = [vp, vp]


_ref = None
_compat = None


def ref_available():
    return os.path.exists(REF_POLY)


class FlatPoly:
    """opoly_* (oracle/liboracle.so) or rpoly_* (oracle/_ref/libref_poly.so)"""

    def __init__(self, kind, dim, v2h=CONE_POLAR, c=None):
        global _ref, _compat
        if kind == "oracle":
            import oracle_api
            self.L, self.pre = oracle_api.load(), "opoly_"
            if not getattr(self.L, "_poly_bound", False):
                _bind(self.L, "opoly_")
                self.L._poly_bound = True
        elif kind == "compat":
            # the same driver over the PRODUCT's poly__* symbols (oracle/_ref/libref_poly_hip.so; needs a GPU)
            if _compat is None:
                _compat = ctypes.CDLL(REF_POLY_HIP)
                _bind(_compat, "rpoly_")
            self.L, self.pre = _compat, "rpoly_"
        else:
            if _ref is None:
                _ref = ctypes.CDLL(REF_POLY)
                _bind(_ref, "rpoly_")
            self.L, self.pre = _ref, "rpoly_"
        self.d = dim
        cc = None if c is None else np.ascontiguousarray(c, np.float64)
        self.h = self._f("create")(dim, v2h, None if cc is None else cc.ctypes.data)

    def _f(self, n):
        return getattr(self.L, self.pre + n)

    def dual0_apex(self):
        self._f("dual0_apex")(self.h)

    def add(self, val, ideal=0):
        v = np.ascontiguousarray(val, np.float64)
        return self._f("add")(self.h, v.ctypes.data, int(ideal))

    def init(self):
        return self._f("init")(self.h)

    def next(self):
        v = np.empty(self.d)
        ideal, idx = ctypes.c_int(), ctypes.c_int()
        if self._f("next")(self.h, v.ctypes.data, ctypes.byref(ideal), ctypes.byref(idx)):
            return None
        return v, ideal.value, idx.value

    def mark(self, idx):
        self._f("mark")(self.h, idx)

    def dual_adjacency(self):
        self._f("dual_adjacency")(self.h)

    def dump(self):
        d = self.d
        nv, nf = self._f("nprimal")(self.h), self._f("ndual")(self.h)
        pu, pi, ps = (np.zeros(nv, np.uint8) for _ in range(3))
        X = np.zeros((nv, d))
        self._f("get_primal")(self.h, pu.ctypes.data, pi.ctypes.data, ps.ctypes.data, X.ctypes.data)
        du, di = np.zeros(nf, np.uint8), np.zeros(nf, np.uint8)
        Y = np.zeros((nf, d))
        self._f("get_dual")(self.h, du.ctypes.data, di.ctypes.data, Y.ctypes.data)
        E = np.zeros((self._f("nedges")(self.h), 2), np.int32)
        self._f("get_edges")(self.h, E.ctypes.data)
        I = np.zeros((self._f("ninc")(self.h), 2), np.int32)
        self._f("get_inc")(self.h, I.ctypes.data)
        DE = np.zeros((self._f("ndual_edges")(self.h), 2), np.int32)
        self._f("get_dual_edges")(self.h, DE.ctypes.data)
        return dict(d=d, pu=pu, pi=pi, ps=ps, X=X, du=du, di=di, Y=Y, E=E, I=I, DE=DE)

    def close(self):
        if self.h:
            self._f("free")(self.h)
            self.h = None


def _normalise(X, ideal):
    """poly_chop (|x|<1e-10 -> 0) and poly_normalize_dir (inf-norm 1) semantics, bslv_algs.c:186-279"""
    X = X.copy()
    X[np.abs(X) < 1e-10] = 0.0
    for i in np.nonzero(ideal)[0]:
        mx = np.abs(X[i]).max()
        X[i] = X[i] / mx if mx > 1e-9 else 0.0
    return X


def _order(X, ideal, decimals):
    key = np.round(X, decimals) + 0.0
    cols = [key[:, j] for j in range(X.shape[1] - 1, -1, -1)] + [ideal.astype(np.int64)]
    return np.lexsort(cols)


def canonical(dump, decimals=7, live_dual=None):
    """Drop dead slots, normalise, sort points then directions lexicographically, relabel, and
    return coordinates + index sets (adjacency, incidence) in the new labelling."""
    pu, du = dump["pu"].astype(bool), dump["du"].astype(bool)
    # reference quirk: poly__cut clears dual.used only when it VISITS a facet whose vertex list is
    # already empty (bslv_poly.c:697-705), so a facet that lost its last vertex in that same loop
    # stays flagged with zero vertices.  A facet is compared only if a live vertex lies on it.
    has_vertex = np.zeros(len(du), bool)
    for a, f in dump["I"]:
        if pu[a]:
            has_vertex[f] = True
    du = du & has_vertex
    pid, did = np.nonzero(pu)[0], np.nonzero(du)[0]
    X = _normalise(dump["X"][pid], dump["pi"][pid])
    Y = _normalise(dump["Y"][did], dump["di"][did])
    po, do = _order(X, dump["pi"][pid], decimals), _order(Y, dump["di"][did], decimals)
    pmap = -np.ones(len(pu), np.int64)
    dmap = -np.ones(len(du), np.int64)
    pmap[pid[po]] = np.arange(len(pid))
    dmap[did[do]] = np.arange(len(did))
    E = {tuple(sorted((pmap[a], pmap[b]))) for a, b in dump["E"] if pu[a] and pu[b]}
    I = {(pmap[a], dmap[f]) for a, f in dump["I"] if pu[a] and du[f]}
    DE = {tuple(sorted((dmap[a], dmap[b]))) for a, b in dump["DE"] if du[a] and du[b]}
    return dict(X=X[po], pi=dump["pi"][pid][po], Y=Y[do], di=dump["di"][did][do], E=E, I=I, DE=DE)


def assert_same(a, b, rtol=1e-9, atol=1e-9, dual_edges=True):
    assert a["X"].shape == b["X"].shape, "primal count %s vs %s" % (a["X"].shape, b["X"].shape)
    assert a["Y"].shape == b["Y"].shape, "dual count %s vs %s" % (a["Y"].shape, b["Y"].shape)
    assert np.array_equal(a["pi"], b["pi"])
    assert np.array_equal(a["di"], b["di"])
    np.testing.assert_allclose(a["X"], b["X"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(a["Y"], b["Y"], rtol=rtol, atol=atol)
    assert a["E"] == b["E"], "adjacency sets differ: %d vs %d" % (len(a["E"]), len(b["E"]))
    assert a["I"] == b["I"], "incidence sets differ: %d vs %d" % (len(a["I"]), len(b["I"]))
    if dual_edges:
        assert a["DE"] == b["DE"], "dual adjacency sets differ"


# ---- cut-sequence generators (SURVEY.md section 8d 'poly-only synthetic') ----
def tangent_halfspaces(q, N, seed):
    """N random unit normals: halfspaces d.y >= -1 through cone_polar (points d)"""
    rng = np.random.default_rng(seed)
    D = rng.normal(size=(N, q))
    return D / np.linalg.norm(D, axis=1, keepdims=True)


def run_sequence(P, vals, ideals=None, init_after=None):
    """feed dual vertices; poly__intl_apprx after the first `init_after` (default: all queued first
    like cone_vertenum does, bslv_algs.c:341-350)"""
    vals = np.asarray(vals, np.float64)
    n = len(vals)
    ideals = np.zeros(n, int) if ideals is None else ideals
    init_after = n if init_after is None else init_after
    rcs = []
    for k in range(n):
        if k == init_after:
            assert P.init() == 0
        rcs.append(P.add(vals[k], ideals[k]))
    if init_after >= n:
        assert P.init() == 0
    return rcs


PARITY_MODES = []          # (test id, "exact" | "sliver"): printed at the end of the session (conftest.py)


def _record_mode(mode):
    PARITY_MODES.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], mode))
    return mode


def unmatched_points(a, b, tol):
    """how many points of either result have no partner within tol in the other"""
    from scipy.spatial import cKDTree
    n = 0
    for A, B in ((a["X"], b["X"]), (b["X"], a["X"])):
        dist, _ = cKDTree(B).query(A)
        n += int((dist > tol).sum())
    return n


# End-to-end comparisons (two LP solvers in the loop): 1e-8.  Measured at 1e-9 (BSLV_TEST_TOL=1e-9, round 3): the ex/*.vlp suite
# passes at 1e-9 (tests/test_cli_gpu.py EX_TOL); on the synthetic problems 8 of 11 884 coordinates differ by up to 5e-9 between the
# HIP engine and the CPU oracle -- the weights w of a cut are LP duals after a few hundred pivots on tableaux that are never
# refactorised, on both sides at the solvers' own 1e-9 tolerances.  Index sets stay exact at every tolerance.
DEFAULT_TOL = 1e-8


def assert_benson_results_agree(a, b, c=None, tol=None, allow_sliver=None):
    """Compare two canonicalised Benson results: index sets (adjacency, incidence, dual adjacency) EXACTLY, coordinates within
    `tol`.  Returns "exact".

    allow_sliver = (reason, max_unmatched): a named exception, every use says why and how much.  It additionally accepts the
    tolerance-level difference SURVEY.md 8c describes -- Benson accepts a vertex un-cut when its LP value is <= eps
    (bslv_algs.c:1063) and classifies within +-1e-9 of a cut as 'on' it (bslv_poly.h:47), so two runs that apply cuts in a
    different order can differ in sliver facets of that width -- but only up to `max_unmatched` points (both directions
    summed) without a partner within `tol`, vertex counts within max_unmatched, and each polyhedron must contain the other's
    vertices within `tol`; returns "sliver".  Every call is recorded in PARITY_MODES; the session summary (conftest.py) and
    gpurun_out/parity_modes.json list the comparisons that were not exact."""
    if tol is None:
        tol = float(os.environ.get("BSLV_TEST_TOL", DEFAULT_TOL))
    exact_error = None
    if a["X"].shape == b["X"].shape and a["Y"].shape == b["Y"].shape:
        try:
            assert_same(a, b, rtol=tol, atol=tol)
            return _record_mode("exact")
        except AssertionError as e:
            exact_error = AssertionError("not exact (%d points without a partner within %g; %d / %d vertices, %d / %d facets, %d / %d edges): %s" % (
                unmatched_points(a, b, tol), tol, len(a["X"]), len(b["X"]), len(a["Y"]), len(b["Y"]), len(a["E"]), len(b["E"]), str(e)[:300]))
    else:
        exact_error = AssertionError("counts differ: primal %s vs %s, dual %s vs %s; %d points without a partner within %g" % (
            a["X"].shape, b["X"].shape, a["Y"].shape, b["Y"].shape, unmatched_points(a, b, tol), tol))
    if not allow_sliver:
        raise exact_error
    reason, max_unmatched = allow_sliver
    assert isinstance(reason, str) and len(reason) > 20, "allow_sliver needs a reason"
    tol = max(tol, 1e-6)          # (the allow-lists are counts of points without a partner at 1e-6, the scale of a sliver facet at eps = 1e-7)
    nun = unmatched_points(a, b, tol)
    assert nun <= max_unmatched, "%d points without a partner within %g (allowed: %d) -- %s" % (nun, tol, max_unmatched, exact_error)
    assert abs(len(a["X"]) - len(b["X"])) <= max_unmatched
    q = a["X"].shape[1]
    c = np.ones(q) if c is None else c
    for pts_from, fac_from in ((a, b), (b, a)):
        pts = pts_from["X"][pts_from["pi"] == 0]
        Yp = fac_from["Y"][fac_from["di"] == 0]
        w = np.hstack([Yp[:, :-1], 1 - Yp[:, :-1] @ c[:-1, None]])
        assert (pts @ w.T - Yp[:, -1][None, :]).min() > -tol
    return _record_mode("sliver(%d)" % nun)


def digest_pairs(pairs):
    """SHA-256 + size of a canonical index set (sorted pairs as int64): the form tests/golden/poly_ref_large.json stores"""
    import hashlib
    a = np.array(sorted(pairs), np.int64).reshape(-1, 2)
    return [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(), len(a)]


def assert_matches_large_golden(can, gold_npz, gold_meta, name, rtol=1e-9, atol=1e-9):
    """canonicalised result against a reference golden of SURVEY.md 8c's larger sizes: coordinates within 1e-9, the index sets
    (adjacency, incidence, dual adjacency in the canonical labelling) bit-exact through their SHA-256"""
    g = lambda k: gold_npz[name + "/" + k]
    assert can["X"].shape == g("X").shape and can["Y"].shape == g("Y").shape, (can["X"].shape, g("X").shape, can["Y"].shape, g("Y").shape)
    assert np.array_equal(can["pi"], g("pi")) and np.array_equal(can["di"], g("di"))
    np.testing.assert_allclose(can["X"], g("X"), rtol=rtol, atol=atol)
    np.testing.assert_allclose(can["Y"], g("Y"), rtol=rtol, atol=atol)
    for k in ("E", "I", "DE"):
        assert digest_pairs(can[k]) == list(gold_meta[name][k]), "%s: index set %s differs from the reference's" % (name, k)
