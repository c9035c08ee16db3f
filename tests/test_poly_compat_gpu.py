"""GPU: the reference's OWN driver (bslv_main.c, bslv_algs.c, bslv_vlp.c, bslv_lists.c -- unmodified, compiled where they lie)
WITHOUT bslv_poly.c: the 16 poly__* symbols and the polytope / poly_args structs that bslv_algs.c reads and writes directly
come from the product (include/bslv_poly_compat.h: a host mirror of the HIP polyhedron engine, coherent at every return).
  oracle/_ref/bensolve_ref_hippoly  reference driver + oracle LP (CPU)  + HIP polyhedron engine
  oracle/_ref/bensolve_ref_hip      reference driver + HIP LP engine    + HIP polyhedron engine   (bslv_lp.o, GLPK and
                                    bslv_poly.o replaced by -lbslv_hip, nothing else)
Both run phases 0/1/2 (phase 1 and the ordering cones go through cone_vertenum, which uses the struct fields hardest) of the
example suite and are compared with the committed outputs of the hybrid (same driver + the reference's own bslv_poly.c)."""
import json
import os
import subprocess
import numpy as np
import pytest

# north_star: the ex/*.vlp suite within 1e-9 relative (index sets exact).  The result files carry 14 digits ("%.14g",
# bslv_main.h:61-63), the goldens come from the reference's own driver: ex01/05/06/08/11 are small rational problems, nothing
# in them is only 1e-7 accurate.
EX_TOL = 1e-9

from bensolve_amd import synth
from test_lp_compat_gpu import rows, gold_rows

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXES = {"hippoly": os.path.join(ROOT, "oracle", "_ref", "bensolve_ref_hippoly"), "hip": os.path.join(ROOT, "oracle", "_ref", "bensolve_ref_hip")}
EXDIR = os.path.join(ROOT, "tests", "golden", "ex")
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "hybrid.npz"))
STATUS = json.load(open(os.path.join(ROOT, "tests", "golden", "hybrid_status.json")))

needs_exe = pytest.mark.skipif(not all(os.path.exists(e) for e in EXES.values()),
                               reason="oracle/_ref is built only where /root/reference exists (it travels to the GPU box)")


@needs_exe
@pytest.mark.parametrize("which", sorted(EXES))
@pytest.mark.parametrize("ex", ["ex01", "ex05", "ex06", "ex08", "ex11"])
def test_reference_driver_on_hip_polyhedron_matches_hybrid(tmp_path, ex, which):
    base = os.path.join(tmp_path, ex)
    r = subprocess.run([EXES[which], os.path.join(EXDIR, ex + ".vlp"), "-m", "0", "-o", base], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for side in ("p", "d"):
        t, X = rows(base + "_img_%s.sol" % side)
        gt, gX = gold_rows(GOLD["%s/%s_type" % (ex, side)], GOLD["%s/%s" % (ex, side)])
        assert np.array_equal(t, gt), (ex, side)
        np.testing.assert_allclose(X, gX, rtol=EX_TOL, atol=EX_TOL)
    # the list files written through the mirror's adjacence / incidence lists: one row per element, symmetric adjacency,
    # incidence of one side the transpose of the other
    n_p, n_d = len(open(base + "_img_p.sol").read().strip().splitlines()), len(open(base + "_img_d.sol").read().strip().splitlines())
    lists = {k: [[int(x) for x in l.split()] for l in open(base + "_%s.sol" % k).read().split("\n")[:-1]] for k in ("adj_p", "adj_d", "inc_p", "inc_d")}
    assert len(lists["adj_p"]) == n_p and len(lists["adj_d"]) == n_d
    for k, n in (("adj_p", n_p), ("adj_d", n_d)):
        pairs = {(a, b) for a, row in enumerate(lists[k]) for b in row}
        assert all((b, a) in pairs for a, b in pairs) and all(0 <= b < n for _, b in pairs)
    assert len(lists["inc_p"]) == n_d and len(lists["inc_d"]) == n_p           # row = facet: vertices on it / row = vertex: facets through it
    assert {(v, f) for f, row in enumerate(lists["inc_p"]) for v in row} == {(v, f) for v, row in enumerate(lists["inc_d"]) for f in row}


@needs_exe
@pytest.mark.parametrize("which", sorted(EXES))
@pytest.mark.parametrize("ex,frag", [("ex02", "infeasible"), ("ex03", "no vertex"), ("ex04", "totally unbounded")])
def test_documented_outcomes(tmp_path, ex, frag, which):
    r = subprocess.run([EXES[which], os.path.join(EXDIR, ex + ".vlp"), "-m", "1", "-o", os.path.join(tmp_path, ex)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1
    assert frag in r.stdout, r.stdout
    assert frag in STATUS[ex]["msg"]


@needs_exe
@pytest.mark.parametrize("which", sorted(EXES))
def test_solution_option_through_the_mirror(tmp_path, which):
    """option -s of the reference driver: it writes x into primal.data_primg + dim_primg * idx and hands (u, w) over in
    val_primg_dl (bslv_algs.c:1064-1079) -- pre-images live only in the host mirror; the files must be consistent with the
    images: P x = vertex for the points of the upper image, feasibility, (u, w) a dual solution."""
    m, n, q = 30, 15, 3
    prob = synth.covering_vlp(m, n, q, 5)
    path = os.path.join(tmp_path, "prob.vlp")
    synth.write_vlp(prob, path)
    base = os.path.join(tmp_path, "ref")
    r = subprocess.run([EXES[which], path, "-s", "-b", "-m", "0", "-o", base], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rd = lambda f: np.array([[float(x) for x in l.split()] for l in open(f).read().strip().splitlines()])
    img, pre = rd(base + "_img_p.sol"), rd(base + "_pre_img_p.sol")
    assert pre.shape == (len(img), n)
    pts = img[:, 0] == 1
    np.testing.assert_allclose(pre[pts] @ prob["P"].T, img[pts][:, 1:], rtol=1e-7, atol=1e-7)
    assert np.all(pre[pts] >= -1e-9) and np.all(pre[pts] @ prob["A"].T >= 1 - 1e-7)
    imd, prd = rd(base + "_img_d.sol"), rd(base + "_pre_img_d.sol")
    assert prd.shape == (len(imd), m + q)
    vd = imd[:, 0] == 1
    U, W, Ys = prd[vd][:, :m], prd[vd][:, m:], imd[vd][:, 1:]
    np.testing.assert_allclose(W[:, :-1], Ys[:, :-1], rtol=0, atol=1e-8)
    np.testing.assert_allclose(U.sum(axis=1), Ys[:, -1], rtol=1e-7, atol=1e-7)
    assert np.all(U >= -1e-9) and np.all(U @ prob["A"] <= W @ prob["P"] + 1e-7)


def test_c_abi_exports_the_reference_names():
    """(no reference needed) the symbols bslv_algs.o imports from bslv_poly.c are exported"""
    import ctypes
    from bensolve_amd._lib import load_library
    lib = load_library()
    for name in ("poly__set_default_args", "poly__initialise", "poly__add_vrtx", "poly__intl_apprx", "poly__get_vrtx", "poly__update_adjacence", "poly__swap",
                 "poly__kill", "poly__initialise_permutation", "poly__kill_permutation", "poly__vrtx2file", "poly__primg2file", "poly__adj2file", "poly__inc2file",
                 "poly__plot", "poly__polyck"):
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr), name


import poly_harness as ph

needs_driver = pytest.mark.skipif(not os.path.exists(ph.REF_POLY_HIP), reason="oracle/_ref/libref_poly_hip.so is built only where /root/reference exists (it travels to the GPU box)")


@needs_driver
@pytest.mark.parametrize("fixture", ["poly_ref.npz", "poly_ref_snap.npz"])
def test_cut_sequences_through_the_reference_struct_interface(fixture):
    """The dump driver that made the polyhedron goldens from the unmodified bslv_poly.c (oracle/ref_poly_driver.c: poly__set_default_args,
    poly__initialise, poly__add_vrtx, poly__intl_apprx, poly__update_adjacence, then the polytope structs read field by field) linked
    against the PRODUCT's poly__* symbols instead (oracle/_ref/libref_poly_hip.so): every cut sequence of the goldens gives the return
    codes, coordinates (1e-9) and index sets of the reference.  The snap fixtures (a cut crafted to pass 5e-10 / 5e-11 / 5e-12 above a
    vertex, bslv_poly.c:666-674) at 1e-12: through these symbols the projection sub-band is on."""
    G = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    tol = 1e-12 if "snap" in fixture else 1e-9
    for name in sorted({k.split("/")[0] for k in G.files}):
        q, v2h, apex, init_after = [int(x) for x in G[name + "/in_meta"]]
        P = ph.FlatPoly("compat", q, v2h)
        if apex:
            P.dual0_apex()
        rcs = ph.run_sequence(P, G[name + "/in_vals"], list(G[name + "/in_ideals"]), None if init_after < 0 else init_after)
        P.dual_adjacency()
        can = ph.canonical(P.dump())
        P.close()
        g = lambda k: G[name + "/" + k]
        gold = dict(X=g("X"), pi=g("pi"), Y=g("Y"), di=g("di"), E={tuple(e) for e in g("E")}, I={tuple(e) for e in g("I")}, DE={tuple(e) for e in g("DE")})
        assert list(rcs) == list(g("rc")), name
        try:
            ph.assert_same(can, gold, rtol=0 if "snap" in fixture else tol, atol=tol)
        except AssertionError as e:
            raise AssertionError("%s: %s" % (name, e))
