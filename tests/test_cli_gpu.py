"""GPU: the C command-line driver end to end (.vlp in, .sol files out) against the hybrid
(reference driver + reference polyhedron engine + oracle LP) run on the same file."""
import os
import subprocess
import numpy as np
import pytest

# north_star: the ex/*.vlp suite within 1e-9 relative (index sets exact).  The result files carry 14 digits ("%.14g",
# bslv_main.h:61-63), the goldens come from the reference's own driver: ex01/05/06/08/11 are small rational problems, nothing
# in them is only 1e-7 accurate.
EX_TOL = 1e-9

from bensolve_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bensolve_amd", "csrc", "bensolve_hip")
HYB = os.path.join(ROOT, "oracle", "_ref", "bensolve_hybrid")


def read_img(path):
    a = np.array([[float(x) for x in l.split()] for l in open(path).read().strip().splitlines()])
    t, X = a[:, 0].astype(int), a[:, 1:]
    for i in np.nonzero(t == 0)[0]:
        X[i] /= np.abs(X[i]).max()
    key = np.round(X, 6) + 0.0
    o = np.lexsort([key[:, j] for j in range(X.shape[1] - 1, -1, -1)] + [1 - t])
    return t[o], X[o], o


def read_lists(path):
    return [[int(x) for x in l.split()] for l in open(path).read().split("\n")[:-1]]


@pytest.mark.parametrize("m,n,q,seed,sense", [(30, 15, 3, 5, 1), (25, 12, 3, 8, -1)])
def test_cli_files_match_hybrid(tmp_path, m, n, q, seed, sense):
    prob = synth.covering_vlp(m, n, q, seed)
    if sense == -1:                       # max problem: maximise -P x  ==  minimise P x, outputs sign-flipped
        prob["P"] = -prob["P"]
        prob["optdir"] = -1
    path = os.path.join(tmp_path, "prob.vlp")
    synth.write_vlp(prob, path)
    r = subprocess.run([CLI, path, "-b", "-m", "2", "-B", "32", "-o", os.path.join(tmp_path, "hip")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    tp, Xp, op = read_img(os.path.join(tmp_path, "hip_img_p.sol"))
    td, Yd, od = read_img(os.path.join(tmp_path, "hip_img_d.sol"))
    adj = read_lists(os.path.join(tmp_path, "hip_adj_p.sol"))
    incp = read_lists(os.path.join(tmp_path, "hip_inc_p.sol"))
    incd = read_lists(os.path.join(tmp_path, "hip_inc_d.sol"))
    assert len(adj) == len(tp) and len(incp) == len(td) and len(incd) == len(tp)
    # adjacency symmetric; incidence files are transposes of each other
    for i, row in enumerate(adj):
        for j in row:
            assert i in adj[j]
    assert sorted((f, v) for f, row in enumerate(incp) for v in row) == sorted((f, v) for v, row in enumerate(incd) for f in row)
    if os.path.exists(HYB):
        r2 = subprocess.run([HYB, path, "-b", "-m", "0", "-o", os.path.join(tmp_path, "ref")], capture_output=True, text=True, timeout=300)
        assert r2.returncode == 0
        t2, X2, _ = read_img(os.path.join(tmp_path, "ref_img_p.sol"))
        t3, Y3, _ = read_img(os.path.join(tmp_path, "ref_img_d.sol"))
        assert np.array_equal(tp, t2) and np.array_equal(td, t3)
        np.testing.assert_allclose(Xp, X2, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(Yd, Y3, rtol=1e-6, atol=1e-6)
        # same number of edges / incidences as the reference's files
        radj = read_lists(os.path.join(tmp_path, "ref_adj_p.sol"))
        rinc = read_lists(os.path.join(tmp_path, "ref_inc_p.sol"))
        assert sum(map(len, adj)) == sum(map(len, radj)) and sum(map(len, incp)) == sum(map(len, rinc))


def test_cli_reports_bad_file(tmp_path):
    path = os.path.join(tmp_path, "p.vlp")
    open(path, "w").write("p vlp min 1 1 1 1 1\na 9 1 1\ne\n")
    r = subprocess.run([CLI, path, "-b"], capture_output=True, text=True)
    assert r.returncode == 1 and "line 2" in r.stdout


# ---- all phases (sol_init, phase 0, phase 1, phase 2) natively: the example suite against the committed outputs of the
#      hybrid (the reference's driver + polyhedron code + oracle LP), tests/golden/hybrid.npz ----
import json
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "hybrid.npz"))
EXDIR = os.path.join(ROOT, "tests", "golden", "ex")


def _gold_rows(t, X):
    X = X.copy()
    for i in np.nonzero(t == 0)[0]:
        X[i] /= np.abs(X[i]).max()
    key = np.round(X, 6) + 0.0
    o = np.lexsort([key[:, j] for j in range(X.shape[1] - 1, -1, -1)] + [1 - t])
    return t[o], X[o]


@pytest.mark.parametrize("alg1,alg", [("primal", "primal"), ("primal", "dual"), ("dual", "dual"), ("dual", "primal")])
@pytest.mark.parametrize("ex", ["ex01", "ex05", "ex06", "ex08", "ex11"])
def test_cli_all_phases_match_hybrid_goldens(tmp_path, ex, alg1, alg):
    """ex01: unbounded upper image (phases 0/1 find the recession cone); ex05 / ex08: ordering cone given by generators,
    ex06: by generators of its dual, a max problem; ex11: q = 5.  alg = dual: the dual variant of Benson's algorithm in
    phase 2 (phase2_dual: outer approximation of the lower image by P1(w) LPs that differ in the objective) must arrive at the
    same pair of images; alg1 = dual: phase1_dual finds the recession cone data on the homogeneous problem the same way."""
    base = os.path.join(tmp_path, ex)
    r = subprocess.run([CLI, os.path.join(EXDIR, ex + ".vlp"), "-m", "2", "-B", "64", "-A", alg1, "-a", alg, "-o", base], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for side in ("p", "d"):
        t, X, _ = read_img(base + "_img_%s.sol" % side)
        gt, gX = _gold_rows(GOLD["%s/%s_type" % (ex, side)], GOLD["%s/%s" % (ex, side)])
        assert np.array_equal(t, gt), (ex, side, r.stdout)
        np.testing.assert_allclose(X, gX, rtol=EX_TOL, atol=EX_TOL)


@pytest.mark.parametrize("ex,frag", [("ex02", "VLP is infeasible"), ("ex03", "no vertex"), ("ex04", "totally unbounded")])
def test_cli_documented_outcomes(tmp_path, ex, frag):
    # ex/example02.m, example03.m, example04.m state these outcomes (phase 2 part 1, phase 0, phase 0)
    r = subprocess.run([CLI, os.path.join(EXDIR, ex + ".vlp"), "-m", "1", "-o", os.path.join(tmp_path, ex)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and frag in r.stdout, r.stdout + r.stderr


def test_cli_ex01_known_answer(tmp_path):
    """ex/example01.m: the feasible vertices map to (-6,6), (0,4), (6,6); the upper image has
    the vertices (0,4) and (-6,6) and the extreme directions (1,0) and (-1,1) (SURVEY.md 8c known-answer test)."""
    base = os.path.join(tmp_path, "ex01")
    r = subprocess.run([CLI, os.path.join(EXDIR, "ex01.vlp"), "-m", "0", "-o", base], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    t, X, _ = read_img(base + "_img_p.sol")
    got = sorted((int(a), round(float(x), 9) + 0.0, round(float(y), 9) + 0.0) for a, (x, y) in zip(t, X))
    assert got == sorted([(1, 0.0, 4.0), (1, -6.0, 6.0), (0, 1.0, 0.0), (0, -1.0, 1.0)])
    # <name>.log carries the fields of the reference's log (bslv_main.c:346-397)
    log = open(base + ".log").read()
    assert "problem file:      %s" % os.path.join(EXDIR, "ex01.vlp") in log          # the input file (the reference prints the log's own name: a quirk)
    for frag in ("Problem parameters", "problem rows:            2", "problem columns:         2", "bounded:            no (run phases 0 to 2)",
                 "alg_phase2:         primal", "# primal solution points:           2", "# primal solution directions:       2", "# LPs:"):
        assert frag in log, (frag, log)
    assert open(base + "_c.sol").read().split() == ["1", "1"]


@pytest.mark.parametrize("alg", ["primal", "dual"])
@pytest.mark.parametrize("m,n,q,seed,bounded", [(30, 15, 3, 5, True), (20, 10, 2, 3, False)])
def test_cli_solution_files(tmp_path, m, n, q, seed, bounded, alg):
    """Option -s (opt->solution == PRE_IMG_ON): <name>_pre_img_p.sol holds an x for every element of the upper image (its
    image P x is the vertex), <name>_pre_img_d.sol a dual solution (u, w) for every vertex of the lower image: u >= 0 on the
    cover rows, A'u <= P'w (the columns are x >= 0), b'u = y*_q, w = (y*_1 .. y*_{q-1}, 1 - sum)."""
    prob = synth.covering_vlp(m, n, q, seed)
    path = os.path.join(tmp_path, "prob.vlp")
    synth.write_vlp(prob, path)
    base = os.path.join(tmp_path, "hip")
    # (-a dual: phase2_dual with PRE_IMG_ON, bslv_algs.c:1388-1389, 1484-1497, 1508-1546 -- x comes with every cut of the lower
    # image, (u, w) with every confirmed vertex of it; same files, same meaning)
    r = subprocess.run([CLI, path, "-s", "-m", "1", "-B", "32", "-a", alg, "-o", base] + (["-b"] if bounded else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.array([[float(x) for x in l.split()] for l in open(base + "_img_p.sol").read().strip().splitlines()])
    pre = np.array([[float(x) for x in l.split()] for l in open(base + "_pre_img_p.sol").read().strip().splitlines()])
    assert pre.shape == (len(img), n)
    pts = img[:, 0] == 1
    assert pts.sum() >= 3
    X = pre[pts]
    np.testing.assert_allclose(X @ prob["P"].T, img[pts][:, 1:], rtol=1e-7, atol=1e-7)
    assert np.all(X >= -1e-9) and np.all(X @ prob["A"].T >= 1 - 1e-7)               # feasible points
    dirs = pre[~pts]
    assert np.all(dirs >= -1e-9) and np.all(dirs @ prob["A"].T >= -1e-7)             # recession directions of the feasible set
    imd = np.array([[float(x) for x in l.split()] for l in open(base + "_img_d.sol").read().strip().splitlines()])
    prd = np.array([[float(x) for x in l.split()] for l in open(base + "_pre_img_d.sol").read().strip().splitlines()])
    assert prd.shape == (len(imd), m + q)
    vd = imd[:, 0] == 1
    U, W, Ys = prd[vd][:, :m], prd[vd][:, m:], imd[vd][:, 1:]
    assert np.all(U >= -1e-9)
    np.testing.assert_allclose(W[:, :-1], Ys[:, :-1], rtol=0, atol=1e-8)
    np.testing.assert_allclose(W.sum(axis=1), 1.0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(U.sum(axis=1), Ys[:, -1], rtol=1e-7, atol=1e-7)      # b = (1..1): dual objective = y*_q
    assert np.all(U @ prob["A"] <= W @ prob["P"] + 1e-7)
    assert "solution:           on" in open(base + ".log").read()


def _certified_run(tmp_path, name, eps, timeout):
    """bensolve_hip -s on an example of the reference's suite, then scripts/check_solution.py: every point of the upper image has a
    feasible x whose outcome dominates it up to eps in the ordering cone, every point lies in every supporting halfspace of the
    lower image's vertices and on one of them (geometric duality) -- a certificate that needs no LP solver."""
    import json
    import sys
    vlp = os.path.join(ROOT, "tests", "golden", "ex", name + ".vlp")
    base = os.path.join(tmp_path, name)
    r = subprocess.run([CLI, vlp, "-e", repr(eps), "-s", "-o", base], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_solution.py"), vlp, base, repr(eps)], capture_output=True, text=True, timeout=900)
    cert = json.loads(c.stdout.strip().splitlines()[-1])
    assert c.returncode == 0 and cert["ok"], cert
    return cert


def test_cli_ex07_at_the_recommended_epsilon_is_certified(tmp_path):
    """ex07 of the reference's suite (1211 x 1143, q = 3, ordering cone with 3 generators; ex/example07.m:8-9 recommends -e 0.05):
    all phases on the dense engine, 367 LPs."""
    cert = _certified_run(tmp_path, "ex07", 0.05, 600)
    assert cert["points"] > 100 and cert["dual_vertices"] > 50 and cert["dual_cone_generators"] == 3


@pytest.mark.skipif(not os.environ.get("BSLV_RUN_EX09"), reason="2.3 minutes of GPU time: set BSLV_RUN_EX09=1 (profiles/r02_ex09/ holds the round-2 run)")
def test_cli_ex09_is_certified(tmp_path):
    """ex09 (4608 x 36 939, 185 856 non-zeros, ordering cone with 6 generators; ex/example09.m: -e 1e-2): the tableau of ONE LP is
    1.36 GB; the pool is cut to what fits into the free device memory and the extended selection is on from the first LP."""
    cert = _certified_run(tmp_path, "ex09", 1e-2, 1500)
    assert cert["points"] >= 10 and cert["dual_cone_generators"] == 6


def test_cli_accepts_the_reference_command_line(tmp_path):
    """The options of the reference's command line that change nothing here (-k / -L / -l METHOD, -M, -f, -p, -t; bslv_main.c:112-170)
    are accepted with the reference's own argument check, so that a documented command such as ex/example09.m's
    '-e 1e-2 -m 3 -L primal_simplex -l primal_simplex -p' runs as it stands; the method is recorded in the .log."""
    vlp = os.path.join(ROOT, "tests", "golden", "ex", "ex01.vlp")
    base = os.path.join(tmp_path, "ex01")
    r = subprocess.run([CLI, vlp, "-e", "1e-2", "-m", "3", "-L", "primal_simplex", "-l", "primal_simplex", "-k", "dual_simplex", "-M", "0", "-f", "long", "-p", "-t", "-o", base],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    log = open(base + ".log").read()
    assert "lp_method_phase0:   dual_simplex" in log and "lp_method_phase1:   primal_simplex" in log and "lp_method_phase2:   primal_simplex" in log
    assert "graphics files are not written" in r.stdout
    bad = subprocess.run([CLI, vlp, "-l", "simplex"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "option --lp_method_phase2 (-l): invalid argument" in bad.stdout
    bad = subprocess.run([CLI, vlp, "-k", "auto"], capture_output=True, text=True, timeout=60)          # (phase 0 has no 'auto', bslv_main.c:131-139)
    assert bad.returncode == 1 and "option --lp_method_phase0 (-k): invalid argument" in bad.stdout
