"""GPU: the C command-line driver end to end (.vlp in, .sol files out) against the hybrid
(reference driver + reference polyhedron engine + oracle LP) run on the same file."""
import os
import subprocess
import numpy as np
import pytest

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


def test_cli_refuses_unbounded_mode_and_bad_file(tmp_path):
    path = os.path.join(tmp_path, "p.vlp")
    synth.write_vlp(synth.covering_vlp(6, 4, 2, 1), path)
    r = subprocess.run([CLI, path], capture_output=True, text=True)
    assert r.returncode == 2 and "not built yet" in r.stdout
    open(path, "w").write("p vlp min 1 1 1 1 1\na 9 1 1\ne\n")
    r = subprocess.run([CLI, path, "-b"], capture_output=True, text=True)
    assert r.returncode == 1 and "line 2" in r.stdout
