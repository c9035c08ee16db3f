"""CPU: deterministic synthetic generators and the .vlp writer (kept file-format contract)."""
import os
import numpy as np

from bensolve_amd import synth


def test_splitmix64_known_values():
    # splitmix64(seed=0): first outputs 0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4 (public reference values)
    u = synth.splitmix64_stream(0, 2)
    exp = np.array([0xE220A8397B1DCDAF >> 11, 0x6E789E6AA1B965F4 >> 11], dtype=np.float64) * 2.0 ** -53
    assert np.array_equal(u, exp)


def test_configs_shapes_and_determinism():
    p = synth.CONFIGS["S-small"]()
    assert (p["m"], p["n"], p["q"]) == (200, 100, 3) and p["A"].shape == (200, 100) and p["P"].shape == (3, 100)
    p2 = synth.covering_vlp(200, 100, 3, 1)
    assert np.array_equal(p["A"], p2["A"]) and np.array_equal(p["P"], p2["P"])
    assert 0 <= p["A"].min() and p["A"].max() < 1


def test_vlp_roundtrip(tmp_path):
    prob = synth.covering_vlp(7, 5, 2, 3)
    path = os.path.join(tmp_path, "t.vlp")
    synth.write_vlp(prob, path)
    lines = open(path).read().splitlines()
    assert lines[0] == "p vlp min 7 5 35 2 10" and lines[-1] == "e"
    A = np.zeros((7, 5)); P = np.zeros((2, 5))
    for l in lines:
        t = l.split()
        if t[0] == "a": A[int(t[1]) - 1, int(t[2]) - 1] = float(t[3])
        if t[0] == "o": P[int(t[1]) - 1, int(t[2]) - 1] = float(t[3])
    assert np.array_equal(A, prob["A"]) and np.array_equal(P, prob["P"])      # %.17g round-trips bit-exactly
    assert sum(1 for l in lines if l.startswith("i ")) == 7 and sum(1 for l in lines if l.startswith("j ")) == 5


def test_read_vlp_helper_matches_the_c_reader():
    """bensolve_amd.synth.read_vlp (scripts, tests) against the product's C reader on the committed example files"""
    import ctypes
    import os
    import numpy as np
    from bensolve_amd import synth
    import test_vlp_parser as tp
    exdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ex")
    for ex in ("ex01", "ex05", "ex06", "ex10"):
        path = os.path.join(exdir, ex + ".vlp")
        a, b = synth.read_vlp(path), tp.c_read(path)
        assert (a["m"], a["n"], a["q"], a["optdir"]) == (b["m"], b["n"], b["q"], b["optdir"])
        np.testing.assert_array_equal(a["A"], b["A"])
        np.testing.assert_array_equal(a["P"], b["P"])
        assert bytes(a["rtype"]).decode() == "".join(b["rtype"]) if isinstance(b["rtype"], list) else True
        for k in ("rlb", "rub", "clb", "cub"):
            np.testing.assert_array_equal(a[k], b[k])
        if a["gen"] is not None:
            np.testing.assert_array_equal(a["gen"], b["gen"])
