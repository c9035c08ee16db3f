"""CPU: the C .vlp reader (kept file-format contract, bslv_vlp.c:275-588) on the reference's own
example files (when /root/reference is present) and on generated files; error behaviour."""
import ctypes
import os
import numpy as np
import pytest

from bensolve_amd import load_library, synth

REF_EX = "/root/reference/ex"


class Vlp(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int), ("n", ctypes.c_int), ("q", ctypes.c_int), ("optdir", ctypes.c_int),
                ("cone_gen", ctypes.c_int), ("n_gen", ctypes.c_int), ("nz", ctypes.c_long), ("nzobj", ctypes.c_long),
                ("A", ctypes.POINTER(ctypes.c_double)), ("P", ctypes.POINTER(ctypes.c_double)),
                ("rtype", ctypes.POINTER(ctypes.c_char)), ("ctype", ctypes.POINTER(ctypes.c_char)),
                ("rlb", ctypes.POINTER(ctypes.c_double)), ("rub", ctypes.POINTER(ctypes.c_double)),
                ("clb", ctypes.POINTER(ctypes.c_double)), ("cub", ctypes.POINTER(ctypes.c_double)),
                ("gen", ctypes.POINTER(ctypes.c_double)), ("c", ctypes.POINTER(ctypes.c_double)),
                ("warnings", ctypes.c_int), ("msg", ctypes.c_char * 256)]


def c_read(path):
    lib = load_library()
    lib.bslv_vlp_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.POINTER(Vlp)), ctypes.POINTER(ctypes.c_int)]
    lib.bslv_vlp_free.argtypes = [ctypes.POINTER(Vlp)]
    p = ctypes.POINTER(Vlp)()
    line = ctypes.c_int()
    rc = lib.bslv_vlp_read(path.encode(), ctypes.byref(p), ctypes.byref(line))
    v = p.contents
    out = dict(rc=rc, line=line.value, msg=v.msg.decode())
    if rc == 0:
        m, n, q = v.m, v.n, v.q
        arr = lambda ptr, k: np.ctypeslib.as_array(ptr, shape=(k,)).copy() if k else np.zeros(0)
        out.update(m=m, n=n, q=q, optdir=v.optdir, cone_gen=v.cone_gen, n_gen=v.n_gen, nz=v.nz, nzobj=v.nzobj,
                   A=arr(v.A, m * n).reshape(m, n), P=arr(v.P, q * n).reshape(q, n),
                   rtype=bytes(v.rtype[:m]).decode(), ctype=bytes(v.ctype[:n]).decode(),
                   rlb=arr(v.rlb, m), rub=arr(v.rub, m), clb=arr(v.clb, n), cub=arr(v.cub, n), c=arr(v.c, q),
                   gen=arr(v.gen, q * v.n_gen).reshape(q, v.n_gen) if v.cone_gen else None)
    lib.bslv_vlp_free(p)
    return out


def py_read(path):
    """independent line-level reading of the same format"""
    d = None
    for line in open(path):
        t = line.split()
        if not t or t[0] == "c":
            continue
        if t[0] == "p":
            m, n, nz, q, nzo = (int(x) for x in t[3:8])
            d = dict(m=m, n=n, q=q, optdir=1 if t[2] == "min" else -1, A=np.zeros((m, n)), P=np.zeros((q, n)),
                     rtype=["f"] * m, ctype=["s"] * n, rlb=np.zeros(m), rub=np.zeros(m), clb=np.zeros(n), cub=np.zeros(n),
                     c=np.zeros(q), cone_gen={8: 0}.get(len(t), 1 if len(t) > 8 and t[8] == "cone" else 2),
                     n_gen=int(t[9]) if len(t) > 8 else 0)
            d["gen"] = np.zeros((q, d["n_gen"])) if len(t) > 8 else None
        elif t[0] == "a":
            d["A"][int(t[1]) - 1, int(t[2]) - 1] = float(t[3])
        elif t[0] == "o":
            d["P"][int(t[1]) - 1, int(t[2]) - 1] = float(t[3])
        elif t[0] == "k":
            if int(t[2]) == 0:
                d["c"][int(t[1]) - 1] = float(t[3])
            else:
                d["gen"][int(t[1]) - 1, int(t[2]) - 1] = float(t[3])
        elif t[0] in "ij":
            types, lb, ub = (d["rtype"], d["rlb"], d["rub"]) if t[0] == "i" else (d["ctype"], d["clb"], d["cub"])
            k = int(t[1]) - 1
            types[k] = t[2]
            rest = [float(x) for x in t[3:]]
            if t[2] in "lds":
                lb[k] = rest.pop(0)
            if t[2] in "ud":
                ub[k] = rest.pop(0)
    return d


@pytest.mark.skipif(not os.path.isdir(REF_EX), reason="reference examples only exist in the build container")
@pytest.mark.parametrize("ex", ["ex01", "ex02", "ex03", "ex04", "ex05", "ex06", "ex07", "ex08", "ex10", "ex11"])
def test_reference_examples_parse(ex):
    path = os.path.join(REF_EX, ex + ".vlp")
    c, p = c_read(path), py_read(path)
    assert c["rc"] == 0, c
    for k in ("m", "n", "q", "optdir", "cone_gen", "n_gen"):
        assert c[k] == p[k], k
    assert np.array_equal(c["A"], p["A"]) and np.array_equal(c["P"], p["P"])
    assert c["rtype"] == "".join(p["rtype"]) and c["ctype"] == "".join(p["ctype"])
    for k in ("rlb", "rub", "clb", "cub", "c"):
        assert np.array_equal(c[k], p[k]), k
    if p["gen"] is not None:
        assert np.array_equal(c["gen"], p["gen"])


def test_generated_file_roundtrips_bit_exact(tmp_path):
    prob = synth.covering_vlp(9, 6, 3, 4)
    prob["rtype"][2] = ord("d"); prob["rlb"][2] = -1.5; prob["rub"][2] = 2.25
    prob["ctype"][1] = ord("f"); prob["ctype"][3] = ord("u"); prob["cub"][3] = 7.0
    path = os.path.join(tmp_path, "t.vlp")
    synth.write_vlp(prob, path)
    c = c_read(path)
    assert c["rc"] == 0
    assert np.array_equal(c["A"], prob["A"]) and np.array_equal(c["P"], prob["P"])
    assert c["rtype"] == "".join(chr(x) for x in prob["rtype"]) and c["ctype"] == "".join(chr(x) for x in prob["ctype"])
    assert c["rlb"][2] == -1.5 and c["rub"][2] == 2.25 and c["cub"][3] == 7.0


@pytest.mark.parametrize("text,frag", [
    ("p vlp min 1 1 1 1 1\na 1 1 1\na 1 1 2\ne\n", "too many constraint"),
    ("p vlp mid 1 1 1 1 1\ne\n", "objective sense"),
    ("p vlp min 1 1 1 1 1\na 2 1 1\ne\n", "row number out of range"),
    ("p vlp min 1 1 1 1 1\ni 1 l 0\ni 1 l 0\ne\n", "duplicate row"),
    ("p vlp min 1 1 1 1 1\nj 1 q 0\ne\n", "column type"),
    ("p vlp min 1 1 1 1 1\nk 1 1 1\ne\n", "invalid designator k"),
    ("p vlp min 1 1 1 1 1\nz 1\ne\n", "line designator"),
    ("p vlp min 1 1 1 1 1\na 1 1 abc\ne\n", "coefficient missing or invalid"),
    ("a 1 1 1\n", "problem line"),
    # headers of untrusted files must not size an allocation before they are bounded
    ("p vlp min 0 2147483647 0 1 0\ne\n", "out of range"),
    ("p vlp min 1 1 1 2000000000 1\ne\n", "more than 16 objectives"),
    ("p vlp min 100000 100000 1 2 1\ne\n", "too large for the dense path"),
    ("p vlp min 1 1 1 2 1 cone 2000000000 1\ne\n", "out of range"),
    ("p vlp min 1 1 1 1 1\na 1 1 1\n", "end of file"),
])
def test_malformed_files_are_rejected(tmp_path, text, frag):
    path = os.path.join(tmp_path, "bad.vlp")
    open(path, "w").write(text)
    c = c_read(path)
    assert c["rc"] == 1 and frag in c["msg"], c


def test_defaults_comments_and_missing_newline(tmp_path):
    path = os.path.join(tmp_path, "d.vlp")
    open(path, "w").write("c a comment\np vlp max 2 2 1 1 1\n\na 1 2 3.5\no 1 1 -2\ni 2 s 4\nj 2 d 0 1\ne")
    c = c_read(path)
    assert c["rc"] == 0 and c["optdir"] == -1
    assert c["rtype"] == "fs" and c["ctype"] == "sd"          # rows default 'f', columns default 's' (fixed at 0)
    assert c["rlb"][1] == 4 and c["A"][0, 1] == 3.5 and c["P"][0, 0] == -2
