"""Deterministic synthetic VLP generators.

PRNG: splitmix64 seeded with the stated seed, u01 = (x >> 11) * 2**-53, values drawn in
row-major order A, then P (then b where used).  The same stream is reproduced in C by
bensolve_amd/csrc/host (bslv_synth.c) so files and in-memory data are bit-identical.

  S-small      q=3  n=100  m=200   seed 1   (BASELINE.json configs[1])
  S-mid        q=5  n=500  m=1000  seed 2   (configs[2], configs[3])
  S-degenerate q=10 n=2000 m=4000  seed 3   (configs[4])
"""
import numpy as np

MASK = (1 << 64) - 1


def splitmix64_stream(seed, count):
    """Return `count` u01 doubles from splitmix64(seed) (vectorised)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def covering_vlp(m, n, q, seed):
    """Dense covering-type MOLP: min Px s.t. Ax >= 1, x >= 0, A,P ~ U[0,1).

    Bounded in bensolve's sense (recession cone of the upper image is R^q_+), so phase 2 only
    ('-b').  Returns dict(A, P, row bounds, col bounds)."""
    u = splitmix64_stream(seed, m * n + q * n)
    A = u[: m * n].reshape(m, n)
    P = u[m * n:].reshape(q, n)
    return dict(m=m, n=n, q=q, A=A, P=P, optdir=1,
                rtype=np.full(m, ord("l"), np.uint8), rlb=np.ones(m), rub=np.zeros(m),
                ctype=np.full(n, ord("l"), np.uint8), clb=np.zeros(n), cub=np.zeros(n))


def degenerate_vlp(m, n, q, seed):
    """Hypercube + integer rows, integer lattice objectives (SURVEY 8d S-degenerate).

    Recipe: A = [I_n ; G], rows 1..n 'd 0 1', G (m-n x n) entries in {0,1,2}, rows 'l 1';
    P[k][j] in {-2..2}; columns free.  Recorded deviation: the m-n rows of G are 'l 1'
    cover rows as in the recipe, and the first n rows box x in [0,1]^n, so the upper image is bounded
    below in every objective (P x is bounded on the cube) and phase 2 alone suffices."""
    assert m > n
    g = m - n
    u = splitmix64_stream(seed, g * n + q * n)
    G = np.floor(u[: g * n] * 3.0).reshape(g, n)
    P = (np.floor(u[g * n:] * 5.0) - 2.0).reshape(q, n)
    A = np.vstack([np.eye(n), G])
    rtype = np.concatenate([np.full(n, ord("d"), np.uint8), np.full(g, ord("l"), np.uint8)])
    rlb = np.concatenate([np.zeros(n), np.ones(g)])
    rub = np.concatenate([np.ones(n), np.zeros(g)])
    return dict(m=m, n=n, q=q, A=A, P=P, optdir=1, rtype=rtype, rlb=rlb, rub=rub,
                ctype=np.full(n, ord("f"), np.uint8), clb=np.zeros(n), cub=np.zeros(n))


def fold_singleton_rows(prob):
    """Rows of A with a single non-zero become bounds of that column (what the batched driver's presolve does,
    bensolve_amd/csrc/benson_driver.hip bslv_benson_create): same VLP, smaller LP, boxed columns."""
    A = prob["A"]
    n = prob["n"]
    lo_of = {ord("f"): lambda l, u: (-np.inf, np.inf), ord("l"): lambda l, u: (l, np.inf), ord("u"): lambda l, u: (-np.inf, u),
             ord("d"): lambda l, u: (l, u), ord("s"): lambda l, u: (l, l)}
    clo = np.empty(n); cup = np.empty(n)
    for j in range(n):
        clo[j], cup[j] = lo_of[int(prob["ctype"][j])](prob["clb"][j], prob["cub"][j])
    keep = []
    for i in range(prob["m"]):
        nz = np.flatnonzero(A[i])
        if len(nz) == 1:
            j = int(nz[0]); a = A[i, j]
            lo, up = lo_of[int(prob["rtype"][i])](prob["rlb"][i], prob["rub"][i])
            lo, up = (lo / a, up / a) if a > 0 else (up / a, lo / a)
            lo, up = max(lo, clo[j]), min(up, cup[j])
            if lo <= up:
                clo[j], cup[j] = lo, up
                continue
        keep.append(i)
    ctype = np.empty(n, np.uint8)
    for j in range(n):
        fl, fu = np.isfinite(clo[j]), np.isfinite(cup[j])
        ctype[j] = ord("s") if fl and fu and clo[j] == cup[j] else ord("d") if fl and fu else ord("l") if fl else ord("u") if fu else ord("f")
    out = dict(prob)
    out.update(m=len(keep), A=np.ascontiguousarray(A[keep]), rtype=prob["rtype"][keep].copy(), rlb=prob["rlb"][keep].copy(),
               rub=prob["rub"][keep].copy(), ctype=ctype, clb=np.where(np.isfinite(clo), clo, 0.0), cub=np.where(np.isfinite(cup), cup, 0.0))
    return out


CONFIGS = {
    "S-small": lambda: covering_vlp(200, 100, 3, 1),
    "S-mid": lambda: covering_vlp(1000, 500, 5, 2),
    "S-degenerate": lambda: degenerate_vlp(4000, 2000, 10, 3),
    # the same LPs (4000 x 2000 before the presolve) with q reduced until the run TERMINATES on one GPU ("reduce q and
    # record the change"): q = 4 ends after 1.7e5 LPs / 3.3e4 facets; q = 5 at n = 200 after 4.2e6 LPs / 7.2e5 facets
    "S-degenerate-q4": lambda: degenerate_vlp(4000, 2000, 4, 3),
    "S-degenerate-q5-n200": lambda: degenerate_vlp(400, 200, 5, 3),
    # covering problems of the S-mid recipe that TERMINATE on one GPU in seconds to minutes: the whole-run yardstick of a batch
    # selection rule (S-mid itself never terminates: its steady state is an ever-growing frontier)
    "C-q4-300": lambda: covering_vlp(300, 150, 4, 11),
    "C-q4-400": lambda: covering_vlp(400, 200, 4, 9),
    "C-q5-120": lambda: covering_vlp(120, 60, 5, 7),
}


def write_vlp(prob, path):
    """Write `prob` in the reference's .vlp format (bslv_vlp.c:275-588, ex/prob2vlp.m:105-182)
    with %.17g numbers so that file and in-memory data are bit-identical."""
    A, P = prob["A"], prob["P"]
    m, n, q = prob["m"], prob["n"], prob["q"]
    ai, aj = np.nonzero(A)
    pi, pj = np.nonzero(P)
    with open(path, "w") as f:
        f.write("p vlp %s %d %d %d %d %d\n" % ("min" if prob["optdir"] == 1 else "max", m, n, len(ai), q, len(pi)))
        f.write("".join("a %d %d %.17g\n" % (i + 1, j + 1, A[i, j]) for i, j in zip(ai, aj)))
        f.write("".join("o %d %d %.17g\n" % (i + 1, j + 1, P[i, j]) for i, j in zip(pi, pj)))
        for kind, types, lb, ub, cnt in (("i", prob["rtype"], prob["rlb"], prob["rub"], m),
                                         ("j", prob["ctype"], prob["clb"], prob["cub"], n)):
            for k in range(cnt):
                t = chr(types[k])
                if t == "f":
                    if kind == "j":
                        f.write("%s %d f\n" % (kind, k + 1))
                elif t == "l":
                    f.write("%s %d l %.17g\n" % (kind, k + 1, lb[k]))
                elif t == "u":
                    f.write("%s %d u %.17g\n" % (kind, k + 1, ub[k]))
                elif t == "d":
                    f.write("%s %d d %.17g %.17g\n" % (kind, k + 1, lb[k], ub[k]))
                elif t == "s":
                    f.write("%s %d s %.17g\n" % (kind, k + 1, lb[k]))
        f.write("e\n")


def read_vlp(path):
    """Line-level reader of the .vlp format into the problem dict the engines take; the C reader
    (bslv_vlp_read) is the product's, this one serves scripts and tests.  Defaults as the reference: rows 'f', columns 's'."""
    d = None
    for line in open(path):
        t = line.split()
        if not t or t[0] == "c":
            continue
        if t[0] == "p":
            m, n, nz, q, nzo = (int(x) for x in t[3:8])
            d = dict(m=m, n=n, q=q, optdir=1 if t[2] == "min" else -1, A=np.zeros((m, n)), P=np.zeros((q, n)),
                     rtype=np.full(m, ord("f"), np.uint8), ctype=np.full(n, ord("s"), np.uint8), rlb=np.zeros(m), rub=np.zeros(m),
                     clb=np.zeros(n), cub=np.zeros(n), c=np.zeros(q), cone_kind=0 if len(t) <= 8 else (1 if t[8] == "cone" else 2),
                     gen=np.zeros((q, int(t[9]))) if len(t) > 8 else None)
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
            ty, lb, ub = (d["rtype"], d["rlb"], d["rub"]) if t[0] == "i" else (d["ctype"], d["clb"], d["cub"])
            k = int(t[1]) - 1
            ty[k] = ord(t[2])
            rest = [float(x) for x in t[3:]]
            if t[2] in "lds":
                lb[k] = rest.pop(0)
            if t[2] in "ud":
                ub[k] = rest.pop(0)
    return d
