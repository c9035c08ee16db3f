"""Execution modes of the single-cut pipeline (speculative launch of round B, hot mode for chunks of cuts, the
one-workgroup adjacency prune and its multi-kernel fallback, the decline-and-rerun path when the device finds a
capacity short) must all build the SAME polyhedron, slot by slot.  The modes are forced through the engine's
test hooks (environment variables read by bslv_poly_create) and compared bit for bit with the CPU oracle
(oracle/poly_dd.c restates bslv_poly.c:104-330) and with each other."""
import os
import ctypes
import numpy as np
import pytest

import poly_harness as ph
from bensolve_amd.poly import PolyEngine
from test_poly_gpu import assert_slotwise_equal

pytestmark = pytest.mark.gpu

HOOKS = ["BSLV_NO_SPEC", "BSLV_NO_HOT", "BSLV_CROSS_UB", "BSLV_K2_LDS", "BSLV_NO_ROUNDS2", "BSLV_R2_MIN_CUTS", "BSLV_CHUNK_CUTS", "BSLV_R2_MIS", "BSLV_R2_SPEC", "BSLV_R2_FUSE", "BSLV_R2_FORK", "BSLV_R2_RULE", "BSLV_R2_MINIT_WG", "BSLV_R2_FC_SLICES"]
MODES = {
    "default": {},
    "no_spec": {"BSLV_NO_SPEC": "1"},
    "no_hot": {"BSLV_NO_HOT": "1"},
    "no_spec_no_hot": {"BSLV_NO_SPEC": "1", "BSLV_NO_HOT": "1"},
    "decline": {"BSLV_CROSS_UB": "3"},                 # most cuts create more than 3 vertices: round B is declined and rerun
    "multi_kernel_prune": {"BSLV_K2_LDS": "64"},       # the bit matrix never fits: every prune takes the fallback
    "multi_kernel_prune_no_spec": {"BSLV_K2_LDS": "64", "BSLV_NO_SPEC": "1"},
}


class hooks:
    def __init__(self, env):
        self.env = env

    def __enter__(self):
        self.old = {k: os.environ.pop(k, None) for k in HOOKS}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k in HOOKS:
            os.environ.pop(k, None)
        for k, v in self.old.items():
            if v is not None:
                os.environ[k] = v


def gpu_run(env, q, vals, k0, batch_mode, batch):
    """first k0 dual vertices one by one (poly__add_vrtx), poly__intl_apprx, the rest through add_cuts in batches"""
    with hooks(env):
        G = PolyEngine(q, 0, None)
        G.set_batch_mode(batch_mode)
        rcs = [G.add(vals[i], 0) for i in range(k0)]
        assert G.init() == 0
        for b0 in range(k0, len(vals), batch):
            rcs += list(G.add_cuts(vals[b0:b0 + batch], None))
        G.dual_adjacency()
        d = G.dump()
        d["paths"] = G.path_stats()
        G.close()
    return rcs, d


def check_paths(name, p, hot_expected):
    """the hook must have driven the run down the path it is meant to cover"""
    assert p["single_cuts"] > 20, (name, p)
    assert (p["hot_chunks"] > 0) == bool(hot_expected), (name, p)
    if "no_spec" in name:
        assert p["speculative"] == 0, (name, p)
    elif name != "decline":            # (a declined cut is rerun and counted without speculation)
        assert p["speculative"] > 20, (name, p)
    if name == "decline":
        assert p["declined"] > 10, (name, p)
    if name.startswith("multi_kernel_prune"):
        assert p["prune_fallbacks"] > 10, (name, p)
    if name == "default":
        assert p["declined"] == 0 and p["prune_fallbacks"] == 0, (name, p)


def clustered_halfspaces(q, N, seed, spread=0.15):
    """unit normals crowded around one direction: neighbouring cuts, every pair in conflict (what a batch of
    newest-first Benson vertices looks like), so the multi-cut path falls back to sequences of single cuts"""
    rng = np.random.default_rng(seed)
    base = np.ones(q) / np.sqrt(q)
    D = base + spread * rng.normal(size=(N, q))
    return D / np.linalg.norm(D, axis=1, keepdims=True)


@pytest.mark.parametrize("q,N,seed", [(3, 300, 21), (4, 160, 22), (5, 120, 23)])
def test_sequential_modes_match_oracle_slotwise(q, N, seed):
    """batch mode 0: cuts in index order, so the oracle's slots and edge order must be reproduced exactly"""
    D = np.vstack([ph.tangent_halfspaces(q, q + 3, seed), clustered_halfspaces(q, N, seed)])
    O = ph.FlatPoly("oracle", q, 0, None)
    rco = ph.run_sequence(O, D, None, q + 3)
    O.dual_adjacency()
    do = O.dump()
    O.close()
    for name, env in MODES.items():
        rcg, dg = gpu_run(env, q, D, q + 3, 0, 64)
        assert list(rco) == list(rcg), name
        assert_slotwise_equal(do, dg)
        check_paths(name, dg["paths"], hot_expected=False)


@pytest.mark.parametrize("q,N,seed", [(4, 400, 31), (5, 300, 32)])
def test_chunk_modes_agree_bitwise(q, N, seed):
    """batch mode 1 on crowded cuts: the conflict pass gives up and runs sequences of single cuts, in hot mode by
    default.  Every hook combination must give the dump of the default mode bit for bit, and the same polyhedron
    as the oracle."""
    D = np.vstack([ph.tangent_halfspaces(q, q + 3, seed), clustered_halfspaces(q, N, seed, 0.05)])
    # (the device-selected rounds inside hot chunks apply the cuts in another order -- same sets, other slot numbers: they
    # have their own test below; here every mode runs the single-cut pipeline)
    ref_rc, ref = gpu_run({"BSLV_NO_ROUNDS2": "1"}, q, D, q + 3, 1, 128)
    for name, env in MODES.items():
        if name == "default":
            continue
        rc, d = gpu_run(dict(env, BSLV_NO_ROUNDS2="1"), q, D, q + 3, 1, 128)
        assert rc == ref_rc, name
        for key in ("pu", "pi", "ps", "X", "du", "di", "Y", "E", "I", "DE"):
            assert np.array_equal(d[key], ref[key]), (name, key)
        check_paths(name, d["paths"], hot_expected="no_hot" not in name)
    check_paths("default", ref["paths"], hot_expected=True)
    O = ph.FlatPoly("oracle", q, 0, None)
    ph.run_sequence(O, D, None, q + 3)
    O.dual_adjacency()
    do = O.dump()
    O.close()
    ph.assert_same(ph.canonical(do), ph.canonical(ref))


R2_MODES = {
    "rounds always": {"BSLV_R2_MIN_CUTS": "-1"},
    "rounds, default stop rule": {},
    "rounds, small chunks": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_CHUNK_CUTS": "64"},
    "rounds, every prune through the multi-kernel path": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_K2_LDS": "64"},
    "rounds without speculation in the tail": {"BSLV_NO_SPEC": "1"},
    "rounds that take a maximal independent set from a conflict matrix": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_MIS": "1"},
    "rounds read by the host one at a time (none queued ahead)": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_SPEC": "0"},
    "prune, classification and pair emission in three launches": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FUSE": "0"},
    "three launches, every prune through the multi-kernel path": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FUSE": "0", "BSLV_K2_LDS": "64"},
    "rounds queued ahead, short capacities (declined rounds)": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_CHUNK_CUTS": "96"},
    "conflict matrix, chunks of 1024": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_MIS": "1", "BSLV_CHUNK_CUTS": "1024"},
    "rounds by the local minima of one random order (round 2's rule, no conflict matrix)": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_MIS": "0"},
    "local minima, chunks of 96 (declined rounds)": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_MIS": "0", "BSLV_CHUNK_CUTS": "96"},
    "prunes and classification in one launch (measured slower, kept as an arm)": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FUSE": "1"},
    "prunes, classification and pair emission in one launch": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FUSE": "2"},
    "stop rule over the last four rounds, tail through the single-cut pipeline": {"BSLV_R2_MIN_CUTS": "3", "BSLV_R2_RULE": "1"},
    "conflict matrix built by 16 workgroups": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_MINIT_WG": "16"},
    "one slice of global counters for prunes with long member lists": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FC_SLICES": "1"},
    "new vertices classified on a second stream beside the prunes": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FORK": "1"},
    "second stream, short capacities (declined rounds, halted queue)": {"BSLV_R2_MIN_CUTS": "-1", "BSLV_R2_FORK": "1", "BSLV_CHUNK_CUTS": "96"},
}


def _r2_cases():
    import itertools
    cases = []
    for q, N, seed in [(3, 600, 41), (4, 300, 42), (5, 250, 43), (6, 70, 44)]:
        cases.append(("tangent q%d" % q, q, np.vstack([ph.tangent_halfspaces(q, q + 3, seed), ph.tangent_halfspaces(q, N, seed + 100)]), q + 3))
    for q, N, seed in [(4, 300, 45), (5, 200, 46)]:
        cases.append(("crowded q%d" % q, q, np.vstack([ph.tangent_halfspaces(q, q + 3, seed), clustered_halfspaces(q, N, seed, 0.05)]), q + 3))
    for q in (4, 5):
        # degenerate: a cube, then many cuts through its vertices and edges (on-plane elements, redundant cuts, ties)
        cube = np.vstack([np.eye(q), -np.eye(q)])
        signs = np.array(list(itertools.product([-1, 1], repeat=q)), float)
        extra = np.vstack([signs / (q - 1), signs / (q - 2), signs / q, cube * 1.5, cube[::-1] * 1.25])
        cases.append(("truncated cube q%d" % q, q, np.vstack([cube, extra]), 2 * q))
    return cases


@pytest.mark.parametrize("case", _r2_cases(), ids=lambda c: c[0])
def test_rounds_of_independent_cuts_build_the_same_polyhedron(case):
    """poly_rounds2: inside a hot chunk the device picks rounds of mutually independent cuts (local minima of a shuffled
    order) and applies each round in the passes of one cut.  Independent cuts commute, so the result must be the polyhedron
    of the sequential definition: vertices, facets, incidence, adjacency and dual adjacency equal the oracle's as sets
    (coordinates 1e-9), and the same cuts are found redundant -- with rounds to the end, with the default stop rule (the tail
    goes through the single-cut pipeline), with small chunks, and with every prune forced through the multi-kernel path."""
    name, q, D, k0 = case
    O = ph.FlatPoly("oracle", q, 0, None)
    rco = ph.run_sequence(O, D, None, k0)
    O.dual_adjacency()
    exp = ph.canonical(O.dump())
    O.close()
    for mode, env in R2_MODES.items():
        with hooks(env):
            G = PolyEngine(q, 0, None)
            rcs = [G.add(D[i], 0) for i in range(k0)]
            assert G.init() == 0
            rcs += list(G.add_cuts(D[k0:], None))
            G.dual_adjacency()
            got = ph.canonical(G.dump())
            st = G.rounds2_stats()
            G.close()
        assert st["cuts"] > 0 and st["rounds"] > 0, (mode, st)
        if "multi-kernel" in mode:
            assert st["fallback_prunes"] > 0, (mode, st)
        if mode == "rounds always":
            assert st["cuts"] + sum(rcs[k0:]) == len(D) - k0, (mode, st)          # every cut applied in a round or found redundant
        # (which cuts come back 'redundant' depends on the order: of two cuts through the same corner the shallower one is
        # redundant only when the deeper one went first; the facets that carry a vertex at the end are the same)
        if name.startswith("tangent"):
            assert sum(rcs) == sum(rco), (mode, sum(rcs), sum(rco))
        ph.assert_same(got, exp)


def _smid_cut_sequence(steps):
    """the dual vertices (cuts, in the order they were applied) of a batched Benson run on the bench workload"""
    from bensolve_amd import synth
    from bensolve_amd.benson import BensonEngine
    prob = synth.CONFIGS["S-mid"]()
    eng = BensonEngine(prob, eps=1e-7, pool_slots=2 * 1024 + 64)
    assert eng.start() == 0
    for _ in range(steps):
        nl, nt = eng.collect(1024, 0, 1)
        rec, piv, ls = eng.solve_local(nl)
        eng.apply(rec)
    D = eng.poly_dump()
    health = eng.poly_call("rounds2_health")
    eng.close()
    assert health["late_left"] == 0, health         # (every chunk's rounds ended with nothing alive; torn reads are repeated, not hidden)
    return prob["q"], np.ones(prob["q"]), D["Y"], D


def test_smid_cut_sequence_matches_oracle_slotwise():
    """~2000 real cuts of the bench workload (extreme directions with long incidence lists, degenerate points, facets of
    10^2 elements), replayed one at a time through the engine and through oracle/poly_dd.c: slots, edges and incidence
    lists bit for bit.  The replay settles every prune with the multi-kernel path in a second engine (all fallbacks)."""
    q, c, Y, _ = _smid_cut_sequence(6)
    O = ph.FlatPoly("oracle", q, 1, c)
    for k in range(1, q + 1):
        O.add(Y[k], 0)
    assert O.init() == 0
    for y in Y[q + 1:]:
        O.add(y, 0)
    do = O.dump()
    O.close()
    for force_multi in (False, True, "member lists"):
        G = PolyEngine(q, 1, c)
        G.set_batch_mode(0)
        if force_multi:
            G.debug_set(0, 64)
        if force_multi == "member lists":       # ... and every prune confirms its edges through the facet-major member lists
            G.debug_set(4, 2)
        for k in range(1, q + 1):
            G.add(Y[k], 0)
        assert G.init() == 0
        rest = Y[q + 1:]
        for b0 in range(0, len(rest), 256):
            G.add_cuts(rest[b0:b0 + 256], None)
        dg = G.dump()
        paths = G.path_stats()
        G.close()
        assert len(rest) > 1500 and paths["single_cuts"] > 1500
        assert (paths["prune_fallbacks"] > 1500) == bool(force_multi)
        assert (paths["member_list_prunes"] > 1500) == (force_multi == "member lists")
        for key in ("pu", "pi", "du", "di", "E", "I"):
            assert np.array_equal(do[key], dg[key]), (force_multi, key)
        live = do["pu"].astype(bool)
        np.testing.assert_allclose(do["X"][live], dg["X"][live], rtol=1e-12, atol=1e-12)


def test_smid_benson_steps_identical_in_all_modes():
    """the batched driver on the bench workload: default mode, everything conservative, every prune through the fallback
    (which exercises the abort / decline / rerun machinery on every cut) -- the same polyhedron bit for bit"""
    dumps = {}
    for name, env in (("default", {"BSLV_NO_ROUNDS2": "1"}), ("conservative", {"BSLV_NO_SPEC": "1", "BSLV_NO_HOT": "1"}),
                      ("fallback_prune", {"BSLV_K2_LDS": "64", "BSLV_NO_ROUNDS2": "1"}), ("rounds", {"BSLV_R2_MIN_CUTS": "-1"})):
        with hooks(env):
            dumps[name] = _smid_cut_sequence(5)[3]
    for name in ("conservative", "fallback_prune"):
        for key in ("pu", "pi", "ps", "X", "du", "di", "Y", "E", "I"):
            assert np.array_equal(dumps[name][key], dumps["default"][key]), (name, key)
    # the device-selected rounds apply the cuts of a batch in another order, which changes the vertices the next batch starts
    # from: a different (equally valid) run of the algorithm.  It must be a consistent polyhedron: every vertex satisfies every
    # cut, lies on at least q facets and on exactly those its incidence list names
    d = dumps["rounds"]
    q = d["d"]
    live = d["pu"].astype(bool) & (d["pi"] == 0)
    Y = d["Y"][d["du"].astype(bool) & (d["di"] == 0)]
    w = np.hstack([Y[:, :-1], 1 - Y[:, :-1].sum(axis=1, keepdims=True)])
    assert (d["X"][live] @ w.T - Y[:, -1][None, :]).min() > -1e-7
    cnt = np.bincount(d["I"][:, 0], minlength=len(live))
    assert cnt[live].min() >= q
    Yall = d["Y"]; wall = np.hstack([Yall[:, :-1], 1 - Yall[:, :-1].sum(axis=1, keepdims=True)])
    I = d["I"][live[d["I"][:, 0]] & (d["di"][d["I"][:, 1]] == 0)]
    assert np.abs(np.einsum("ij,ij->i", d["X"][I[:, 0]], wall[I[:, 1]]) - Yall[I[:, 1], -1]).max() < 1e-6


def _host_edge_test_sample(d, npairs, seed):
    """Independent recomputation of the adjacency prune on the host for sampled pairs of the LARGEST facet of a dump: two
    vertices are adjacent iff they share at least dim-1 facets and no third live vertex lies on all of them (edge_test,
    bslv_poly.c:467-512).  Returns (pairs checked, mismatches, adjacent pairs among them)."""
    dim = d["d"]
    live = d["pu"].astype(bool)
    I = d["I"][live[d["I"][:, 0]]]
    order = np.argsort(I[:, 1], kind="stable")
    fac, verts = I[order, 1], I[order, 0]
    fstart = np.searchsorted(fac, np.arange(fac.max() + 2))
    members = lambda f: verts[fstart[f]:fstart[f + 1]]            # ascending vertex ids (I is vertex-major, stable sort)
    order2 = np.argsort(I[:, 0], kind="stable")
    vv, ff = I[order2, 0], I[order2, 1]
    vstart = np.searchsorted(vv, np.arange(len(d["pu"]) + 1))
    inc = lambda v: ff[vstart[v]:vstart[v + 1]]
    sizes = np.diff(fstart)
    big = int(np.argmax(sizes))
    mem = members(big)
    E = d["E"]
    ekeys = set((np.minimum(E[:, 0], E[:, 1]).astype(np.int64) << 32 | np.maximum(E[:, 0], E[:, 1]).astype(np.int64)).tolist())
    rng = np.random.default_rng(seed)
    # half of the sample: random pairs of the facet (mostly non-adjacent); the other half: edges the engine reports inside it
    inbig = np.zeros(len(d["pu"]), bool); inbig[mem] = True
    Ein = E[inbig[E[:, 0]] & inbig[E[:, 1]]]
    pairs = [tuple(sorted(map(int, rng.choice(mem, 2, replace=False)))) for _ in range(npairs // 2)]
    pairs += [tuple(sorted(map(int, Ein[k]))) for k in rng.choice(len(Ein), min(npairs // 2, len(Ein)), replace=False)]
    bad = nadj = 0
    for u, v in pairs:
        M = np.intersect1d(inc(u), inc(v), assume_unique=True)
        adj = False
        if dim == 1 or len(M) >= dim - 1:
            Ms = sorted(M.tolist(), key=lambda f: sizes[f])
            cand = members(Ms[0])
            for f in Ms[1:]:
                if len(cand) <= 2:
                    break
                cand = np.intersect1d(cand, members(f), assume_unique=True)
            adj = len(cand) == 2                                  # u and v themselves
        nadj += adj
        bad += adj != (((u << 32) | v) in ekeys)
    return len(pairs), bad, nadj, int(sizes[big])


def test_large_facets_member_list_prune_equals_full_scan():
    """S-degenerate at full size, first five steps: new facets of 10^4-10^5 elements go through the multi-kernel prune with the
    row-tiled pair kernel, which confirms edges against the members of the smallest mutual facet.  Three arms: (1) that
    default, (2) edges confirmed against ALL elements of the facet, (3) the one-dimensional, untiled pair kernel
    (k_pair_flags_bits; only the first four steps: the fifth brings a facet of 114 296 elements = 2.5e7 pair blocks, more than
    a one-dimensional grid of 256-thread workgroups can hold -- the runtime wraps such a grid silently, which is why the
    large-facet kernel uses a two-dimensional one and the one-dimensional launch refuses).  Same polyhedron bit for bit in all
    arms; and, independent of all three kernels, the adjacency of sampled pairs of the largest facet is recomputed on the
    host from the incidence lists."""
    import hashlib
    from bensolve_amd import synth
    from bensolve_amd.benson import BensonEngine
    prob = synth.CONFIGS["S-degenerate"]()
    out = {}

    def digest(d):
        h = hashlib.sha256()
        for key in ("pu", "pi", "du", "di", "X", "Y", "E", "I"):
            h.update(np.ascontiguousarray(d[key]).tobytes())
        return h.hexdigest(), int(d["pu"].sum()), len(d["E"])

    for name, lists, steps in (("member lists", 1, 5), ("full scan", 0, 5), ("untiled", 1, 4), ("no flag array", 1, 5)):
        eng = BensonEngine(prob, eps=1e-7, pool_slots=4 * 64 + 64)
        eng.poly_call("debug_set", 5, lists)
        if name == "untiled":
            eng.poly_call("debug_set", 4, 1 << 30)            # no facet is 'large': one-dimensional k_pair_flags_bits for every fallback prune
        if name == "no flag array":
            # (round 4) the pairs are written by testing the listed pair blocks AGAIN instead of from a flag byte per pair: what facets of
            # several 10^5 elements need (163 GB of flags at 571 084 elements); forced here for every large facet
            eng.poly_call("debug_set", 16, 1)
        assert eng.start() == 0
        d4 = None
        for it in range(steps):
            nl, nt = eng.collect(64, 0, 1)
            rec, piv, ls = eng.solve_local(nl)
            assert np.all(rec[:, 1] == 4)
            eng.apply(rec)
            if it == 3:
                d4 = digest(eng.poly_dump())
        d = eng.poly_dump()
        paths = eng.poly_call("path_stats")
        eng.lib.bslv_poly_noflag_prunes.restype = ctypes.c_long
        eng.lib.bslv_poly_noflag_prunes.argtypes = [ctypes.c_void_p]
        paths["noflag"] = int(eng.lib.bslv_poly_noflag_prunes(eng._poly_h))
        eng.close()
        out[name] = (digest(d), d4, paths)
        if name == "member lists":
            n, bad, nadj, fsize = _host_edge_test_sample(d, 1200, 7)
            assert fsize > 50000 and n >= 1000 and nadj >= 300, (n, nadj, fsize)
            assert bad == 0, "%d of %d sampled pairs of the largest facet (%d elements) disagree with the host edge test" % (bad, n, fsize)
    assert out["member lists"][2]["member_list_prunes"] > 20 and out["full scan"][2]["member_list_prunes"] == 0 and out["untiled"][2]["member_list_prunes"] == 0, out
    assert out["member lists"][0][1] > 500000       # (live elements after five steps: 596 861 with the batches taken newest first, round 3)
    assert out["member lists"][0] == out["full scan"][0] == out["no flag array"][0], out
    assert out["no flag array"][2]["noflag"] > 20 and out["member lists"][2]["noflag"] == 0, out
    assert out["member lists"][1] == out["full scan"][1] == out["untiled"][1], out       # after four steps, all three kernels
