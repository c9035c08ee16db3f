"""An engine must not depend on what its fresh device memory holds.

Round 3 met a memory access fault in the third engine of a bench.py process.  Its cause, found in round 4:
`k_flags2` of a round of independent cuts that had been queued BEHIND a round that did not go ahead (nothing selected at the end of a
chunk, a capacity that was short) did not return at once like the other kernels of such a round; it walked the edge buffer the round
in front of it would have written -- fresh, unwritten memory -- with the old edge count and used what it found there as element
numbers (`P.cls[E[e].x]`, bslv_poly.c has no counterpart: the reference applies one cut at a time).  On zero pages that reads element
0 and flags nothing; on recycled memory it read wherever the garbage pointed.

The library fills every fresh allocation and every uncopied tail of a re-allocation with the byte BSLV_FILL (default 0, the memory
every other test sees).  With 0x7F an int taken from unwritten memory is 2 139 062 143 -- an index 8.5 GB outside its array, a length
of two billion; with 0xFF it is -1, a class byte reads MINUS, a double NaN.  The runs must end with the same polyhedron bit for bit
(SHA-256 over the canonical dump: vertices, incidence, adjacency, dual adjacency) and the same LP / cut / pivot counts, and must not
report a GPU fault.  Each run is a process of its own: the fill byte is read once per process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "scripts", "probe", "fill_probe.py")


def _run(fill, *args):
    env = dict(os.environ, BSLV_FILL=fill)
    env.pop("BSLV_ALLOC_LOG", None)
    p = subprocess.run([sys.executable, PROBE] + list(args), env=env, capture_output=True, text=True, timeout=600)
    assert "Memory access fault" not in p.stderr, "fill %s: %s" % (fill, p.stderr[-800:])
    assert p.returncode == 0, "fill %s: rc %d\n%s" % (fill, p.returncode, p.stderr[-1500:])
    row = json.loads(p.stdout.strip().splitlines()[-1])
    assert row["fill"] == fill
    return row


def _same(rows):
    ref = rows[0]
    for r in rows[1:]:
        for k in ("sha256", "lps", "cuts", "pivots", "steps", "rounds2", "path", "shapes"):
            assert r[k] == ref[k], "fill %s against fill %s: %s differs: %s vs %s" % (r["fill"], ref["fill"], k, r[k], ref[k])


@pytest.mark.gpu
def test_s_small_to_termination_does_not_depend_on_the_fill_byte():
    """BASELINE configs[1] to termination (the run that faulted in round 3), rounds queued ahead of the host, 16 hot chunks."""
    rows = [_run(f, "S-small") for f in ("0x00", "0x7F", "0xFF")]
    assert rows[0]["rounds2"]["rounds"] > 100 and rows[0]["path"]["hot_chunks"] > 4, rows[0]      # the path in question was taken
    _same(rows)


@pytest.mark.gpu
def test_recycled_memory_of_a_destroyed_engine_does_not_matter_either():
    """bench.py's sequence in one process: an S-mid engine that stays, a second one with the rounds-1-2 rules that is destroyed,
    then S-small to termination -- all three on poisoned memory -- against S-small alone on zeros."""
    rows = [_run("0x00", "S-small"), _run("0x7F", "S-small", "dirty")]
    _same(rows)


@pytest.mark.gpu
def test_small_batches_and_other_dimensions_do_not_depend_on_the_fill_byte():
    """covering problems at q = 4 (40 x 20) with batches of 8 -- chunks too small for the rounds: the one-cut pipeline and the
    multi-cut passes of poly_rounds_host.inc -- and at q = 3 with batches of 64"""
    for name, batch in (("40x20x4x9", "8"), ("30x15x3x5", "64")):
        _same([_run(f, name, batch) for f in ("0x00", "0x7F", "0xFF")])
