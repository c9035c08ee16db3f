#!/usr/bin/env python3
"""Where does a single cut's time go?  Joins a rocprofv3 kernel trace (results db) of a bench run with the engine's cut log
(BSLV_CUT_LOG: nv ne nminus nzero zero_ub nsurv ncross newlen per cut) and prints, per kernel of the single-cut pipeline,
the duration distribution and its correlation with the sizes of the cut."""
import sqlite3, sys, glob
import numpy as np
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name,start,end from kernels order by start").fetchall()
short = lambda n: n.split('(')[0].replace('bslv::', '').replace('void ', '').split('<')[0]
seq = [(short(n), s, e) for n, s, e in rows]
log = np.loadtxt(sys.argv[2]) if len(sys.argv) > 2 else None
# cuts: k_flags2 -> k_emit2 -> k2_fused
trip = []
i = 0
while i + 2 < len(seq):
    if seq[i][0] == "k_flags2" and seq[i + 1][0] == "k_emit2" and seq[i + 2][0] == "k2_fused":
        trip.append((seq[i], seq[i + 1], seq[i + 2], seq[i + 3] if i + 3 < len(seq) else None))
        i += 3
    else:
        i += 1
print("cuts found in the trace: %d, lines in the cut log: %s" % (len(trip), None if log is None else len(log)))
T = np.array([[(a[2] - a[1]) / 1e3, (b[1] - a[2]) / 1e3, (b[2] - b[1]) / 1e3, (k[1] - b[2]) / 1e3, (k[2] - k[1]) / 1e3, ((n[1] - k[2]) / 1e3 if n else 0)] for a, b, k, n in trip])
names = ["k_flags2", "gap", "k_emit2", "gap", "k2_fused", "gap to next"]
for j, n in enumerate(names):
    v = T[:, j]
    print("%-12s mean %6.2f median %6.2f p10 %6.2f p90 %6.2f p99 %6.2f" % (n, v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90), np.percentile(v, 99)))
print("sum per cut: mean %.2f median %.2f" % (T.sum(axis=1).mean(), np.median(T.sum(axis=1))))
if log is not None and len(log) >= len(trip):
    # the log also holds cuts that went other ways (redundant etc.); align on the applied ones when counts match
    L = log[log[:, 2] > 0] if (log[:, 2] > 0).sum() == len(trip) else log[-len(trip):]
    if len(L) == len(trip):
        cols = ["nv", "ne", "nminus", "nzero", "zero_ub", "nsurv", "ncross", "newlen"]
        for j in (0, 2, 4):
            print(names[j], "correlation with", ", ".join("%s %.2f" % (cols[k], np.corrcoef(T[:, j], L[:, k])[0, 1]) for k in range(8) if L[:, k].std() > 0))
        for j in (0, 2, 4):
            order = np.argsort(T[:, j])
            print(names[j], "slowest 5 cuts:", [(round(T[o, j], 1), dict(zip(cols, L[o].astype(int)))) for o in order[-5:]])
            print(names[j], "fastest 5 cuts:", [(round(T[o, j], 1), dict(zip(cols, L[o].astype(int)))) for o in order[:5]])
# position of a cut in its sequence: time since the last tableau pass (k_flush) ended, and rank within the step
flush_ends = np.array([e for n, s, e in seq if n == "k_flush"])
if len(flush_ends) and len(trip):
    t0 = np.array([a[1] for a, b, k, n in trip])
    idx = np.searchsorted(flush_ends, t0) - 1
    since = np.where(idx >= 0, (t0 - flush_ends[np.maximum(idx, 0)]) / 1e6, -1.0)      # ms
    rank = np.zeros(len(trip), int)
    for i in range(1, len(trip)):
        rank[i] = rank[i - 1] + 1 if idx[i] == idx[i - 1] else 0
    print("duration by rank of the cut within its sequence (after the LP phase):")
    for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 100000)):
        m = (rank >= lo) & (rank < hi)
        if m.any():
            print("  rank %5d..%-6d n %5d  flags %.1f emit %.1f k2 %.1f gap %.1f  (ms since flush %.2f)" % (lo, hi, m.sum(), np.median(T[m, 0]), np.median(T[m, 2]), np.median(T[m, 4]), np.median(T[m, 5]), np.median(since[m])))
    fast = T[:, 2] < 6.0
    print("cuts with k_emit2 < 6 us: %d; their ranks: %s" % (fast.sum(), sorted(rank[fast].tolist())[:40]))
