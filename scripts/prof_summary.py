#!/usr/bin/env python3
"""Per-kernel duration statistics and inter-kernel gaps from a rocprofv3 results database (kernel trace)."""
import sqlite3, sys, collections
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end from kernels order by start").fetchall()
short = lambda n: n.split('(')[0].replace('bslv::', '').replace('void ', '')
seq = [(short(n), s, e) for n, s, e in rows]
d = collections.defaultdict(list)
for n, s, e in seq: d[n].append((e - s) / 1e3)
print("kernel                  calls   mean   median    p90    p99   total_ms")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    v = np.array(v); print(k.ljust(22), "%6d %7.1f %7.1f %7.1f %7.1f %9.1f" % (len(v), v.mean(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.sum() / 1e3))
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(seq[:-1], seq[1:]): gaps.setdefault((n0, n1), []).append((s1 - e0) / 1e3)
print("gaps (us) between consecutive kernels:")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    v = np.array(v); print(" ", k, len(v), 'median %.1f mean %.1f p90 %.1f' % (np.median(v), v.mean(), np.percentile(v, 90)))
