"""sums the `lp solve_batch:` lines a run printed under BSLV_LP_TIMING=1: solves, pivots and seconds per form, and the longest solves"""
import sys, re, collections
rows = []
for l in open(sys.argv[1], errors="replace"):
    m = re.search(r"(revised|tableau) form (\d+) x (\d+), B (\d+), (\d+) lock-step rounds, (\d+) pivots, (\d+) passes, ([0-9.]+) ms", l)
    if m:
        rows.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(6)), float(m.group(8))))
agg = collections.defaultdict(lambda: [0, 0, 0.0])
for f, M, N, B, p, ms in rows:
    a = agg[(f, M, N)]; a[0] += 1; a[1] += p; a[2] += ms
for k, v in agg.items():
    print(k, "solves", v[0], "pivots", v[1], "seconds %.1f" % (v[2] / 1e3), "ms per pivot %.3f" % (v[2] / max(v[1], 1)))
print("longest:", sorted(rows, key=lambda r: -r[5])[:6])
