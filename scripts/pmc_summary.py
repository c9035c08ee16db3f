#!/usr/bin/env python3
"""HBM traffic of the LP kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, with
--kernel-trace only, as MI355X_MICROARCH.md prescribes).  Units: rocprofv3 reports KB; on gfx950 FETCH_SIZE counts
half of the bytes of wide coalesced reads and is doubled.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <probe log with the pivot counts>"""
import csv, json, re, sys, collections

def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc

F = per_kernel(sys.argv[1], "FETCH_SIZE")
W = per_kernel(sys.argv[2], "WRITE_SIZE")
pivots = passes = 0
dims = None
for line in open(sys.argv[3]):
    md = re.search(r"LP (\d+) x (\d+)", line)
    if md:
        dims = (int(md.group(1)), int(md.group(2)))
    m = re.search(r"pivots \[(\d+)\] passes (\d+)", line)
    if m:
        pivots += int(m.group(1)); passes += int(m.group(2))
    m = re.search(r"passes (\d+) pivots (\d+) total", line)
    if m:
        passes += int(m.group(1)); pivots += int(m.group(2))
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 scripts/lp_probe.py S-mid 256  (two separate passes)",
       "units": "rocprofv3 reports KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE as is",
       "workload": "S-mid P2 LPs 1011x506, batch 256, cold start + three warm-started batches (scripts/lp_probe.py); pass and pivot counts from the probe's own report",
       "kernels": {k: {"fetch_KB_sum": round(F[k][0], 1), "launches": F[k][1], "write_KB_sum": round(W.get(k, [0, 0])[0], 1)} for k in F}}
ku = "bslv::k_flush"
rd = 2.0 * F[ku][0] * 1024 / passes
wr = W[ku][0] * 1024 / passes
alg = 16.0 * (dims[0] + 1) * dims[1] if dims and len(sys.argv) > 4 else 16.0 * (1000 + 5 + 1) * (500 + 2)      # one read + one write of the (M + 1) x N tableau
if len(sys.argv) > 4:
    out["command"] = out["command"].replace("S-mid 256", sys.argv[4])
    out["workload"] = "%s: P2 LPs %d x %d (scripts/lp_probe.py); pass and pivot counts from the probe's own report" % (sys.argv[4], dims[0], dims[1])
out["k_flush"] = {"tableau_passes_in_run": passes, "pivots_in_run": pivots, "pivots_per_pass": round(pivots / passes, 2),
                  "read_bytes_per_pass_corrected": round(rd), "write_bytes_per_pass": round(wr),
                  "traffic_bytes_per_pass": round(rd + wr), "algorithmic_bytes_per_pass": round(alg), "traffic_over_algorithmic": round((rd + wr) / alg, 4),
                  "traffic_bytes_per_pivot": round((rd + wr) * passes / pivots), "per_pivot_algorithmic_of_the_reference_update": round(alg)}
print(json.dumps(out, indent=1))
