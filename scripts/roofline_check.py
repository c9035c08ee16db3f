#!/usr/bin/env python3
"""Cross-check of bench.py's live HIP-event timing of the dominant kernel (k_flush) against a rocprofv3 kernel trace
of the same command: the trace covers every launch of the process (ramp, warm-up, timed steps), bench.py reports the
timed steps only, so the LAST `launches` k_flush dispatches of the trace are the ones to compare.
usage: roofline_check.py <kernel_trace.csv> <bench json line file>"""
import csv, json, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_flush" in r["Kernel_Name"]]
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
b = None
for line in open(sys.argv[2]):
    line = line.strip()
    if line.startswith("{"):
        b = json.loads(line)
n = b["roofline"]["launches"]
out = {"kernel": "bslv::k_flush", "trace_launches_total": int(len(d)), "trace_avg_us_all_launches": round(float(d.mean()), 2),
       "timed_region_launches": n, "trace_avg_us_timed_region": round(float(d[-n:].mean()), 2),
       "bench_hip_event_avg_us": b["roofline"]["avg_launch_us"], "bench_achieved_GBps": b["roofline"]["achieved"],
       "ratio_trace_over_bench": round(float(d[-n:].mean()) / b["roofline"]["avg_launch_us"], 4)}
print(json.dumps(out))
