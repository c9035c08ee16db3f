"""the figures of a `bench.py --workload ex09-lp` line"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "pivots_per_lp", "passes", "lockstep_rounds", "lps_not_optimal")})
print("k_flush frac", d["roofline"]["frac"], "avg launch us", d["roofline"]["avg_launch_us"])
