#!/bin/bash
# the stop rule of the rounds on S-mid: thin tail rounds through the single-cut pipeline instead (BSLV_R2_MIN_CUTS x BSLV_R2_RULE)
out=gpurun_out/tail_ab.log
: > $out
run() { echo "== $*" >> $out; env $* timeout -k 10 200 python bench.py --no-cpu-baseline >> $out 2>> gpurun_out/tail_ab.err || echo "FAILED rc=$?" >> $out; }
run BSLV_R2_MIN_CUTS=0 && run BSLV_R2_MIN_CUTS=2 BSLV_R2_RULE=1 && run BSLV_R2_MIN_CUTS=3 BSLV_R2_RULE=1 && run BSLV_R2_MIN_CUTS=4 BSLV_R2_RULE=1 && run BSLV_R2_MIN_CUTS=6 BSLV_R2_RULE=1 && run BSLV_R2_MIN_CUTS=0
python3 - <<PY | tee gpurun_out/tail_ab_summary.txt
import json
for l in open("$out"):
    l = l.strip()
    if l.startswith("=="): print(l)
    elif l.startswith("{"):
        d = json.loads(l); print("  ", {k: d.get(k) for k in ("value", "value_min", "value_max", "ms_per_step", "cuts_applied", "poly_rounds", "phase_ms_per_step", "pivots_per_lp")}, d.get("long_window", {}).get("lps_per_sec"), d.get("roofline_cuts"))
    elif "FAILED" in l: print("  ", l)
PY
BSLV_R2_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 --warmup 4 > gpurun_out/r2dbg.json 2> gpurun_out/r2dbg.err; grep -c . gpurun_out/r2dbg.err
