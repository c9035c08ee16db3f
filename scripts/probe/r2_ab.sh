#!/bin/bash
# A/B of the device-selected rounds (poly_rounds2) on the bench workloads; output in gpurun_out/r2_ab.log
out=gpurun_out/r2_ab.log
: > $out
run() { echo "== $*" >> $out; env $1 BSLV_TIMING=1 timeout -k 10 300 python bench.py --no-cpu-baseline "${@:2}" >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run "BSLV_NO_ROUNDS2=1" --workload S-small --steps 6 --warmup 1
run "BSLV_R2_MIN_CUTS=3" --workload S-small --steps 6 --warmup 1
run "BSLV_R2_MIN_CUTS=0" --workload S-small --steps 6 --warmup 1
run "BSLV_NO_ROUNDS2=1" --workload S-mid --steps 8 --warmup 2
run "BSLV_R2_MIN_CUTS=0" --workload S-mid --steps 8 --warmup 2
run "BSLV_R2_MIN_CUTS=2" --workload S-mid --steps 8 --warmup 2
run "BSLV_R2_MIN_CUTS=3" --workload S-mid --steps 8 --warmup 2
run "BSLV_R2_MIN_CUTS=4" --workload S-mid --steps 8 --warmup 2
run "BSLV_R2_MIN_CUTS=3 BSLV_R2_RULE=1" --workload S-mid --steps 8 --warmup 2
run "BSLV_R2_MIN_CUTS=3" --workload S-mid --steps 8 --warmup 2 --batch 4096
python3 - <<PY > gpurun_out/r2_ab_summary.txt
import json
for l in open("$out"):
    l=l.strip()
    if l.startswith("=="): print(l)
    elif l.startswith("{"):
        d=json.loads(l); print("  ", {k:d[k] for k in ("value","useful_lps_per_sec","lps_redundant_frac","ms_per_step","cuts_applied","pivots_per_lp","phase_ms_per_step","poly_rounds","live_vertices")})
    elif "poly timing" in l or "FAILED" in l or "rror" in l: print("  ", l[:260])
PY
