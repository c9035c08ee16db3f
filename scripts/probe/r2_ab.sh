#!/bin/bash
out=gpurun_out/r2_ab.log
: > $out
run() { echo "== $*" >> $out; env $1 BSLV_TIMING=1 timeout -k 10 300 python bench.py --no-cpu-baseline "${@:2}" >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run "BSLV_CHUNK_CUTS=512" --workload S-small --steps 6 --warmup 1
run "BSLV_CHUNK_CUTS=1024" --workload S-small --steps 6 --warmup 1
run "BSLV_CHUNK_CUTS=512" --workload S-mid --steps 8 --warmup 2
run "BSLV_CHUNK_CUTS=768" --workload S-mid --steps 8 --warmup 2
run "BSLV_CHUNK_CUTS=384" --workload S-mid --steps 8 --warmup 2
run "BSLV_CHUNK_CUTS=512" --workload S-mid --steps 8 --warmup 2 --batch 3072
run "BSLV_CHUNK_CUTS=512" --workload S-mid --steps 8 --warmup 2 --batch 1536
python3 - <<PY > gpurun_out/r2_ab_summary.txt
import json
for l in open("$out"):
    l=l.strip()
    if l.startswith("=="): print(l)
    elif l.startswith("{"):
        d=json.loads(l); print("  ", {k:d[k] for k in ("value","useful_lps_per_sec","lps_redundant_frac","ms_per_step","cuts_applied","pivots_per_lp","phase_ms_per_step","poly_rounds","live_vertices")})
    elif "FAILED" in l or "rror" in l: print("  ", l[:260])
PY
