#!/bin/bash
# A/B of BSLV_R2_FORK (classification of a round's new vertices on a second stream beside its prunes): the mode / oracle tests with the
# fork on, then the driver's bench command with it off and on (twice each, interleaved), then a kernel trace with it on
out=gpurun_out/fork_ab.log
: > $out
BSLV_R2_FORK=1 timeout -k 10 500 python -m pytest tests/test_poly_modes_gpu.py tests/test_benson_gpu.py tests/test_fill_gpu.py -x -q -m gpu > gpurun_out/fork_tests.log 2>&1 || { tail -40 gpurun_out/fork_tests.log; exit 1; }
tail -4 gpurun_out/fork_tests.log
run() { echo "== $1" >> $out; env $1 timeout -k 10 200 python bench.py --no-cpu-baseline >> $out 2>> gpurun_out/fork_ab.err || echo "FAILED rc=$?" >> $out; }
run BSLV_R2_FORK=0 && run BSLV_R2_FORK=1 && run BSLV_R2_FORK=0 && run BSLV_R2_FORK=1
python3 - <<PY | tee gpurun_out/fork_ab_summary.txt
import json
for l in open("$out"):
    l = l.strip()
    if l.startswith("=="): print(l)
    elif l.startswith("{"):
        d = json.loads(l); print("  ", {k: d.get(k) for k in ("value", "value_min", "value_max", "ms_per_step", "cuts_applied", "poly_rounds", "phase_ms_per_step", "long_window")})
    elif "FAILED" in l: print("  ", l)
PY
bash scripts/probe/prof_bench.sh fork1 BSLV_R2_FORK=1
