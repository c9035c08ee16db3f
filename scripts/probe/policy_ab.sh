#!/bin/bash
# A/B of the batch selection policies on S-mid (bench.py without the CPU leg); output in gpurun_out/policy_ab.log
out=gpurun_out/policy_ab.log
: > $out
run() { echo "== $*" >> $out; BSLV_TIMING=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run --policy 1
run --policy 3 --sib-cap 1 --sib-window 8 --pool 16384
run --policy 3 --sib-cap 2 --sib-window 8 --pool 16384
run --policy 3 --sib-cap 1 --sib-window 32 --pool 32768
run --policy 3 --sib-cap 1 --sib-window 8 --pool 16384 --batch 4096
