#!/bin/bash
# HBM traffic of k_flush from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (they do not fit one pass on gfx950), with --kernel-trace only, the program itself after `--`.
# usage: collect_pmc.sh <tag> <workload> <batch>   -> gpurun_out/<tag>.json (copy into profiles/)
tag=$1; wl=${2:-S-mid}; B=${3:-256}
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f_$tag gpurun_out/pmc_w_$tag
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f_$tag -o f -- python3 scripts/lp_probe.py $wl $B > gpurun_out/${tag}_probe_f.log 2> gpurun_out/${tag}_f.err || { tail -3 gpurun_out/${tag}_f.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w_$tag -o w -- python3 scripts/lp_probe.py $wl $B > gpurun_out/${tag}_probe_w.log 2> gpurun_out/${tag}_w.err || { tail -3 gpurun_out/${tag}_w.err; exit 1; }
f=$(find gpurun_out/pmc_f_$tag -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/pmc_w_$tag -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py "$f" "$w" gpurun_out/${tag}_probe_f.log "$wl $B" > gpurun_out/$tag.json || exit 1
rm -rf gpurun_out/pmc_f_$tag gpurun_out/pmc_w_$tag
python3 -c "import json; d=json.load(open('gpurun_out/$tag.json')); print(d['workload']); print(d['k_flush'])"
