#!/bin/bash
# Collects the records of a round on the GPU box: the default bench line, the rocprofv3 --kernel-trace --stats summary of the
# SAME command (per-kernel average durations) and the cross-check of bench.py's HIP-event timing of k_flush against the trace.
# usage: bash scripts/collect_profiles.sh r02   -> gpurun_out/r02_*.{json,csv}; copy what is to be judged into profiles/
tag=${1:-rXX}
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${tag}_bench_smid.json 2> gpurun_out/${tag}_bench_smid.err || exit 1
python3 bench.py --workload S-small --no-pair > gpurun_out/${tag}_bench_ssmall.json 2>> gpurun_out/${tag}_bench_smid.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o ${tag} -- python3 bench.py --no-cpu-baseline --no-long-window > gpurun_out/${tag}_bench_smid_profiled.json 2> gpurun_out/${tag}_prof.err || exit 1
ks=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1)
kt=$(find gpurun_out/prof_${tag} -name "*kernel_trace.csv" | head -1)
cp "$ks" gpurun_out/${tag}_bench_smid_kernel_stats.csv
python3 scripts/roofline_check.py "$kt" gpurun_out/${tag}_bench_smid_profiled.json > gpurun_out/${tag}_k_flush_trace_vs_events.json
rm -rf gpurun_out/prof_${tag}
head -12 gpurun_out/${tag}_bench_smid_kernel_stats.csv | cut -c1-160
cat gpurun_out/${tag}_k_flush_trace_vs_events.json
