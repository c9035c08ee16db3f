#!/bin/bash
# kernel-trace profile of one bench configuration: prof_bench.sh tag [ENV=..] [bench flags]
tag=$1; shift
envs=""; flags=""
for w in "$@"; do case $w in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
export TMPDIR=/tmp
for e in $envs; do export $e; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o ${tag} -- python3 bench.py --no-cpu-baseline $flags > gpurun_out/${tag}_profiled.json 2> gpurun_out/${tag}_prof.err || { tail -5 gpurun_out/${tag}_prof.err; exit 1; }
ks=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1)
cp "$ks" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_${tag}
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
for r in rows[:16]:
    print("%-60s calls %6s avg_us %9.2f total_ms %9.2f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
