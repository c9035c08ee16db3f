#!/bin/bash
# timing experiment (results are WRONG with a probe bit set): which part of the revised selection costs what on one cold ex09 LP
export TMPDIR=/tmp
for p in ${PROBES:-0 1 2 4 7}; do
  BSLV_REV_PROBE=$p BSLV_LP_MAXROUNDS=300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p$p -o p -- python3 scripts/probe/ex09_lp_stats.py > gpurun_out/rev_probe_$p.log 2>&1
  ks=$(find gpurun_out/prof_p$p -name "*kernel_stats.csv" | head -1)
  python3 - "$ks" $p <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_select" in r["Name"] or "k_flush" in r["Name"]:
        print("probe", sys.argv[2], r["Name"].split("(")[0], "calls", r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
  grep -a "status" gpurun_out/rev_probe_$p.log | head -2
  rm -rf gpurun_out/prof_p$p
done
