#!/bin/bash
# Kernel trace of the default bench run, cut-phase kernels by their position in the hot chunk (a chunk starts at k_r2_words3).
# usage: scripts/probe/round_trace.sh <tag>   (environment switches are taken from the caller's environment)
tag=${1:-t}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -o t -- python3 bench.py --no-cpu-baseline --no-long-window --no-pair > gpurun_out/rt_$tag.json 2> gpurun_out/rt_$tag.err || { tail -3 gpurun_out/rt_$tag.err; exit 1; }
kt=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$kt" gpurun_out/rt_$tag.json <<EOF
import csv, sys, collections, json
import numpy as np
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].split("(")[0].replace("bslv::","").replace("void ","")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
try:
    b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("value %.0f LPs/s, ms/step %.2f, phases %s, us/cut %.2f, cuts/pass %.1f"%(b["value"],b["ms_per_step"],b["phase_ms_per_step"],b["roofline_cuts"]["us_per_cut"],b["roofline_cuts"]["cuts_per_pass"]))
except Exception as e: print("no bench line:", e)
d=collections.defaultdict(list)
for s,e,n in rows: d[n].append((e-s)/1e3)
print("kernel                  calls   mean   median    p90    p99   total_ms")
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:14]:
    v=np.array(v); print(k[:24].ljust(24), "%6d %7.1f %7.1f %7.1f %7.1f %9.1f"%(len(v),v.mean(),np.median(v),np.percentile(v,90),np.percentile(v,99),v.sum()/1e3))
names=("k_r2_minit","k_r2_select3","k_r2_assign3","k_flags2","k_r2_emit","k_r2_classify3","k2_fused_t<true>","k_r2_k2emit")
pos={n:0 for n in names}; bypos={n:collections.defaultdict(list) for n in names}
for s,e,n in rows:
    if n=="k_r2_words3":
        for k in pos: pos[k]=0
    n2=n.split("<")[0] if not n.startswith("k2_fused") else n
    for k in names:
        if n2==k or n==k or (k.startswith("k_r2_emit") and n.startswith("k_r2_emit<")) or (k=="k_r2_classify3" and n.startswith("k_r2_classify3<")):
            bypos[k][min(pos[k],45)].append((e-s)/1e3); pos[k]+=1; break
print("mean us by round of the chunk:   ", "  ".join("%5d"%p for p in range(0,46,3)))
for k in names:
    print(k[:18].ljust(18), "  ".join("%5.0f"%np.mean(bypos[k][p]) if bypos[k][p] else "    -" for p in range(0,46,3)))
EOF
rm -rf gpurun_out/prof_$tag
