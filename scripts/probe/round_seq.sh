#!/bin/bash
# the sequence of device operations of a few consecutive rounds in the middle of the default bench run (kernel trace incl. fills and copies)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_seq
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_seq -o t -- python3 bench.py --no-cpu-baseline --no-long-window --no-pair > /dev/null 2> gpurun_out/seq.err || { tail -3 gpurun_out/seq.err; exit 1; }
python3 - <<EOF
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/prof_seq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bslv::","").replace("void ","")[:40]))
for f in glob.glob("gpurun_out/prof_seq/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# find the 700th minit
idx=[i for i,r in enumerate(rows) if r[2]=="k_r2_minit"]
i0=idx[700]; i1=idx[704]
t0=rows[i0][0]
for s,e,n in rows[i0:i1]:
    print("%9.1f us  +%6.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
EOF
rm -rf gpurun_out/prof_seq
