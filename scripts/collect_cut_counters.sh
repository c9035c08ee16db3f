#!/bin/bash
# SQ / GRBM counters of the cut-phase kernels over a short bench run (one rocprofv3 --pmc pass per group, --kernel-trace only,
# the program itself after `--`): per kernel and launch the mean of every counter -> gpurun_out/<tag>_cut_counters.json
tag=$1
export TMPDIR=/tmp
i=0
: > gpurun_out/${tag}_cut_counters.files
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "MemUnitStalled" "LDSBankConflict"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_c${i}_$tag
  if rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_c${i}_$tag -o c -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 6 > /dev/null 2> gpurun_out/${tag}_c$i.err; then
    find gpurun_out/pmc_c${i}_$tag -name "*counter_collection.csv" | head -1 >> gpurun_out/${tag}_cut_counters.files
  else
    echo "group $i ($grp) failed: $(tail -2 gpurun_out/${tag}_c$i.err)"
  fi
done
python3 - <<PY
import csv, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in open("gpurun_out/${tag}_cut_counters.files").read().split():
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
want = ("k_r2_minit", "k_r2_select3", "k_r2_assign3", "k_r2_prep", "k_r2_select", "k_r2_assign", "k_flags2", "k_r2_emit", "k_r2_classify3", "k2_fused_t<true>", "k_r2_k2emit", "k_flush", "k_select")
out = {}
for name, cs in acc.items():
    if not any(w in name for w in want):
        continue
    row = {c: round(v[0] / max(v[1], 1), 1) for c, v in cs.items()}
    row["launches"] = max(v[1] for v in cs.values())
    if "SQ_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row and row["GRBM_GUI_ACTIVE"]:
        row["sq_busy_over_gui_active"] = round(row["SQ_BUSY_CYCLES"] / row["GRBM_GUI_ACTIVE"], 3)
    if "SQ_WAIT_ANY" in row and row.get("SQ_WAVE_CYCLES"):
        row["wait_share_of_wave_cycles"] = round(row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"], 3)
        row["active_inst_share_of_wave_cycles"] = round(row.get("SQ_ACTIVE_INST_ANY", 0) / row["SQ_WAVE_CYCLES"], 3)
    out[name] = row
json.dump({"command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 6 (one pass per group)", "per_launch_means": out}, open("gpurun_out/${tag}_cut_counters.json", "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
rm -rf gpurun_out/pmc_c*_$tag
