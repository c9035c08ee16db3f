# ex09, all phases, tableau form (the default) at the final code; the ex09-lp kernel statistics; the default bench line once more
export TMPDIR=/tmp
mkdir -p gpurun_out/ex09r
( cd gpurun_out/ex09r && BSLV_LP_TIMING=1 BSLV_LP_REV=0 timeout -k 10 400 ../../bensolve_amd/csrc/bensolve_hip ../../tests/golden/ex/ex09.vlp -e 1e-2 -o tab > tab.out 2>&1; python3 ../../scripts/probe/lp_timing_sum.py tab.out; grep -a "CPU time\|LPs solved\|not bounded" tab.out )
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ex09 -o e -- python3 bench.py --workload ex09-lp > /dev/null 2> gpurun_out/r04_ex09_prof.err && cp $(find gpurun_out/prof_ex09 -name "*kernel_stats.csv" | head -1) gpurun_out/r04_ex09_lp_kernel_stats.csv && rm -rf gpurun_out/prof_ex09 && head -5 gpurun_out/r04_ex09_lp_kernel_stats.csv | cut -c1-160
python3 bench.py --no-cpu-baseline --no-pair > gpurun_out/r04_bench_smid_check.json 2> /dev/null; python3 -c "
import json; b=json.loads(open('gpurun_out/r04_bench_smid_check.json').read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'], b['cuts_applied'], b['poly_rounds'], b['pivots_per_lp'])"
