export BSLV_RUN_EX09=1
BSLV_LP_REV=1 timeout -k 10 600 python -m pytest tests/test_cli_gpu.py -m gpu -x -q -k ex09_is_certified 2>&1 | tail -2
timeout -k 10 300 python3 bench.py --workload ex09-lp > gpurun_out/r04_ex09_lp.json 2> gpurun_out/r04_ex09_lp.err
python3 scripts/probe/show_ex09_lp.py gpurun_out/r04_ex09_lp.json
unset BSLV_RUN_EX09
timeout -k 10 600 python -m pytest tests/test_lp_gpu.py tests/test_lp_compat_gpu.py tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -2
