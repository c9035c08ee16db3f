#!/bin/bash
# A/B of batch selection rules on S-mid through bench.py (no CPU baseline): one line per configuration
# usage: fronts_ab.sh out.log "ENV=.. --flags" ...
out=$1; shift
: > $out
for cfg in "$@"; do
  envs=""; flags=""
  for w in $cfg; do case $w in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  echo "== $cfg" >> $out
  env $envs python bench.py --no-cpu-baseline $flags 2>>$out | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        rc = d['roofline_cuts']
        print('   ', {k: d[k] for k in ('value','useful_lps_per_sec','lps_redundant_frac','ms_per_step','cuts_applied','pivots_per_lp','phase_ms_per_step','poly_rounds','live_vertices')}, {k: rc[k] for k in ('us_per_cut','cuts_per_pass','single_cut_pipeline_cuts','hot_chunks','mailbox')})
" >> $out
done
cat $out
