#!/bin/bash
# pool size of the tableaux against pivots per LP over long windows: pool_ab.sh out.log "flags" ...
out=$1; shift
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  python bench.py --no-cpu-baseline --no-long-window $cfg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('   ', {k: d[k] for k in ('value','useful_lps_per_sec','ms_per_step','pivots_per_lp','phase_ms_per_step','warm_starts')})
" >> $out
done
