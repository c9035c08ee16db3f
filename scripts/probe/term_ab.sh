#!/bin/bash
# whole runs to termination under different batch selection rules: term_ab.sh out.log WORKLOAD BATCH "ENV ..." ...
out=$1; wl=$2; B=$3; shift 3
: > $out
for cfg in "$@"; do
  echo "== $wl batch $B | $cfg" >> $out
  env $cfg timeout -k 10 500 python scripts/run_to_termination.py $wl $B 1e-7 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('   ', {k: d[k] for k in ('seconds','steps','lps','cuts','redundant','confirmed','pivots','lps_per_sec','useful_lps_per_sec','ms_lp','ms_poly','vertices','facets')}, d['vertices_sha256_at_1e6'][:12])
" >> $out
done
cat $out
