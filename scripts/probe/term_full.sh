#!/bin/bash
# like term_ab.sh, but the whole JSON row of every run: term_full.sh out.log WORKLOAD BATCH "ENV ..." ...
out=$1; wl=$2; B=$3; shift 3
: > $out
for cfg in "$@"; do
  echo "== $wl batch $B | $cfg" >> $out
  env $cfg timeout -k 10 500 python scripts/run_to_termination.py $wl $B 1e-7 2>/dev/null | grep '^{' >> $out
done
