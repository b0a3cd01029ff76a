#!/bin/bash
# A/B of the full bench on ONE box: tools/ab_bench.sh <rounds> libA.so libB.so ...  (GR4PM_LIB selects the build;
# "tree" = the working tree's library).  Boxes differ by +-5 %: only numbers of one call compare.
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    if [ "$L" = tree ]; then unset GR4PM_LIB; else export GR4PM_LIB=$PWD/$L; fi
    python3 bench.py --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --repeats 3 ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['values'], 'corr', d['roofline']['launch_ms'])"
  done
done
