#!/bin/bash
# A/B of the full bench on ONE box over environment settings: tools/ab_env.sh <rounds> "VAR=a" "VAR=b OTHER=c" ...
# ("-" = no setting).  Boxes differ by +-5 %: only numbers of one call compare.
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then E=""; fi
    env $E python3 bench.py --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --repeats 3 ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$E]', d['value'], d['ms_per_step'], d['values'], 'corr', d['roofline']['launch_ms'])"
  done
done
