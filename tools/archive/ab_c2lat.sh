#!/bin/bash
# config2 leg (rate + latency block) under different settings: tools/ab_c2lat.sh "VAR=a" "VAR=b" ...  ("-" = none)
for E in "$@"; do
  if [ "$E" = "-" ]; then E=""; fi
  env $E python3 bench.py --no-cpu-baseline --no-config5-leg --no-host-stream-leg --no-per-bins --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config2']; b=c['latency']['batches']
print('[$E] headline', d['value'], 'config2', c['value'], {k:(v['latency_ms'], v['msps']) for k,v in b.items()})"
done
