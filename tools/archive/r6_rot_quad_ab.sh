#!/bin/bash
# the four-lane form of the phasor chain (k_rot_checkpoints_quad) against the one-lane form, same box:
#   GR4PM_ROT_QUAD_MAX = 0 (never) | default (up to 2048 segments a call) | 100000000 (always)
R=$GRAFT_REPO_ROOT
for q in 0 2048 100000000; do
  echo "== GR4PM_ROT_QUAD_MAX=$q"
  GR4PM_ROT_QUAD_MAX=$q python3 $R/bench.py --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-pmc-traffic --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'Msps;', ' '.join('%s %.0f' % (k, v['value']) for k, v in d['sparse']['streams'].items()))"
  GR4PM_ROT_QUAD_MAX=$q bash $R/tools/r6_step_kstats.sh rq$q | grep -E "k_rot_checkpoints"
done
