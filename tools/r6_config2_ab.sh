#!/bin/bash
# config 2 (64 channels x 2^22 samples through gr4pm_multichannel_receiver) by library: the one of commit 3b8143c
# (tools/r6_tail_ab.sh says how to build it) | the tree's with the one generic chain kernel (GR4PM_ROT_GENERIC=1) | the tree's
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
for cfg in "GR4PM_LIB=$R/tools/ab/libgr4pm_hip_before_tail.so" "GR4PM_ROT_GENERIC=1" "X=1"; do
  echo -n "[$cfg] "
  env $cfg python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --no-pmc-traffic 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config2']; print('config 2', c['value'], c['ms_per_step'], c.get('value_min'), c.get('value_max'), 'warm regions', c.get('warm_regions'))"
done; done
