#!/bin/bash
# round 6: same-box A/B of the packets_only receiver on the dense-packet stream over environment settings
# usage: tools/r6_dense_ab.sh <rounds> "VAR=a [VAR2=b]" "VAR=c" ...   (each argument one configuration; "-" = defaults)
R=$GRAFT_REPO_ROOT
rounds=$1; shift
for r in $(seq $rounds); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    echo -n "round $r [$cfg] "
    env $envs R6_LEAN=${R6_LEAN:-1} python3 $R/tools/r6_dense_kstats.py ${PASSES:-12} 2>/dev/null | tail -1
  done
done
