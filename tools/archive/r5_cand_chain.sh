#!/bin/bash
# k_candidates_wave<12, true> (candidates + median tests in one pass): time per 2^28 samples against the blocks a wave walks
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  O=$R/gpurun_out/r5_chain_$c; mkdir -p $O
  GR4PM_CAND_CHAIN=$c rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --repeats 1 --no-pipeline --no-lookahead > /dev/null 2>&1
  echo "chain $c: $(python3 $R/tools/kstats.py $O k_candidates_wave k_tile_visit | tr '\n' ' ')"
  rm -rf $O
done
