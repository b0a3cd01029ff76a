#!/bin/bash
# the detector tail's kernels per 2^28-sample step, one stream: fused candidate + median kernel | round 4's two passes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in fused separate; do
  O=$R/gpurun_out/r5_tail_$mode; mkdir -p $O
  if [ $mode = separate ]; then export GR4PM_SD_SEPARATE_MEDIAN=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --repeats 1 --no-pipeline --no-lookahead > /dev/null 2>&1
  echo "== $mode"; python3 $R/tools/kstats.py $O k_candidates_wave k_tile_visit k_median_tests k_resolve k_scan_entries k_tags k_compact
  rm -rf $O
done
