#!/bin/bash
# the detector tail's kernels per 2^28-sample step on ONE stream (stand-alone durations, nothing beside them), round 6:
#   bash tools/r6_tail_kstats.sh [tag]      -> gpurun_out/r6_tail_<tag>.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-now}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r6_tail_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --repeats 1 --no-pipeline --no-lookahead --no-pmc-traffic > /dev/null 2>&1
python3 $R/tools/kstats.py $O k_candidates_wave k_tile_tables k_group_tables k_super_tables k_tile_visit k_median_tests k_resolve k_scan_entries k_tags k_compact | tee $R/gpurun_out/r6_tail_$TAG.txt
rm -rf $O
