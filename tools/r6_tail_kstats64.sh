#!/bin/bash
# the detector tail's kernels per step of config 2 (64 channels x 2^22 samples) on ONE stream:
#   bash tools/r6_tail_kstats64.sh [tag]      -> gpurun_out/r6_tail64_<tag>.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-now}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r6_tail64_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --channels 64 --items 4194304 --steps 5 --warmup 2 --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --repeats 1 --no-pipeline --no-lookahead --no-pmc-traffic > /dev/null 2>&1
python3 $R/tools/kstats.py $O k_candidates_wave k_tile_tables k_group_tables k_super_tables k_tile_visit k_median_tests k_resolve k_scan_entries k_tags k_compact | tee $R/gpurun_out/r6_tail64_$TAG.txt
rm -rf $O
