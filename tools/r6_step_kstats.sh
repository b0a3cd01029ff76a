#!/bin/bash
# every kernel of one 2^28-sample step of the headline workload on ONE stream (stand-alone durations):
#   bash tools/r6_step_kstats.sh [tag] [extra bench.py flags]      -> gpurun_out/r6_step_<tag>.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-now}; shift
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r6_step_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --repeats 1 --no-pipeline --no-lookahead --no-pmc-traffic "$@" > /dev/null 2>&1
python3 $R/tools/kstats.py $O | tee $R/gpurun_out/r6_step_$TAG.txt
rm -rf $O
