#!/bin/bash
# per-kernel durations of one bench.py run (pipelined chain and one-stream) for the library in GR4PM_LIB (default: the
# tree's): tools/r4_kstats.sh <out-dir-name>   (run on the GPU box; summaries land in gpurun_out/<name>/)
NAME=${1:-r4_kstats}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in pipe one; do
  extra=""; [ $mode = one ] && extra="--no-pipeline --no-lookahead"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME/$mode -- python3 $R/bench.py --steps 12 --warmup 4 \
      --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --repeats 1 $extra > /dev/null 2>&1
  echo "== $NAME $mode"; python3 $R/tools/kstats.py $R/gpurun_out/$NAME/$mode | sort -k5 -n -r | head -16
  find $R/gpurun_out/$NAME/$mode -name "*kernel_trace.csv" -delete; find $R/gpurun_out/$NAME/$mode -name "*.db" -delete
done
