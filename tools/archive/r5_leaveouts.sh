#!/bin/bash
# round 5: what every part of the step costs the pipelined chain (timing only: EXPERIMENTS build, wrong results), one box
R=$GRAFT_REPO_ROOT
cd $R
export GR4PM_LIB=$R/tools/ab/libgr4pm_exp.so
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --steps 30 --warmup 6"
run() { name=$1; shift; python bench.py $COMMON "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-34s %7.2f Gsps %7.3f ms" % (sys.argv[1], d["value"]/1e3, d["ms_per_step"]))' "$name"; }
run all
for sk in costas rot symf costas,rot costas,rot,symf; do GR4PM_TIMING_SKIP=$sk run without_$sk; done
run detector_only --detector-only
run all_again
python3 tools/bench_correlate.py 268435456 10 4 2>/dev/null | tail -1
