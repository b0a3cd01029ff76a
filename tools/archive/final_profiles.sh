cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_stats_pipe -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-channels-leg --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_stats_one -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-channels-leg --repeats 1 --no-pipeline --no-lookahead > /dev/null 2>&1
cd $R && python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
