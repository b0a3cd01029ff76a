#!/bin/bash
# The round's evidence in one call on the GPU box (everything lands in gpurun_out/r4_final/, then copy what is to be
# judged into profiles/): the driver's bench call, kernel statistics pipelined / one stream (no per-bin legs, no
# config5 / channel legs: the correlator's average is the nine-bin launch of the headline chain), HBM traffic of the
# correlator (FETCH_SIZE / WRITE_SIZE in their own --pmc passes) and of the other kernels, the reference benchmarks'
# counterparts, the correlator alone.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_final
mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pipe -- python3 $R/bench.py --steps 20 --warmup 5 $COMMON > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one -- python3 $R/bench.py --steps 5 --warmup 2 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
cd $R
python3 tools/pmc_other_kernels.py $O/pmc_fetch $O/pmc_write $O/kernels_hbm_traffic.json > /dev/null
for m in pipe one; do cp $(ls -t $(find $O/stats_$m -name "*kernel_stats.csv") | head -1) $O/kernel_stats_$m.csv; done
bash tools/pmc_correlate.sh r4_final/pmc_corr9 67108864 4 > $O/pmc_corr9.log 2>&1
bash tools/pmc_correlate.sh r4_final/pmc_corr1 67108864 0 > $O/pmc_corr1.log 2>&1
{ python3 tools/benchmark_syncword_detection.py 4 9.5 | tail -1; python3 tools/benchmark_syncword_detection.py 0 9.5 | tail -1;
  python3 tools/benchmark_packet_receiver.py all 9.5 67108864 2 | tail -1;
  python3 tools/bench_correlate.py 67108864 10 4 | tail -1; python3 tools/bench_correlate.py 67108864 10 0 | tail -1;
  python3 tools/c4096_variants.py 67108864 4 5 0,1 | tail -2; python3 tools/c4096_variants.py 67108864 0 5 0,1 | tail -2;
  python3 tools/symf_long_time.py; GR4PM_SYMF_GENERIC=1 python3 tools/symf_long_time.py; } > $O/tools.txt 2>/dev/null
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
