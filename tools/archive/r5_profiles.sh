#!/bin/bash
# Round 5's evidence in one call on the GPU box (everything lands in gpurun_out/r5_final/; tools/r5_collect.py copies what
# is to be judged into profiles/r5_*): the driver's bench call, kernel statistics pipelined / one stream (no per-bin,
# config5, channel, sparse or host legs: the correlator's average is the nine-bin launch of the headline chain), HBM
# traffic of the correlator (FETCH_SIZE / WRITE_SIZE in their own --pmc passes) and of the other kernels, vector
# instructions per kernel, the A/B records of the round, the kernels the at-size parity test runs, the reference
# benchmarks' counterparts.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final
rm -rf $O; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-sparse-leg --no-per-bins --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pipe -- python3 $R/bench.py --steps 20 --warmup 5 $COMMON > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one -- python3 $R/bench.py --steps 5 --warmup 2 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/valu_one -- python3 $R/bench.py --steps 4 --warmup 2 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
# the kernels the headline's at-size parity test runs (k_costas_cap on the first two batches, k_costas<*, 8> on the last
# one and in the sequential receiver)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_parity -- python3 -m pytest $R/tests/test_gpu_parity.py -q -x -k headline_configuration_at_its_size > $O/parity_test.log 2>&1
cd $R
python3 tools/pmc_other_kernels.py $O/pmc_fetch $O/pmc_write $O/kernels_hbm_traffic.json > /dev/null
python3 tools/pmc_chain_valu.py $O/valu_one $O/chain_valu_one_stream.json > $O/chain_valu_one_stream.txt
for m in pipe one parity; do cp $(ls -t $(find $O/stats_$m -name "*kernel_stats.csv") | head -1) $O/kernel_stats_$m.csv; done
bash tools/pmc_correlate.sh r5_final/pmc_corr9 67108864 4 > $O/pmc_corr9.log 2>&1
bash tools/pmc_correlate.sh r5_final/pmc_corr1 67108864 0 > $O/pmc_corr1.log 2>&1
{ echo "## A/B, same box (tools/w64_variants.py 2^28 items, nine bins then one: default | round 4's power stores)";
  python3 tools/w64_variants.py 268435456 4 7 -1,262144 | tail -3; python3 tools/w64_variants.py 268435456 0 7 -1,262144 | tail -3;
  echo "## A/B, same box (tools/ab_env.sh): candidates + median tests in one pass | round 4's two passes";
  BENCH_ARGS="--no-sparse-leg --steps 30 --warmup 6" bash tools/ab_env.sh 2 - GR4PM_SD_SEPARATE_MEDIAN=1;
  echo "## A/B, same box (tools/ab_env.sh): symbol filter with one tile per workgroup | two (round 3 / 4)";
  BENCH_ARGS="--no-sparse-leg --steps 30 --warmup 6" bash tools/ab_env.sh 2 - GR4PM_SYMF_TILES=2;
  echo "## A/B, same box: phasor fixed point | every segment a serial chain (zeros -> whole receiver, nine templates, 2^26 per batch)";
  python3 tools/benchmark_packet_receiver.py 4 9.5 67108864 2 | tail -1; GR4PM_ROT_NO_FIXED_POINT=1 python3 tools/benchmark_packet_receiver.py 4 9.5 67108864 2 | tail -1;
} > $O/ab.txt 2>/dev/null
{ python3 tools/benchmark_syncword_detection.py 4 9.5 | tail -1; python3 tools/benchmark_syncword_detection.py 0 9.5 | tail -1;
  python3 tools/benchmark_packet_receiver.py all 9.5 268435456 2 | tail -1;
  python3 tools/bench_correlate.py 67108864 10 4 | tail -1; python3 tools/bench_correlate.py 67108864 10 0 | tail -1;
  python3 tools/c4096_variants.py 67108864 4 5 1,17 | tail -3; python3 tools/c4096_variants.py 67108864 0 5 1,17 | tail -3;
  python3 tools/symf_long_time.py; } > $O/tools.txt 2>/dev/null
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
