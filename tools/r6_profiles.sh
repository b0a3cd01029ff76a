#!/bin/bash
# Round 6's evidence in one call on the GPU box (everything lands in gpurun_out/r6_final/; tools/r6_collect.py copies what
# is to be judged into profiles/r6_*): the driver's bench call; kernel statistics of the headline chain pipelined / one
# stream; kernel statistics of the WHOLE receiver on packets back to back (decode_headers: the full form and the
# packets_only form) -- profiles/r6_kernel_stats_decode_headers*.csv; HBM traffic of the correlator at 2^28 samples per
# launch with warm launches (nine bins, one bin); the access-pattern microbenchmark; the round's same-box A/B records.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_final
rm -rf $O; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-sparse-leg --no-per-bins --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pipe -- python3 $R/bench.py --steps 20 --warmup 5 $COMMON > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one -- python3 $R/bench.py --steps 5 --warmup 2 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
# the whole receiver on packets back to back: per-kernel statistics of 24 passes, the two forms
R6_FIELDS=packets R6_LEAN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dense_lean -- python3 $R/tools/r6_dense_kstats.py 24 > $O/dense_lean_run.txt 2>&1
R6_LEAN=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dense_full -- python3 $R/tools/r6_dense_kstats.py 24 > $O/dense_full_run.txt 2>&1
cd $R
python3 tools/pmc_other_kernels.py $O/pmc_fetch $O/pmc_write $O/kernels_hbm_traffic.json > /dev/null
for m in pipe one dense_lean dense_full; do cp $(ls -t $(find $O/stats_$m -name "*kernel_stats.csv") | head -1) $O/kernel_stats_$m.csv; done
bash tools/pmc_correlate.sh r6_final/pmc_corr9 268435456 4 > $O/pmc_corr9.log 2>&1
bash tools/pmc_correlate.sh r6_final/pmc_corr1 268435456 0 > $O/pmc_corr1.log 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/osp tools/overlap_save_pattern.hip && /tmp/osp 28 20 > $O/overlap_save_pattern.txt 2>&1
{ echo "## the correlator alone, pairs of lines: round 5's fixed shares (GR4PM_W64_BALANCED=0), then grids of whole rounds (the default); three rounds of 2^28 and 2^26 samples at 1 / 3 / 9 bins (tools/r6_balanced_ab.sh)";
  tools/r6_balanced_ab.sh /tmp/bal.txt > /dev/null 2>&1; grep -v amdgpu.ids /tmp/bal.txt;
  echo "## the whole receiver on packets back to back, 48 passes of 2^28 samples (tools/r6_dense_ab.sh): two rounds of the packets_only receiver (result_fields=packets), then two rounds of the full form";
  R6_FIELDS=packets PASSES=48 tools/r6_dense_ab.sh 2 "-"; R6_LEAN=0 PASSES=48 tools/r6_dense_ab.sh 2 "-";
  echo "## ... packets_only, the PLL's kernel form in the decoding stage: 121 | 71 | 32 VGPRs (GR4PM_COSTAS_SMALL_DECODE)";
  R6_FIELDS=packets PASSES=48 tools/r6_dense_ab.sh 2 "GR4PM_COSTAS_SMALL_DECODE=0" "GR4PM_COSTAS_SMALL_DECODE=1" "GR4PM_COSTAS_SMALL_DECODE=2";
  echo "## ... run length (fill and drain of the six-stage pipeline against the steady state)";
  for p in 6 12 24 48 96; do R6_FIELDS=packets R6_LEAN=1 python3 tools/r6_dense_kstats.py $p 2>/dev/null | tail -1; done;
  echo "## CrcCheck inside the packets_only receiver, packets back to back, 48 passes: slicing by eight (default) against the byte-wise kernel";
  R6_FIELDS=packets PASSES=48 tools/r6_dense_ab.sh 3 "-" "GR4PM_CRC_BYTEWISE=1";
  echo "## one packet per 2^20 samples (bench.py --sparse-leg-only): value Msps, steady-state ms per 2^28 -- hardware queues of the process x the phasor chains of consecutive batches one kernel per batch (GR4PM_ROT_SERIAL=1) | side by side";
  for q in 4 16 32; do for ser in 1 0; do
    if [ $ser = 1 ]; then export GR4PM_ROT_SERIAL=1; else unset GR4PM_ROT_SERIAL; fi
    echo -n "GPU_MAX_HW_QUEUES=$q $([ $ser = 1 ] && echo 'one kernel per batch' || echo 'side by side (where the library chooses it)'): ";
    GPU_MAX_HW_QUEUES=$q python3 bench.py --sparse-leg-only --sparse-streams "one_packet_per_2^20" 2>/dev/null | tail -1 | python3 -c "import json,sys; v=json.loads(sys.stdin.read())['streams']['one_packet_per_2^20']; print(v['value'], v['steady_state_ms_per_2^28'])";
  done; done; unset GR4PM_ROT_SERIAL;
} > $O/ab.txt 2>/dev/null
{ python3 tools/benchmark_syncword_detection.py 4 9.5 | tail -1; python3 tools/benchmark_syncword_detection.py 0 9.5 | tail -1;
  python3 tools/benchmark_packet_receiver.py all 9.5 268435456 2 | tail -1;
  WARM=30 python3 tools/bench_correlate.py 268435456 40 4 | tail -1; WARM=30 python3 tools/bench_correlate.py 268435456 40 1 | tail -1;
  WARM=30 python3 tools/bench_correlate.py 268435456 40 0 | tail -1; } > $O/tools.txt 2>/dev/null
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
