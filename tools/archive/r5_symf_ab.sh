#!/bin/bash
# round 5, k_symbol_filter_fast with its uniform index arithmetic on the scalar unit: parity first (tests + the CFC / symbol
# filter fuzzer), then the pipelined chain and the kernel alone with one tile per workgroup and with two.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_symf
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "symbol_filter or cfc or receiver or headline or multichannel" 2>&1 | tail -3
python tools/fuzz_cfc_symf.py 150 2>&1 | tail -2
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-sparse-leg --no-per-bins"
for t in 1 2 1 2; do
  echo "tiles=$t: $(GR4PM_SYMF_TILES=$t python bench.py --steps 30 --warmup 6 $COMMON 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
cd /tmp && export TMPDIR=/tmp
for t in 1 2; do
  export GR4PM_SYMF_TILES=$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one_t$t -- python3 $R/bench.py --steps 5 --warmup 2 $COMMON --repeats 1 --no-pipeline --no-lookahead > /dev/null 2>&1
  f=$(ls -t $(find $O/stats_one_t$t -name "*kernel_stats.csv") | head -1)
  echo "tiles=$t one stream:"; grep -i "symbol_filter_fast\|k_correlate_w64\|k_candidates" $f | cut -d, -f1-5 | cut -c1-150
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/valu_t$t -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON --repeats 1 --no-pipeline --no-lookahead > /dev/null 2>&1
  python3 $R/tools/pmc_chain_valu.py $O/valu_t$t $O/valu_t$t.json | grep -i "symbol_filter"
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
