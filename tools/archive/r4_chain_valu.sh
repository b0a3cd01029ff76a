#!/bin/bash
# vector-ALU wave-instructions of every kernel of a headline step (pipelined and one stream): one --pmc pass each
# (rocprofv3 --kernel-trace --pmc only), summarised by tools/pmc_chain_valu.py into gpurun_out/r4_valu/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_valu
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --repeats 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/pipe -- python3 $R/bench.py --steps 4 --warmup 2 $COMMON > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/one -- python3 $R/bench.py --steps 4 --warmup 2 $COMMON --no-pipeline --no-lookahead > /dev/null 2>&1
cd $R
python3 tools/pmc_chain_valu.py $O/pipe $O/chain_valu_pipelined.json > $O/chain_valu_pipelined.txt
python3 tools/pmc_chain_valu.py $O/one $O/chain_valu_one_stream.json > $O/chain_valu_one_stream.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
head -14 $O/chain_valu_pipelined.txt
