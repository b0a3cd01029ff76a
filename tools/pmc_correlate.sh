#!/bin/bash
# PMC passes over the correlator kernel alone (tools/bench_correlate.py); run on the GPU box:
#   bash tools/pmc_correlate.sh <tag> [items] [bins]
# Counters go to gpurun_out/<tag>/; separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-pmc}; ITEMS=${2:-67108864}; BINS=${3:-4}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P3="GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_F32"
i=0
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  WARM=2 timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 tools/bench_correlate.py $ITEMS 3 $BINS > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 tools/pmc_summary.py k_correlate $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 > /dev/null
# kernel statistics of WARM launches (round 6: 30 untimed launches, then 40; the first launches after host work run at the
# clocks the chip idles at -- bench.py's ROOF_WARM -- and round 5's file averaged six of those)
WARM=30 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_correlate.py $ITEMS 40 $BINS > $OUT/stats.log 2>&1
tail -2 $OUT/stats.log
cat $OUT/summary.json | head -80
