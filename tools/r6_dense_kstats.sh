#!/bin/bash
# round 6: the whole receiver on packets back to back -- per-kernel statistics (rocprofv3) and per-stage wall times
# (GR4PM_TIMING build) of the decode_headers pipeline.  usage: tools/r6_dense_kstats.sh <out dir under gpurun_out>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6_dense}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/r6_dense_kstats.py 12 > $O/plain.txt 2>&1; tail -1 $O/plain.txt
if [ -n "$R6_LEAN" ]; then echo "(packets_only receiver)"; fi
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/r6_dense_kstats.py 8 > $O/prof_run.txt 2>&1
tail -1 $O/prof_run.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
python3 $R/tools/kstats.py $O/prof > $O/kstats.txt; sort -k5 -n -r $O/kstats.txt | head -45
rm -rf $O/prof
if [ -f $R/tools/ab/libgr4pm_timing.so ]; then
  GR4PM_LIB=$R/tools/ab/libgr4pm_timing.so python3 $R/tools/r6_dense_kstats.py 16 > $O/timing.txt 2>&1
  grep "gr4pm timing" $O/timing.txt | tail -12
fi
