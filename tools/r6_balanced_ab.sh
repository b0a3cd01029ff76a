#!/bin/bash
# round 6: the correlator alone with the grid a whole number of rounds of one workgroup per CU (default) against round 5's
# fixed shares (GR4PM_W64_BALANCED=0), 1 / 3 / 9 bins, 2^28 and 2^26 samples, interleaved, one box
out=${1:-gpurun_out/r6_balanced_ab.txt}
: > $out
for round in 1 2 3; do
  for items in $((1 << 28)) $((1 << 26)); do
    for bins in 0 1 4; do
      for b in 0 1; do
        echo -n "round $round balanced=$b " >> $out
        GR4PM_W64_BALANCED=$b python3 tools/bench_correlate.py $items 40 $bins >> $out 2>&1
      done
    done
  done
done
cat $out
