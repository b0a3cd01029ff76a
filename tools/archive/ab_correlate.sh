#!/bin/bash
# A/B timing of the correlator of two builds of the library (same ABI): interleaved processes on one box.
# tools/ab_correlate.sh libA.so libB.so [items] [bins] [rounds]   (each round: tools/w64_variants.py with variant 0)
A=$1; B=$2; N=${3:-67108864}; BINS=${4:-4}; R=${5:-3}
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    echo -n "$(basename $L): "
    GR4PM_LIB=$L python3 tools/w64_variants.py $N $BINS 7 -1 2>&1 | grep "median"
  done
done
