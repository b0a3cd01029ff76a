#!/bin/bash
# long soak of every randomised differential tool and stress tool on the GPU box (one gpurun call; seeds given as $1..)
# usage on the box: bash tools/r4_soak.sh [seed0=101]   -> gpurun_out/r4_soak.txt
s=${1:-101}
out=gpurun_out/r4_soak.txt
mkdir -p gpurun_out
: > $out
run() { echo "\$ $*" >> $out; timeout 900 "$@" 2>&1 | tail -2 >> $out; }
run python tools/fuzz_detector.py 400 $s
run python tools/fuzz_cfc_symf.py 300 $s
run python tools/fuzz_costas.py 150 $s
for k in 0 1 2 3 4 5; do run python tools/stress_receiver.py 150 $((s + k)) decode; done
for k in 0 1; do run python tools/stress_receiver.py 150 $((s + k)) plain; run python tools/stress_receiver.py 150 $((s + k)) soft; done
run python tools/stress_multichannel.py 8 120 $s
run python tools/stress_multichannel.py 64 40 $((s + 1))
cat $out
