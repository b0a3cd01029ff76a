#!/bin/bash
# round 6: the randomised differential tools and the soak tools in one gpurun call (tools/archive/r4_soak.sh with the
# packets_only receiver added).  usage: tools/r6_soak.sh [seed]
S=${1:-6}
cd $GRAFT_REPO_ROOT
python3 tools/fuzz_detector.py 300 $S | tail -1
python3 tools/fuzz_cfc_symf.py 200 $S | tail -1
python3 tools/fuzz_costas.py 90 $S | tail -1
GR4PM_COSTAS_FORM=2 GR4PM_COSTAS_CAP_MIN_LOG2=0 python3 tools/fuzz_costas.py 60 $S | tail -1
for m in plain soft decode lean; do python3 tools/stress_receiver.py 150 $S $m | tail -1; done
for s in $((S+1)) $((S+2)) $((S+3)); do python3 tools/stress_receiver.py 120 $s lean | tail -1; python3 tools/stress_receiver.py 120 $s decode | tail -1; done
python3 tools/stress_multichannel.py 2>&1 | tail -1
