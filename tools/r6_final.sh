#!/bin/bash
# Round 6's closing evidence in one gpurun call: tools/r6_profiles.sh, then the detector tail's kernels on one stream (one
# channel and 64), the headline's kernel timeline and the soak tools.  tools/r6_collect.py copies the results into profiles/.
R=$GRAFT_REPO_ROOT
bash $R/tools/r6_profiles.sh > /dev/null 2>&1
O=$R/gpurun_out/r6_final
{ echo "## detector tail, stand-alone kernel durations per 2^28-sample step, one channel (tools/r6_tail_kstats.sh)"; bash $R/tools/r6_tail_kstats.sh fin;
  echo "## ... config 2: 64 channels x 2^22 samples (tools/r6_tail_kstats64.sh)"; bash $R/tools/r6_tail_kstats64.sh fin;
  echo "## ... one channel, the two-level scan of rounds 1 - 5 (GR4PM_SD_NO_SUPER=1)"; GR4PM_SD_NO_SUPER=1 bash $R/tools/r6_tail_kstats.sh nosuper | grep -E "k_scan|k_group|k_super"; } > $O/detector_tail.txt 2>&1
bash $R/tools/r6_headline_trace.sh r6_final/trace > $O/headline_trace.txt 2>&1
bash $R/tools/r6_soak.sh 6 > $O/soak.txt 2>&1
bash $R/tools/r6_soak.sh 61 >> $O/soak.txt 2>&1
ls $O
