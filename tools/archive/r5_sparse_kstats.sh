#!/bin/bash
# which kernels a pass over the one-packet-per-2^20 stream waits for
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in 0; do
  :
  O=$R/gpurun_out/r5_sparse_$e; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/r5_sparse_kstats.py 6 2>/dev/null | tail -1
  python3 $R/tools/kstats.py $O k_rot_checkpoints k_costas k_correlate_w64 k_symbol_filter k_rot_const
  rm -rf $O
done
