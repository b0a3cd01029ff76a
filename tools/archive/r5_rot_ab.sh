#!/bin/bash
# round 5: what k_rot_checkpoints costs the pipelined chain and how much of that its (scattered) stores are -- timing only
# (EXPERIMENTS builds: tools/build_variant.sh exp; tools/build_variant.sh rotnostore EXTRA=-DGR4PM_ROT_NO_STORES)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "cfc or rotator or receiver or headline" 2>&1 | tail -2
python tools/fuzz_cfc_symf.py 100 2>&1 | tail -1
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --steps 30 --warmup 6"
run() { name=$1; shift; python bench.py $COMMON "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-34s %7.2f Gsps %7.3f ms" % (sys.argv[1], d["value"]/1e3, d["ms_per_step"]))' "$name"; }
for i in 1 2 3; do
GR4PM_LIB=$PWD/tools/ab/libgr4pm_exp.so run all
[ -f tools/ab/libgr4pm_rotnostore.so ] && GR4PM_LIB=$PWD/tools/ab/libgr4pm_rotnostore.so run rot_without_stores
[ -f tools/ab/libgr4pm_rot8.so ] && GR4PM_LIB=$PWD/tools/ab/libgr4pm_rot8.so run rot_8_byte_stores
GR4PM_LIB=$PWD/tools/ab/libgr4pm_exp.so GR4PM_TIMING_SKIP=rot run without_rot
done
