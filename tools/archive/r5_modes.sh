#!/bin/bash
# every view of bench.py that the tag-count assertion guards, one short run each (a mode whose tags do not match the
# generator's packets exits non-zero)
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --steps 10 --warmup 3"
run() { name=$1; shift; out=$(timeout 600 python bench.py $C "$@" 2>&1 | tail -1); echo "$out" | python -c 'import json,sys
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print("%-26s %8.2f Gsps %8.3f ms" % (sys.argv[1], d["value"]/1e3, d["ms_per_step"]))
except Exception: print(sys.argv[1], "FAILED:", t[:300])' "$name"; }
run default
run lookahead1 --lookahead-depth 1
run no_lookahead --no-lookahead
run one_stream --no-pipeline --no-lookahead
run python_pipeline --python-pipeline
run soft_bits --soft-bits
run decode_headers --decode-headers
run detector_only --detector-only
run copy_delay --copy-delay
run channels64 --channels 64 --steps 20 --warmup 4
run channels64_sync --channels 64 --steps 20 --warmup 4 --no-pipeline
run channels64_detector --channels 64 --steps 20 --warmup 4 --detector-only
run config5 --config 5 --steps 5 --warmup 2
