#!/bin/bash
# every view of bench.py on one box, back to back (profiles/r<N>_bench_views.txt); run on the GPU box
OUT=${1:-gpurun_out/r3_views}
mkdir -p $OUT
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-channels-leg --repeats 3 "$@" 2>/dev/null | tail -1 > $OUT/$name.json; python3 - $OUT/$name.json "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
    print(f"{sys.argv[2]:<28} {d['value']/1e3:8.2f} Gsps  {d['ms_per_step']:8.3f} ms/step   correlator launch {r.get('launch_ms')} ms")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run default
run lookahead1 --lookahead-depth 1
run no_lookahead --no-lookahead
run one_stream --no-pipeline --no-lookahead
run python_pipeline --python-pipeline
run soft_bits --soft-bits
run decode_headers --decode-headers
run detector_only --detector-only
run channels64 --channels 64 --steps 40 --warmup 6
run channels64_sync --channels 64 --steps 40 --warmup 6 --no-pipeline
run channels64_detector --channels 64 --steps 40 --warmup 6 --detector-only
run config5 --config 5 --steps 10 --warmup 3
GR4PM_W64_VARIANT=0 run default_round2_correlator
GR4PM_CORRELATOR=wave run default_round1_correlator
for sk in costas rot symf costas,rot costas,rot,symf; do GR4PM_TIMING_SKIP=$sk run timing_only_without_$sk; done
python3 tools/benchmark_syncword_detection.py 4 9.5 2>/dev/null | tail -2
python3 tools/benchmark_syncword_detection.py 0 9.5 2>/dev/null | tail -2
python3 tools/bench_correlate.py 67108864 10 4 2>/dev/null | tail -1
python3 tools/bench_correlate.py 67108864 10 0 2>/dev/null | tail -1
