#!/bin/bash
# round 6: kernel timeline of the headline (pipelined front end): how continuously the correlator runs, what a steady step holds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6_headline_trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
COMMON="--no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-sparse-leg --no-per-bins --repeats 1"
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 30 --warmup 5 $COMMON > $O/run.txt 2>&1
python3 -c "
import json
d=json.loads([l for l in open('$O/run.txt').read().splitlines() if l.startswith('{')][-1]); print('bench under the tracer:', d['value'], d['ms_per_step'])"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 $R/tools/corr_gaps.py $f | tee $O/corr_gaps.txt
python3 - "$f" > $O/steady_step.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30], r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
corr = [e for e in ev if e[2].startswith("k_correlate")]
mid = 18  # inside the timed region of the headline (5 warm-up + 30 steps come first; the roofline leg's launches follow)
a, b = corr[mid][0], corr[mid + 2][0]
span = corr[30][0] - corr[10][0]
busy = sum(e - s for s, e, n, q in corr[10:30])
print(f"correlator launches 10 .. 30 of the region: launch-to-launch {span / 20 / 1e3:.0f} us, mean duration {busy / 20 / 1e3:.0f} us")
print(f"window {(b - a) / 1e3:.0f} us = two correlator launches apart (launch {mid} of {len(corr)})")
for s, e, n, q in ev:
    if s >= a and s < b and (e - s) > 10000:
        print(f"{q:>4} {n:<34}{(s - a) / 1e3:10.1f}{(e - s) / 1e3:10.1f}")
PY
head -70 $O/steady_step.txt
rm -rf $O/prof
