#!/bin/bash
# round 6: kernel timeline of the packets_only receiver on the dense-packet stream: correlator gaps and what runs in them
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6_trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
R6_LEAN=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/tools/r6_dense_kstats.py 9 > $O/run.txt 2>&1
tail -1 $O/run.txt
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
a, b = corr[-5][0], corr[-3][0]   # two steady steps near the end
print(f"window {(b - a) / 1e3:.0f} us = two correlator launches apart")
for s, e, n, q in ev:
    if s >= a and s < b and (e - s) > 15000:
        print(f"{q:>4} {n:<34}{(s - a) / 1e3:10.1f}{(e - s) / 1e3:10.1f}")
PY
head -80 $O/steady_step.txt
rm -rf $O/prof
