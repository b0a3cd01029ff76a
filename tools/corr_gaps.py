#!/usr/bin/env python3
"""How continuously the correlator has a launch in flight in a pipelined run, from a rocprofv3 --kernel-trace csv:
per steady-state step the time with no k_correlate launch running, and what ran in those gaps.
tools/corr_gaps.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24]))
ev.sort()
corr = [(s, e) for s, e, n in ev if n.startswith("k_correlate")]
corr = corr[len(corr) // 3:2 * len(corr) // 3]  # steady state: the middle third (warm-up before, roofline legs after)
gaps = [(a[1], b[0]) for a, b in zip(corr[:-1], corr[1:]) if b[0] > a[1]]
span = corr[-1][1] - corr[0][0]
busy = sum(e - s for s, e in corr)
print(f"{len(corr)} correlator launches over {span / 1e6:.2f} ms: mean duration {busy / len(corr) / 1e3:.0f} us, "
      f"launch-to-launch {span / (len(corr) - 1) / 1e3:.0f} us; overlap of consecutive launches "
      f"{sum(max(0, a[1] - b[0]) for a, b in zip(corr[:-1], corr[1:])) / (len(corr) - 1) / 1e3:.0f} us per step; "
      f"gaps {sum(b - a for a, b in gaps) / (len(corr) - 1) / 1e3:.0f} us per step ({len(gaps)} gaps)")
inside = Counter()
for a, b in gaps:
    for s, e, n in ev:
        if not n.startswith("k_correlate") and s < b and e > a:
            inside[n] += min(e, b) - max(s, a)
for n, t in inside.most_common(8):
    print(f"   in the gaps: {n:<26}{t / (len(corr) - 1) / 1e3:8.0f} us per step")
