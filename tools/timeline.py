#!/usr/bin/env python3
"""print the kernel timeline (stream, kernel, start, duration in us) of the last `n` kernels of a
rocprofv3 --kernel-trace csv: tools/timeline.py <kernel_trace.csv> [n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    name = r["Kernel_Name"]
    m = re.search(r"(k_\w+|rocclr\w+|elementwise)", name)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r.get("Stream_Id", "?"):>3} {(m.group(1) if m else name[:30]):<26}{(s - t0) / 1e3:12.1f}{(e - s) / 1e3:10.1f}')
