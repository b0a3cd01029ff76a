#!/usr/bin/env python3
"""HBM bytes per launch of every kernel of a bench.py step, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
counter_collection.csv below the two directories).  read = 2 x FETCH_SIZE KiB (the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md), write = WRITE_SIZE KiB; mean of the middle half of each kernel's launches.
Usage: python3 tools/pmc_other_kernels.py <fetch-dir> <write-dir> <out.json>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                m = re.search(r"(k_\w+)", row["Kernel_Name"])
                if m:
                    acc[m.group(1)][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = {}
    for k, per in acc.items():
        v = sorted(per.values())
        mid = v[len(v) // 4: len(v) - len(v) // 4] or v
        out[k] = sum(mid) / len(mid)
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    r, w = 2.0 * fetch.get(k, 0.0) * 1024 / 1e6, write.get(k, 0.0) * 1024 / 1e6
    if r + w >= 5.0:
        res[k] = {"read_MB": round(r, 1), "write_MB": round(w, 1)}
with open(sys.argv[3], "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res, indent=1))
