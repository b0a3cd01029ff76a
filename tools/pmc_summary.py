#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc output (counter_collection.csv files below the given
directories) as mean counter value per launch of one kernel.
Usage: python3 tools/pmc_summary.py <kernel-name-substring> <out.json> <dir> [<dir> ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

kernel, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))   # counter -> dispatch id -> value (summed over XCDs / SEs)
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel not in row["Kernel_Name"]:
                    continue
                acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
res = {}
for c, per in sorted(acc.items()):
    vals = list(per.values())
    res[c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
res["_kernel"] = kernel
with open(out, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res, indent=1))
