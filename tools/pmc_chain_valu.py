#!/usr/bin/env python3
"""Which kernels of a bench.py step consume the vector ALU: mean SQ_INSTS_VALU (wave-instructions) per launch and
launches per kernel from one rocprofv3 --pmc pass (counter_collection.csv below <dir>).  The chain is bound by the
vector ALU (the correlator keeps it 87 % busy by itself), so a kernel's share of this sum is what it costs the step.
Usage: python3 tools/pmc_chain_valu.py <dir> [<out.json>]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(k_\w+)", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"][:40]
            acc[name][row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
res = {}
for k, per in acc.items():
    e = {}
    for c, d in per.items():
        v = list(d.values())
        e[c] = {"mean_per_launch": round(sum(v) / len(v)), "launches": len(v), "total": round(sum(v))}
    res[k] = e
order = sorted(res, key=lambda k: -res[k].get("SQ_INSTS_VALU", {}).get("total", 0))
tot = sum(res[k].get("SQ_INSTS_VALU", {}).get("total", 0) for k in order) or 1
for k in order:
    e = res[k].get("SQ_INSTS_VALU")
    if e and e["total"] / tot >= 0.001:
        extra = " ".join(f"{c.replace('SQ_', '')}={res[k][c]['mean_per_launch']}" for c in sorted(res[k]) if c != "SQ_INSTS_VALU")
        print(f"{k:32s} launches {e['launches']:5d}  VALU/launch {e['mean_per_launch']:12d}  share {100 * e['total'] / tot:5.1f} %  {extra}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        json.dump({k: res[k] for k in order}, fh, indent=1)
