#!/usr/bin/env python3
"""Prints calls / average / min / max duration (us) per kernel from a rocprofv3 --kernel-trace --stats directory.
Usage: python3 tools/kstats.py <dir> [name-substring ...]"""
import csv, glob, os, re, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
if not fs:
    sys.exit("no kernel_stats.csv below " + sys.argv[1])
for r in csv.DictReader(open(fs[0])):
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"])
    name = m.group(1) if m else r["Name"][:50]
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print(f"{name:40s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f}  min {float(r['MinNs'])/1e3:10.1f}  max {float(r['MaxNs'])/1e3:10.1f} us")
