#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/summary.json (tools/pmc_correlate.sh) into the two committed files
profiles/<name>_pmc.json and profiles/<name>_hbm_traffic.json.
Usage: python3 tools/pmc_to_profiles.py <tag> <name> <items> <bins> "<kernel description>" """
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, items, bins, desc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
fft, L = 2048, 297
S = fft - L + 1
samples = ((items - fft) // S + 1) * S
d = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "summary.json")))
flat = {k: v["mean_per_launch"] for k, v in d.items() if isinstance(v, dict)}
fetch_kib, write_kib = flat.pop("FETCH_SIZE"), flat.pop("WRITE_SIZE")
read_b = 2.0 * fetch_kib * 1024.0   # gfx950: FETCH_SIZE reports half of a wide streaming read (MI355X_MICROARCH.md, HBM)
write_b = write_kib * 1024.0
traffic = {
    "kernel": desc, "bins": 2 * bins + 1, "samples_per_launch": samples,
    "FETCH_SIZE_mean_KiB": fetch_kib, "WRITE_SIZE_mean_KiB": write_kib,
    "hbm_read_bytes_corrected": read_b, "hbm_write_bytes": write_b,
    "traffic_bytes_per_launch": read_b + write_b,
    "traffic_bytes_per_sample": (read_b + write_b) / samples,
    "algorithmic_read_bytes": 8 * samples,
    "command": f"bash tools/pmc_correlate.sh {tag} {items} {bins}  (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, "
               "separate passes, python3 tools/bench_correlate.py)",
}
flat["_note"] = (f"mean per launch of {desc}; {samples} samples, {2 * bins + 1} bins; bash tools/pmc_correlate.sh {tag} {items} {bins} "
                 "(rocprofv3 --kernel-trace --pmc, three SQ passes)")
json.dump(flat, open(os.path.join(ROOT, "profiles", f"{name}_pmc.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{name}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
