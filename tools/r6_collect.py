#!/usr/bin/env python3
"""Copies what tools/r6_profiles.sh left under gpurun_out/r6_final/ into profiles/r6_* (the tracked evidence)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r6_final")
P = os.path.join(ROOT, "profiles")
pairs = [("bench.json", "r6_bench.json"), ("kernel_stats_pipe.csv", "r6_kernel_stats_pipelined.csv"),
         ("kernel_stats_one.csv", "r6_kernel_stats_one_stream.csv"),
         ("kernel_stats_dense_lean.csv", "r6_kernel_stats_decode_headers.csv"),
         ("kernel_stats_dense_full.csv", "r6_kernel_stats_decode_headers_full_form.csv"),
         ("kernels_hbm_traffic.json", "r6_other_kernels_hbm_traffic.json"),
         ("overlap_save_pattern.txt", "r6_overlap_save_pattern.txt"), ("ab.txt", "r6_ab.txt"), ("tools.txt", "r6_tools.txt"),
         ("detector_tail.txt", "r6_detector_tail.txt"), ("headline_trace.txt", "r6_headline_trace.txt"),
         ("soak.txt", "r6_soak.txt"), ("tail_ab.txt", "r6_tail_ab.txt")]
for src, dst in pairs:
    s = os.path.join(O, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, dst))
    else:
        print("missing", src)
for tag, name, bins, desc in (("r6_final/pmc_corr9", "r6_k_correlate", 4, "k_correlate_w64 (instantiation <114688>, round 6: grid of whole rounds of one workgroup per CU)"),
                              ("r6_final/pmc_corr1", "r6_k_correlate_1bin", 0, "k_correlate_w64_one (one frequency bin)")):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", tag, "summary.json")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_profiles.py"), tag, name, "268435456", str(bins), desc])
    else:
        print("missing", tag)
    st = os.path.join(ROOT, "gpurun_out", tag, "stats")
    found = [os.path.join(dirpath, f) for dirpath, _, files in os.walk(st) for f in files if f.endswith("kernel_stats.csv")]
    if found:  # (gpurun MERGES its output into gpurun_out/: the files of earlier calls are still there -- the newest one)
        shutil.copy(max(found, key=os.path.getmtime), os.path.join(P, name + "_kernel_stats.csv"))
print(sorted(f for f in os.listdir(P) if f.startswith("r6_")))
