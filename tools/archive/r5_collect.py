#!/usr/bin/env python3
"""Copies what tools/r5_profiles.sh left under gpurun_out/r5_final/ into profiles/r5_* (the tracked evidence)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/archive/ -> the repository root)
O = os.path.join(ROOT, "gpurun_out", "r5_final")
P = os.path.join(ROOT, "profiles")
pairs = [("bench.json", "r5_bench.json"), ("kernel_stats_pipe.csv", "r5_kernel_stats_pipelined.csv"),
         ("kernel_stats_one.csv", "r5_kernel_stats_one_stream.csv"),
         ("kernel_stats_parity.csv", "r5_kernel_stats_headline_parity_test.csv"),
         ("kernels_hbm_traffic.json", "r5_other_kernels_hbm_traffic.json"),
         ("chain_valu_one_stream.json", "r5_chain_valu_per_kernel.json"),
         ("chain_valu_one_stream.txt", "r5_chain_valu_per_kernel.txt"), ("ab.txt", "r5_ab.txt"), ("tools.txt", "r5_tools.txt")]
for src, dst in pairs:
    s = os.path.join(O, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, dst))
    else:
        print("missing", src)
for tag, name, bins, desc in (("r5_final/pmc_corr9", "r5_k_correlate", 4, "k_correlate_w64 (instantiation <114688>, round 5: power stores without a compare / branch per store)"),
                              ("r5_final/pmc_corr1", "r5_k_correlate_1bin", 0, "k_correlate_w64_one (one frequency bin: the default of a stand-alone SyncwordDetection since round 5)")):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", tag, "summary.json")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_profiles.py"), tag, name, "67108864", str(bins), desc])
    else:
        print("missing", tag)
st = os.path.join(O, "pmc_corr9", "stats")
for dirpath, _, files in os.walk(st):
    for f in files:
        if f.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(dirpath, f), os.path.join(P, "r5_k_correlate_kernel_stats.csv"))
print(sorted(f for f in os.listdir(P) if f.startswith("r5_")))
