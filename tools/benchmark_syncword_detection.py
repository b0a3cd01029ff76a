#!/usr/bin/env python3
"""MI355X counterpart of the reference's benchmarks/benchmark_syncword_detection.cpp:
NullSource (zeros) -> SyncwordDetection -> ProbeRate, same positional arguments

    benchmark_syncword_detection.py [syncword_freq_bins=4] [syncword_threshold=9.5] [items_per_call=2^26]

and the same report (ProbeRate's rate_now / rate_avg, probe_rate.hpp:60-69), here once per
second of wall time for ~5 s.  Every item is a candidate on an all-zero stream (zpow == 0
everywhere): the densest case for the detector kernels.  Reference (Ryzen 7 5800X): 13 Msps at
4 bins, 50 Msps at 0 (benchmarks/results.md:37-41)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 4           # benchmark_syncword_detection.cpp:25
threshold = float(sys.argv[2]) if len(sys.argv) > 2 else 9.5  # :26
items = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 26
pkg = ge.load_package()
rrc = bench.unit_norm_rrc(pkg)                                 # :48-62
x = torch.zeros(items, dtype=torch.complex64, device="cuda")   # NullSource, null_source.hpp:25
sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, np.array([1, -1], np.complex64), -bins, bins,
                           power_threshold=threshold, max_items=items)
for _ in range(2):
    sd.process_bulk(x)
torch.cuda.synchronize()
t_start = t_last = time.perf_counter()
count = last_count = 0
rate_avg = None
while time.perf_counter() - t_start < 5.0:
    st, out, tags, n = sd.process_bulk(x)
    assert tags.size == 0
    count += n
    now = time.perf_counter()
    if now - t_last >= 1.0:
        rate_now = (count - last_count) / (now - t_last)
        rate_avg = rate_now if rate_avg is None else 0.15 * rate_now + 0.85 * rate_avg  # probe_rate.hpp:98-99
        print(f"rate_now = {rate_now:.4e} rate_avg = {rate_avg:.4e}")
        t_last, last_count = now, count
torch.cuda.synchronize()
dt = time.perf_counter() - t_start
print(f"zeros: bins={2*bins+1} {count/dt/1e6:.1f} Msps over {dt:.1f} s")
