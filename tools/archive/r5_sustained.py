#!/usr/bin/env python3
"""k_correlate_w64 alone, nine bins, 2^28 samples, launched back to back for ~12 s: the launch time over time (does the
chip hold its first seconds' clock under this kernel?)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = 1 << 28
rrc = bench.unit_norm_rrc(pkg)
x, _ = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
bpsk = np.array([1, -1], dtype=np.complex64)
sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -4, 4, power_threshold=9.5, max_items=n)
sd.correlate_only(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = []
while time.perf_counter() - t0 < 12.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        sd.correlate_only(x)
    e1.record()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 40))
print(" ".join(f"{t:.1f}s:{ms:.3f}" for t, ms in out[:: max(1, len(out) // 24)]))
time.sleep(3.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    sd.correlate_only(x)
e1.record()
torch.cuda.synchronize()
print(f"after 3 s of idling: {e0.elapsed_time(e1) / 10:.3f} ms")
