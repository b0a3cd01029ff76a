#!/usr/bin/env python3
"""k_correlate_w64 alone against the blocks a workgroup's waves share (GR4PM_W64_BLOCKS_PER_WAVE; 0 = persistent waves),
interleaved rounds in one process.  tools/r5_bpw.py [items] [bins] [rounds] [v,v,...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["6", "0", "4", "5", "7", "8", "12"]
rrc = bench.unit_norm_rrc(pkg)
x, _ = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
bpsk = np.array([1, -1], dtype=np.complex64)
sds = {}
for v in variants:
    os.environ["GR4PM_W64_BLOCKS_PER_WAVE"] = v
    os.environ["GR4PM_W64_ONE"] = os.environ.get("R5_ONE", "0")
    sds[v] = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -bins, bins, power_threshold=9.5, max_items=n)
    sds[v].correlate_only(x)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            sds[v].correlate_only(x)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 3)
for v in variants:
    t = np.array(times[v])
    print(f"blocks per wave {v:>3}: median {np.median(t):.4f} ms  min {t.min():.4f} ms")
