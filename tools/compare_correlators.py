#!/usr/bin/env python3
"""two correlator kernels (GR4PM_CORRELATOR = w64 / wave / pair) on the same input: same powers? how fast?
tools/compare_correlators.py [items] [reps] [bins] [kindA,kindB]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rrc = bench.unit_norm_rrc(pkg)
x, _ = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
bpsk = np.array([1, -1], dtype=np.complex64)
z = {}
for kind in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("wave", "w64")):
    os.environ["GR4PM_CORRELATOR"] = kind
    sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -bins, bins, power_threshold=9.5, max_items=n)
    st, out, tags, nd = sd.process_bulk(x, want_output=False, tags_cap=1 << 16)
    z[kind] = sd.last_zpow(nd).cpu().numpy()
    for _ in range(2):
        sd.correlate_only(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sd.correlate_only(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{kind}: {dt * 1e3:.4f} ms/launch  {n / dt / 1e6:.1f} Msps  tags {tags.size}")
kinds = list(z)
a, b = z[kinds[0]], z[kinds[1]]
print("full scale", float(a.max()), " max abs difference / full scale:", float(np.max(np.abs(a - b)) / a.max()))
same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
rel = np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-30))
print("identical bits:", same, " max relative difference:", rel, " differing:", int(np.sum(a != b)), "of", a.size)
