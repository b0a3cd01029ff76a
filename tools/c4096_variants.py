#!/usr/bin/env python3
"""Times the bit-identical forms of k_correlate_4096 (GR4PM_C4096_VARIANT, correlate_4096.hpp) on the configs[4] stream,
interleaved rounds in one process.  tools/c4096_variants.py [items] [bins] [rounds] [v,v,...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else [str(v) for v in range(8)]
x, rrc, _ = bench.config5_stream(pkg, n, torch.device("cuda"))
bpsk = np.array([1, -1], dtype=np.complex64)
sds = {}
for v in variants:
    os.environ["GR4PM_C4096_VARIANT"] = v
    sds[v] = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -bins, bins, fft_size=4096, power_threshold=30.0, max_items=n)
    st, _, tags, nd = sds[v].process_bulk(x, want_output=False, tags_cap=1 << 17)
    z = sds[v].last_zpow(nd)
    if v == variants[0]:
        zref, tref = z, tags
    else:
        print(f"variant {v}: identical powers: {bool(torch.equal(z, zref))}, tags {tags.size} (same: {bool(np.array_equal(tags['index'], tref['index']))})")
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
    print(f"variant {v}: median {np.median(t):.4f} ms  min {t.min():.4f} ms   ({n / np.median(t) / 1e3:.0f} Msps) at {2 * bins + 1} bins, {n} items")
