#!/usr/bin/env python3
"""Times the variants of k_correlate_w64 (GR4PM_W64_VARIANT) and the round-1 kernel, interleaved rounds in
one process (cdna_hip_programming.md rule 24).  tools/w64_variants.py [items] [bins] [rounds] [v,v,...]"""
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
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["wave", "0", "8", "32", "232", "1256", "2048", "4096", "6144"]
rrc = bench.unit_norm_rrc(pkg)
x, _ = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
bpsk = np.array([1, -1], dtype=np.complex64)
sds = {}
for v in variants:
    if v in ("wave", "pair"):
        os.environ["GR4PM_CORRELATOR"] = v
    else:
        os.environ["GR4PM_CORRELATOR"] = "w64"
        os.environ["GR4PM_W64_VARIANT"] = v
    sds[v] = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -bins, bins, power_threshold=9.5, max_items=n)
    try:
        st, _, tags, nd = sds[v].process_bulk(x, want_output=False, tags_cap=1 << 17)
    except Exception as e:  # timing-only ablations produce garbage
        print(f"variant {v:>5}: {e}")
        sds[v] = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -bins, bins, power_threshold=9.5, max_items=n)
        sds[v].correlate_only(x)
        continue
    z = sds[v].last_zpow(nd)
    if v == variants[0]:
        zref, tref = z, tags
    else:
        print(f"variant {v:>5}: max |dz| / full scale vs {variants[0]} = {float((z - zref).abs().nan_to_num(1e30, 1e30, 1e30).max() / zref.max()):.3e}, "
              f"tags {tags.size} (same indices: {bool(np.array_equal(tags['index'], tref['index']))})")
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
    print(f"variant {v:>5}: median {np.median(t):.4f} ms  min {t.min():.4f} ms   ({n / np.median(t) / 1e3:.0f} Msps)")
