#!/usr/bin/env python3
"""configs[4] detector alone (fft_size 4096, 1025-tap RRC) over 2^28 samples in one call, a few times: the target of
`rocprofv3 --kernel-trace --stats` for its per-kernel times.  tools/config5_kstats.py [bins=4] [items=2^28] [calls=4]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge, bench
pkg = ge.load_package()
bins = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 28
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
x, rrc, _ = bench.config5_stream(pkg, n, torch.device("cuda"))
sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, np.array([1, -1], np.complex64), -bins, bins, fft_size=4096,
                           power_threshold=30.0 if bins else 60.0, max_items=n)
sd.process_bulk(x, want_output=False, tags_cap=1 << 17)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    sd.reset()
    _, _, tags, nd = sd.process_bulk(x, want_output=False, tags_cap=1 << 17)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print(f"SyncwordDetection fft 4096, {2 * bins + 1} bins: {dt * 1e3:.3f} ms per {n} samples ({n / dt / 1e9:.1f} Gsps), {tags.size} tags")
