#!/usr/bin/env python3
"""SyncwordDetection on an all-zero stream: the input of the reference's own
benchmark_syncword_detection (NullSource, benchmarks/README.md:49-53).  Every item is a
candidate here (zpow == 0 everywhere), the densest case for the detector kernels."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

items = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pkg = ge.load_package()
rrc = bench.unit_norm_rrc(pkg)
x = torch.zeros(items, dtype=torch.complex64, device="cuda")
sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, np.array([1, -1], np.complex64), -bins, bins, max_items=items)
for _ in range(2):
    sd.process_bulk(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps, done = 10, 0
for _ in range(reps):
    st, out, tags, n = sd.process_bulk(x)
    done += n
    assert tags.size == 0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"zeros: bins={2*bins+1} {done/dt/1e6:.1f} Msps ({dt/reps*1e3:.3f} ms per {items} items); "
      f"reference publishes 13 Msps at 9 bins, 50 Msps at 1 bin (results.md:37-41)")
