#!/usr/bin/env python3
"""Times / exercises only the overlap-save correlator kernel (k_correlate) -- the target of
rocprofv3 --pmc passes.  Usage: python3 tools/bench_correlate.py [items] [reps] [bins]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

items = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pkg = ge.load_package()
rrc = bench.unit_norm_rrc(pkg)
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.complex(torch.randn(items, generator=g, device="cuda"), torch.randn(items, generator=g, device="cuda"))
sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, np.array([1, -1], np.complex64), -bins, bins, max_items=items)
for _ in range(int(os.environ.get("WARM", "30"))):  # (clocks: the first launches after host work run slower)
    sd.correlate_only(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    sd.correlate_only(x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nb = (items - 2048) // 1752 + 1
print(f"k_correlate bins={2*bins+1} items={items} {ms:.4f} ms/launch  {nb*1752/ms/1e3:.1f} Msps  "
      f"{8*nb*1752/ms/1e6:.1f} GB/s(read)")
