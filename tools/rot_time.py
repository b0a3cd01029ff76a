#!/usr/bin/env python3
"""stand-alone Rotator on an untagged stream (ONE serial chain) and CoarseFrequencyCorrection with many set_freq tags:
ns per sample of the serial phasor chain.  tools/rot_time.py [items]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
x = torch.view_as_complex(torch.randn((n, 2), device="cuda"))
rot = pkg.Rotator(0.01)
rot.process_bulk(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    rot.process_bulk(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"Rotator, one untagged stream of {n} items: {dt * 1e3:.2f} ms = {dt / n * 1e9:.2f} ns per item ({n / dt / 1e6:.1f} Msps)")
