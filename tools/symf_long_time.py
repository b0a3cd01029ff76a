#!/usr/bin/env python3
"""SymbolFilter with 32 arms x 1025 taps (BASELINE configs[4]) on 2^26 samples of the configs[4] stream: ms per call and
Gsps in (k_symbol_filter_long; GR4PM_SYMF_GENERIC=1 times the generic kernel).  tools/symf_long_time.py [items]"""
import sys, time, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge, bench
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
x, rrc, _ = bench.config5_stream(pkg, n, torch.device("cuda"))
pfb = pkg.root_raised_cosine(32.0, 32.0 * 4, 1.0, 0.35, 32 * 1024)[: 32 * 1025]
sf = pkg.SymbolFilter(pfb, 32, 4, delay=1025)
for _ in range(2):
    sf.process_bulk(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    sf.process_bulk(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("SymbolFilter 32 x 1025 taps: %.3f ms per %d samples, %.1f Gsps in, %.1f TFLOP/s (1025 FLOP per sample)" % (dt * 1e3, n, n / dt / 1e9, 1025 * n / dt / 1e12))
