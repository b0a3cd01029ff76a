#!/usr/bin/env python3
"""one 1500-byte packet per 2^20 samples through the whole receiver (bench.py's `sparse` stream), a few passes: run under
rocprofv3 --kernel-trace --stats to see which kernels a pass waits for (tools/r5_sparse_kstats.sh)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
SPS, BINS = 4, 4
n = 1 << 28
gen = pkg.BurstGenerator()
burst = (64 + 128 + 1504 * 4 + gen.RAMP_DOWN + gen.FLUSH) * SPS
period = 1 << 20
n_pkt = n // period
rng = np.random.default_rng(5)
payloads = [rng.integers(0, 256, 1500, dtype=np.uint8).tobytes() for _ in range(n_pkt)]
x = gen.stream(payloads, np.full(n_pkt, period - burst), freq_error=0.01, esn0_db=20.0, seed=6, tail=0, carrier="closed_form")[:n].contiguous()
hist = 2 * 768 + 1
ring = torch.empty(hist + 1 + n, dtype=torch.complex64, device="cuda")
ring[1:1 + hist] = x[-hist:]
ring[1 + hist:] = x
w, history = ring[1 + hist:], ring[1:1 + hist]
rx = pkg.NativePacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=n, tags_cap=4096, pipelined=True, decode_headers=True, output_ring=True)
import time
def run(k):
    ok = 0
    for i in range(k):
        if i + 1 < k:
            rx.announce(w)
        r = rx.process_bulk(w, None, history=history)
        if r is not None:
            ok += int(np.sum(r["packet_lengths"] > 0))
    for r in rx.flush():
        ok += int(np.sum(r["packet_lengths"] > 0))
    return ok
run(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ok = run(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{k} passes: {dt / k * 1e3:.3f} ms per 2^28, {k * n / dt / 1e9:.2f} Gsps, {ok} packets")
