#!/usr/bin/env python3
"""BASELINE configs[4] ("stress": 1 channel, 1024-tap RRC, 4096-point overlap-save FFT): rate of
SyncwordDetection on its generic-size path (k_correlate_generic: one 256-thread workgroup per
block, radix-2 FFT in LDS), and of the 1025-tap InterpolatingFirFilter that shapes such a signal.
tools/bench_config5.py [items] [reps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sps = 4
rrc = pkg.root_raised_cosine(1.0, float(sps), 1.0, 0.35, 1024)
rrc = (rrc / np.sqrt(np.sum(rrc.astype(np.float64) ** 2))).astype(np.float32)
L = 63 * sps + rrc.size
print(f"taps {rrc.size}, syncword samples {L}, stride {4096 - L + 1}")
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.view_as_complex(0.3 * torch.randn((n, 2), device="cuda", generator=g))
bpsk = np.array([1, -1], dtype=np.complex64)
for bins in (0, 4):
    sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -bins, bins, fft_size=4096, power_threshold=30.0, max_items=n)  # no detections on noise: the correlator path is what is timed
    for _ in range(2):
        sd.process_bulk(x, want_output=False, tags_cap=1 << 16)
    torch.cuda.synchronize(); t0 = time.perf_counter(); done = 0
    for _ in range(reps):
        st, _, tags, nd = sd.process_bulk(x, want_output=False, tags_cap=1 << 16)
        done += nd
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"SyncwordDetection N=4096, {2 * bins + 1} bin(s): {done / dt / 1e6:9.1f} Msps  ({dt / reps * 1e3:.2f} ms per {n} items; "
          f"{done / dt * 8 / 1e9:.0f} GB/s read)")
    del sd
# the transmit-side shaping filter of this configuration: 1025 taps, interpolation 4
m = n // 16
sym = torch.view_as_complex(torch.randn((m, 2), device="cuda", generator=g))
fir = pkg.InterpolatingFirFilter(sps, rrc)
for _ in range(2):
    fir.process_bulk(sym)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    y = fir.process_bulk(sym)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"InterpolatingFirFilter 1025 taps x4: {reps * y.numel() / dt / 1e6:9.1f} Msps out")
