#!/usr/bin/env python3
"""What a few hundred long-running one-wave workgroups cost the correlator: k_correlate_w64 timed alone and while
k_costas (one lane per packet, 157 waves, ~1 ms) runs back to back on another stream.
tools/coexec.py [items]"""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
dev = torch.device("cuda")
rrc = bench.unit_norm_rrc(pkg)
x, n_pkt = bench.burst_stream(pkg, n, rrc, 1, dev)
bpsk = np.array([1, -1], dtype=np.complex64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -4, 4, power_threshold=9.5, max_items=n)
n_sym = n // 4
sym = torch.view_as_complex(torch.randn((n_sym, 2), device=dev))
tags = np.zeros(n_pkt, dtype=pkg.TAG_DTYPE)
tags["index"] = np.arange(n_pkt) * (n_sym // n_pkt)
tags["flags"] = pkg.TAG_SYNCWORD
with torch.cuda.stream(s2):
    costas = pkg.CostasLoop(0.01, "QPSK")
torch.cuda.synchronize()

def time_corr(reps=6):
    with torch.cuda.stream(s1):
        sd.correlate_only(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            sd.correlate_only(x)
        e1.record()
        e1.synchronize()
    return e0.elapsed_time(e1) / reps

print(f"{n_pkt} packets -> {(n_pkt + 63) // 64} Costas waves; correlator alone: {time_corr():.3f} ms")
t0 = time.perf_counter()
costas.process_bulk(sym, tags)
print(f"Costas alone (one call incl. host planning and sync): {(time.perf_counter() - t0) * 1e3:.3f} ms")
stop, calls = False, [0]
def spin():
    with torch.cuda.stream(s2):
        while not stop:
            costas.process_bulk(sym, tags)
            calls[0] += 1
th = threading.Thread(target=spin)
th.start()
time.sleep(0.05)
c0, t0 = calls[0], time.perf_counter()
t = time_corr(10)
dt, dc = time.perf_counter() - t0, calls[0] - c0
stop = True
th.join()
print(f"correlator while Costas runs back to back: {t:.3f} ms   ({dc} Costas calls in {dt * 1e3:.1f} ms = {dt / max(dc, 1) * 1e3:.2f} ms each)")
