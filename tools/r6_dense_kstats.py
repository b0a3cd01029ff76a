#!/usr/bin/env python3
"""1500-byte packets back to back (500-symbol gaps: bench.py's `sparse.dense_packets` stream) through the whole receiver
(decode_headers: IQ in, CRC-checked packets out), a few passes of 2^28 samples.  Under rocprofv3 --kernel-trace --stats:
the kernels of the symbol-rate tail (tools/r6_dense_kstats.sh -> profiles/r6_kernel_stats_decode_headers.csv); with a
GR4PM_TIMING build: the wall time of every stage.  R6_LEAN=1: packets_only receiver."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
SPS, BINS = 4, 4
n = 1 << 28
if os.environ.get("R6_CHURN"):
    # what a long-lived process looks like to the device allocator: GiB-sized blocks allocated and freed out of order
    rng0 = np.random.default_rng(1)
    blocks = [torch.empty(int(rng0.integers(1 << 26, 1 << 31)), dtype=torch.uint8, device="cuda") for _ in range(48)]
    for i in rng0.permutation(48)[:30]:
        blocks[i] = None
    torch.cuda.empty_cache()
    small = [torch.empty(int(rng0.integers(1 << 20, 1 << 27)), dtype=torch.uint8, device="cuda") for _ in range(200)]
    del small
    keep = [b for b in blocks if b is not None]
    if os.environ["R6_CHURN"] == "2":
        del keep, blocks
        torch.cuda.empty_cache()
gen = pkg.BurstGenerator()
burst = (64 + 128 + 1504 * 4 + gen.RAMP_DOWN + gen.FLUSH) * SPS
period = burst + 500 * SPS
n_pkt = n // period
rng = np.random.default_rng(5)
payloads = [rng.integers(0, 256, 1500, dtype=np.uint8).tobytes() for _ in range(n_pkt)]
x = gen.stream(payloads, np.full(n_pkt, period - burst), freq_error=0.01, esn0_db=20.0, seed=6, tail=0, carrier="closed_form")
if x.numel() < n:
    x = torch.cat([x, torch.zeros(n - x.numel(), dtype=x.dtype, device=x.device)])
x = x[:n].contiguous()
hist = 2 * 768 + 1
ring = torch.empty(hist + 1 + n, dtype=torch.complex64, device="cuda")
ring[1:1 + hist] = x[-hist:]
ring[1 + hist:] = x
w, history = ring[1 + hist:], ring[1:1 + hist]
del x
kw = {"packets_only": True} if os.environ.get("R6_LEAN", "0") != "0" else {}
if os.environ.get("R6_FIELDS"):
    kw["result_fields"] = os.environ["R6_FIELDS"]
rx = pkg.NativePacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=n, tags_cap=4 * n_pkt, pipelined=True, decode_headers=True,
                              output_ring=True, **kw)


T = {"announce": 0.0, "submit": 0.0, "collect": 0.0, "python": 0.0}


def run(k):
    """the caller's thread, piece by piece (R6_CALLER_TIMES=1 prints where it spends a pass)"""
    ok = 0
    depth = 5
    for i in range(k):
        t0 = time.perf_counter()
        if i + 1 < k:
            rx.announce(w)
        t1 = time.perf_counter()
        rx.submit(w, None, history, None)
        t2 = time.perf_counter()
        r = rx.collect() if pkg.lib().gr4pm_packet_receiver_inflight(rx._h) > depth else None
        t3 = time.perf_counter()
        if r is not None:
            ok += int(np.sum(r["packet_lengths"] > 0))
        t4 = time.perf_counter()
        T["announce"] += t1 - t0
        T["submit"] += t2 - t1
        T["collect"] += t3 - t2
        T["python"] += t4 - t3
    for r in rx.flush():
        ok += int(np.sum(r["packet_lengths"] > 0))
    return ok


run(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ok = run(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if os.environ.get("R6_CALLER_TIMES"):
    print("caller thread, ms per pass:", {a: round(b / (k + 2) * 1e3, 3) for a, b in T.items()})
print(f"{k} passes: {dt / k * 1e3:.3f} ms per 2^28, {k * n / dt / 1e9:.2f} Gsps, {ok} of {k * n_pkt} packets")
