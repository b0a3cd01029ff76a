#!/usr/bin/env python3
"""MI355X counterpart of the reference's benchmarks/benchmark_packet_receiver.cpp:20-73:
NullSource (zeros) -> the whole PacketReceiver (syncword detection ... CRC check) -> NullSink, ProbeRate on the source,
positional arguments as there:

    benchmark_packet_receiver.py [syncword_freq_bins=4 | all] [syncword_threshold=9.5] [items_per_batch=2^26] [seconds=3]

`all` walks syncword_freq_bins 0 .. 4, the rows of benchmarks/results.md:45-51 (28-32 / 16-18 / 10-13 / 8-10 / 6-8 Msps on a
Ryzen 7 5800X, multi-threaded scheduler), and ends with one JSON line.  The receiver is the native pipelined composition
with decode_headers (gr4pm_packet_receiver: header loop on the device, no constant packet length); on an all-zero
stream nothing is ever detected -- as in the reference's benchmark -- so the number is the front part of the chain:
every item is a candidate (zpow == 0 everywhere), the densest case for the detector.  Rate = ProbeRate's definition
(items / elapsed, probe_rate.hpp:60-69), rate_avg its 0.15 / 0.85 smoothing (:98-99)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

arg_bins = sys.argv[1] if len(sys.argv) > 1 else "4"                # benchmark_packet_receiver.cpp:25
threshold = float(sys.argv[2]) if len(sys.argv) > 2 else 9.5        # :26
items = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 26
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
pkg = ge.load_package()
# two windows of one zero ring, each preceded by the 2 T + 1 items "before" it: the receiver reads the detector's
# delayed stream in place, as bench.py does
HIST = 2 * 768 + 1
ring = torch.zeros(HIST + 1 + 2 * items, dtype=torch.complex64, device="cuda")  # NullSource, null_source.hpp:25
windows = [(ring[1 + HIST:1 + HIST + items], ring[1:1 + HIST]),
           (ring[1 + HIST + items:], ring[1 + items:1 + HIST + items])]
rows = {}
for bins in (range(5) if arg_bins == "all" else [int(arg_bins)]):
    rx = pkg.NativePacketReceiver(4, bins, threshold, "QPSK", max_items=items, tags_cap=4096, pipelined=True,
                                  decode_headers=True, output_ring=True)
    k = 0

    def step():
        global k
        w, hist = windows[k % 2]
        rx.announce(windows[(k + 1) % 2][0])
        k += 1
        r = rx.process_bulk(w, None, history=hist)
        return 0 if r is None else r["consumed"]
    rx.announce(windows[0][0])
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t_start = t_last = time.perf_counter()
    count = last_count = 0
    rate_avg = None
    while time.perf_counter() - t_start < seconds:
        count += step()
        now = time.perf_counter()
        if now - t_last >= 1.0:
            rate_now = (count - last_count) / (now - t_last)
            rate_avg = rate_now if rate_avg is None else 0.15 * rate_now + 0.85 * rate_avg
            print(f"syncword_freq_bins = {bins}: rate_now = {rate_now:.4e} rate_avg = {rate_avg:.4e}")
            t_last, last_count = now, count
    for r in rx.flush():
        count += r["consumed"]
        assert r["tags"].size == 0 and r["n_packets"] == 0 if "n_packets" in r else True
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    rows[str(bins)] = round(count / dt / 1e6, 1)
    print(f"zeros -> PacketReceiver(decode_headers): syncword_freq_bins={bins} ({2 * bins + 1} templates) {rows[str(bins)]} Msps over {dt:.1f} s")
    del rx
print(json.dumps({"benchmark": "benchmark_packet_receiver (zeros -> whole receiver -> NullSink)", "unit": "Msamples/s",
                  "syncword_freq_bins": rows, "reference_ryzen_5800x_msps": {"0": "28-32", "1": "16-18", "2": "10-13", "3": "8-10", "4": "6-8"},
                  "items_per_batch": items}))
