#!/usr/bin/env python3
"""where the main thread of the pipelined PacketReceiver spends a step: inside the detector's
C call, waiting for the oldest batch, or in Python glue (monkey-patched timers, MI355X)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
rrc = bench.unit_norm_rrc(pkg)
n = 1 << 26
dev = torch.device("cuda", 0)
xa, npkt = bench.burst_stream(pkg, n, rrc, 1, dev)
xb, _ = bench.burst_stream(pkg, n, rrc, 1001, dev)
H = 1537
ring = torch.empty(H + 1 + 2 * n, dtype=torch.complex64, device=dev)
ring[1:1 + H] = xb[-H:]
ring[1 + H:1 + H + n] = xa
ring[1 + H + n:] = xb
wins = [(ring[1 + H:1 + H + n], ring[1:1 + H]), (ring[1 + H + n:], ring[1 + n:1 + H + n])]
rx = pkg.PacketReceiver(4, 4, 9.5, "QPSK", max_items=n, pipelined=True)
T = {"c_call": 0.0, "stage0": 0.0, "wait": 0.0, "total": 0.0}
lib = pkg._abi.lib() if hasattr(pkg, "_abi") else None
sd = rx.syncword_detection
orig_pb = sd.process_bulk


def timed_pb(*a, **k):
    t = time.perf_counter()
    r = orig_pb(*a, **k)
    T["stage0"] += time.perf_counter() - t
    return r


sd.process_bulk = timed_pb
steps = 30
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize()
        for k in T:
            T[k] = 0.0
        t_all = time.perf_counter()
    w, h = wins[i % 2]
    t0 = time.perf_counter()
    front = rx._stage0(w, 2 * npkt + 64, h, wins[(i + 1) % 2][0])
    f1 = rx._workers[0].submit(rx._stage1, *front, 1500)
    f2 = rx._workers[1].submit(rx._stage12, f1)
    rx._inflight.append(f2)
    t1 = time.perf_counter()
    if len(rx._inflight) > 2:
        rx._inflight.pop(0).result()
    t2 = time.perf_counter()
    T["wait"] += t2 - t1
    T["total"] += t2 - t0
rx.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print(f"per step: wall {dt / steps * 1e3:.3f} ms | SyncwordDetection.process_bulk {T['stage0'] / steps * 1e3:.3f} ms | "
      f"wait for oldest batch {T['wait'] / steps * 1e3:.3f} ms | loop body {T['total'] / steps * 1e3:.3f} ms")
