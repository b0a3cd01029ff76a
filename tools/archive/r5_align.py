#!/usr/bin/env python3
"""k_correlate_w64 alone, nine bins, 2^28 samples: does the alignment of the input window (a ring view 16 bytes off a
256-byte boundary, as bench.py's headline ring had it until round 5, against 256-byte aligned) or the stream matter?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = 1 << 28
rrc = bench.unit_norm_rrc(pkg)
x, _ = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
buf = torch.empty(n + 4096, dtype=torch.complex64, device="cuda")
bpsk = np.array([1, -1], dtype=np.complex64)
views = {}
for off in (0, 2, 8, 16, 1538, 1568):
    v = buf[off:off + n]
    v.copy_(x)
    views[off] = v
side = torch.cuda.Stream()
res = {}
for rounds in range(4):
    for off, v in views.items():
        for sname, stream in (("default", torch.cuda.current_stream()), ("side", side)):
            with torch.cuda.stream(stream):
                key = (off, sname)
                if key not in res:
                    res[key] = (pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, -4, 4, power_threshold=9.5, max_items=n), [])
                sd, ts = res[key]
                sd.correlate_only(v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    sd.correlate_only(v)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
for (off, sname), (_, ts) in res.items():
    print(f"offset {off:5d} items ({(off * 8) % 256:3d} B past a 256-B boundary), {sname:7s} stream: median {np.median(ts):.4f} ms  min {min(ts):.4f}")
