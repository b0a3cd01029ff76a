#!/usr/bin/env python3
"""One-bin correlator: powers / tags of the run (saved to or compared with a file) and the launch time.
GR4PM_W64_ONE=0 python3 tools/one_bin_check.py save /tmp/z.npy ; python3 tools/one_bin_check.py cmp /tmp/z.npy
[items, default 2^26]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
mode, path = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 26
rrc = bench.unit_norm_rrc(pkg)
x, _ = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
bpsk = np.array([1, -1], dtype=np.complex64)
sd = pkg.SyncwordDetection(rrc, bench.SYNCWORD, bpsk, 0, 0, power_threshold=9.5, max_items=n)
st, _, tags, nd = sd.process_bulk(x, want_output=False, tags_cap=1 << 17)
z = sd.last_zpow(nd).cpu().numpy()
if mode == "save":
    np.save(path, z)
    np.save(path + ".tags.npy", tags["index"])
else:
    ref, rt = np.load(path), np.load(path + ".tags.npy")
    print("max |dz| / full scale", float(np.nanmax(np.abs(z - ref)) / np.nanmax(ref)), "identical", bool(np.array_equal(z, ref)),
          "tags", tags.size, "same indices", bool(np.array_equal(tags["index"], rt)))
sd.correlate_only(x)
torch.cuda.synchronize()
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        sd.correlate_only(x)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
t = np.array(ts)
print(f"{mode}: W64_ONE={os.environ.get('GR4PM_W64_ONE', '1')} median {np.median(t):.4f} ms min {t.min():.4f} ms ({n / np.median(t) / 1e3:.0f} Msps)")
