#!/usr/bin/env python3
"""PCIe- and file-inclusive rate of apps/packet_receiver_file.py: writes a burst stream with valid
headers (bench.burst_stream) to a raw complex64 file under /tmp and receives it.
    tools/file_rate.py [log2_items=27] [chunk_log2=24]"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

spec = importlib.util.spec_from_file_location("packet_receiver_file", os.path.join(ROOT, "apps", "packet_receiver_file.py"))
app = importlib.util.module_from_spec(spec)
spec.loader.exec_module(app)
pkg = ge.load_package()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
chunk = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
rrc = bench.unit_norm_rrc(pkg)
x, n_pkt = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"), header=bench.header_symbols(1500))
path = "/tmp/gr4pm_iq.c64"
x.cpu().numpy().astype("<c8").tofile(path)
del x
for rep in range(2):   # the second pass reads the file from the page cache
    r = app.receive_file(path, chunk_items=chunk, pkg=pkg)
    print(f"pass {rep}: {r['items']} samples in {r['seconds']:.3f} s = {r['items'] / r['seconds'] / 1e6:.0f} Msps; "
          f"headers {r['headers']} ({r['invalid_headers']} invalid), packets {len(r['packets'])}, "
          f"CRC failures {r['crc_failures']} (the generator's payloads are random bits: no valid CRC)")
os.remove(path)
