#!/usr/bin/env python3
"""wall time of each PacketReceiver stage, run back to back and pipelined"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = 1 << 26
rrc = bench.unit_norm_rrc(pkg)
x, n_pkt = bench.burst_stream(pkg, n, rrc, 1, torch.device("cuda"))
H = 1537
ring = torch.empty(H + 1 + n, dtype=torch.complex64, device="cuda")
ring[1:1 + H] = x[-H:]; ring[1 + H:] = x
x = ring[1 + H:]; hist = ring[1:1 + H]
rx = pkg.PacketReceiver(max_items=n, pipelined=False)
cap = 2 * n_pkt + 64
for _ in range(2):
    rx.process_bulk(x, 1500, tags_cap=cap, history=hist)
T = [0.0, 0.0, 0.0]
reps = 10
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    front = rx._stage0(x, cap, hist)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    res = rx._stage1(*front, 1500)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = rx._stage2(res)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2
print("stage wall ms (back to back):", [round(t / reps * 1e3, 3) for t in T], "sum", round(sum(T) / reps * 1e3, 3))

# finer: stage 1 pieces
import numpy as np
st, y, det_tags, nn, base = rx._stage0(x, cap, hist)
torch.cuda.synchronize()
tt = {}
def tic(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(5):
    st, y, det_tags, nn, base = rx._stage0(x, cap, hist)
    t0 = tic()
    headers = np.full(det_tags.size, 1500, dtype=np.uint64)
    acc, _ = rx.syncword_detection_filter.gate(base + det_tags["index"], headers, per_tag=True)
    tags = det_tags[acc]
    t1 = tic()
    sym, sym_tags, consumed = pkg.cfc_symbol_filter(rx.freq_correction, rx.symbol_filter, y, tags)
    t2 = tic()
    w = rx.syncword_wipeoff.process_bulk(sym, sym_tags)
    t3 = tic()
    c = rx.costas_loop.process_bulk(w, sym_tags)
    t4 = tic()
print("gate %.3f  cfc+symf %.3f  wipeoff %.3f  costas %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))

# the soft-bit tail, call by call
rs = pkg.PacketReceiver(max_items=n, pipelined=False, soft_bits=True)
for rep in range(4):
    front = rs._stage0(x, cap, hist)
    r1 = rs._stage1(*front, 1500)
    t0 = tic()
    pm = rs.payload_metadata_insert.process_bulk(r1["symbols"], r1["tags"], r1["headers"], per_tag=True)
    t1 = tic()
    z = rs.costas_loop.process_packets(pm["out"], pm["tags"])
    t2 = tic()
    data, data_tags = rs.syncword_remove.process_bulk(z, pm["tags"])
    t3 = tic()
    llr, llr_tags = rs.constellation_decoder.process_bulk(data, data_tags)
    t4 = tic()
print("soft-bit tail: PayloadMetadataInsert %.3f  Costas(tags) %.3f  SyncwordRemove %.3f  LLR %.3f ms"
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
