import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = 1 << 23
rrc = bench.unit_norm_rrc(pkg)
dev = torch.device("cuda")
hs = bench.header_symbols(1500)
xa, n_pkt = bench.burst_stream(pkg, n, rrc, 1, dev, header=hs)
xb, _ = bench.burst_stream(pkg, n, rrc, 1001, dev, header=hs)
rx = pkg.PacketReceiver(max_items=n, decode_headers=True)
for i in range(6):
    w = xa if i % 2 == 0 else xb
    res = rx.process_bulk(w)
    m = res["header_messages"]
    ok = m["invalid_header"] == 0
    bad = np.nonzero(~ok)[0]
    print(i, "det", res["detector_tags"].size, "acc", int(res["accepted"].sum()), "hdr", m.size, "valid", int(ok.sum()),
          "first bad", bad[:8], "last bad", bad[-3:], "pm tags", res["packet_tags"].size, "ign", res["ignored_syncwords"])
