import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import __graft_entry__ as ge
import bench
import _oracle as orc
pkg = ge.load_package()
n = 1 << 24
rrc = bench.unit_norm_rrc(pkg)
dev = torch.device("cuda")
hs = bench.header_symbols(1500)
xa, n_pkt = bench.burst_stream(pkg, n, rrc, 1, dev, header=hs)
rx = pkg.PacketReceiver(max_items=n, soft_bits=True)
res = rx.process_bulk(xa, 1500)
hd = pkg.HeaderDecoder()
t = res["llr_tags"]
resets = t["index"][t["kind"] == pkg.PKT_HEADER_START]
d = hd.descrambler.process_bulk(res["llr"], resets)
hdr, pay, _, _ = hd.header_payload_split.process_bulk(d, t)
print("header llrs", hdr.numel() // 256)
h = hdr.cpu().numpy().reshape(-1, 256)
od = orc.HeaderFecDecoder(pkg.header_ldpc_alist())
its = [od.decode(r[:128] + r[128:])[1] for r in h[:300]]
print("iterations histogram", np.bincount(np.array(its) + 1))
print("llr abs mean", np.abs(h).mean(), "min abs", np.abs(h[:, :128] + h[:, 128:]).min())
dec = pkg.HeaderFecDecoder()
dec.process_bulk(hdr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    dec.process_bulk(hdr)
torch.cuda.synchronize(); print("decode", (time.perf_counter() - t0) / 5 * 1e6, "us for", hdr.numel() // 256)
