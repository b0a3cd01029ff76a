import os, sys, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n = 1 << 26
rrc = bench.unit_norm_rrc(pkg)
dev = torch.device("cuda")
hs = bench.header_symbols(1500)
xa, n_pkt = bench.burst_stream(pkg, n, rrc, 1, dev, header=hs)
rx = pkg.PacketReceiver(max_items=n, decode_headers=True)
cap = 2 * n_pkt + 64
for _ in range(2):
    rx.process_bulk(xa, tags_cap=cap)
def run():
    for _ in range(5):
        rx.process_bulk(xa, tags_cap=cap)
    torch.cuda.synchronize()
cProfile.run("run()", "/tmp/prof.out")
pstats.Stats("/tmp/prof.out").sort_stats("tottime").print_stats(22)
