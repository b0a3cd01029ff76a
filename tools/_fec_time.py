import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
hs = bench.header_symbols(1500)
# LLRs of a valid, still scrambled header -> descramble by hand: use the decoder on unscrambled bits instead
gen = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/header_ldpc_generator.npy"))
info = (1500 >> 8 << 24) | ((1500 & 255) << 16) | 0x55
bits = [(info >> (31 - i)) & 1 for i in range(32)] + [bin(info & int(g)).count("1") & 1 for g in gen]
bits = np.array(bits + bits, dtype=np.float32)
dec = pkg.HeaderFecDecoder()
for n in (250, 2500, 25000):
    llr = torch.from_numpy(np.tile(1.0 - 2.0 * bits, n)).cuda()
    noisy = llr + 0.9 * torch.randn_like(llr)
    for name, t in (("clean", llr), ("noisy", noisy)):
        dec.process_bulk(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out, inv = dec.process_bulk(t)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(n, name, f"{dt*1e6:.0f} us/call", "invalid", int(inv.sum()))
