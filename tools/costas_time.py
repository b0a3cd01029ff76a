#!/usr/bin/env python3
"""stand-alone time of the Costas kernel's three forms (112 / 62 / 32 VGPRs) on the headline's segment shape: 2^26
symbols, a syncword_phase tag every 6600 symbols (one lane per packet).  tools/costas_time.py [log2_symbols=26]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.view_as_complex(torch.randn((n, 2), device="cuda", generator=g)).contiguous()
idx = np.arange(100, n - 64, 6600, dtype=np.uint64)
tags = np.zeros(idx.size, dtype=pkg.TAG_DTYPE)
tags["index"] = idx
tags["flags"] = pkg.TAG_SYNCWORD
tags["phase"] = np.random.default_rng(2).uniform(-3, 3, idx.size).astype(np.float32)
ref = None
for mode in (0, 1, 2):
    cl = pkg.CostasLoop(0.01, "QPSK")
    pkg.lib().gr4pm_costas_loop_set_small_footprint(cl._h, mode)
    y = cl.process_bulk(x, tags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(5):
        cl2 = cl
        e0.record()
        y = cl2.process_bulk(x, tags)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    if ref is None:
        ref = y.clone()
    same = bool(torch.equal(torch.view_as_real(y), torch.view_as_real(ref)))
    print(f"small_footprint {mode}: {min(ms):.3f} ms min, {sorted(ms)[2]:.3f} median per {n} symbols ({idx.size} segments); "
          f"same bits as mode 0: {same}")
