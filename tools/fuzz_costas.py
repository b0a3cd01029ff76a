#!/usr/bin/env python3
"""Randomised differential test of CostasLoop against the CPU oracle: all three constellations, hundreds of
syncword_phase tags at ragged distances (segments of 1 .. 20 000 symbols, i.e. lanes of very different lengths in the
length-sorted launch), random call boundaries.  Every output symbol bit-exact.  tools/fuzz_costas.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import _oracle as orc
import test_gpu_parity as tp
pkg = ge.load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    const = ["PILOT", "BPSK", "QPSK"][case % 3]
    n = int(rng.integers(20000, 200000))
    a = np.float32(np.sqrt(0.5))
    if const == "QPSK":
        s = (np.where(rng.integers(0, 2, n) == 0, a, -a) + 1j * np.where(rng.integers(0, 2, n) == 0, a, -a))
    elif const == "BPSK":
        s = np.where(rng.integers(0, 2, n) == 0, 1.0, -1.0)
    else:
        s = np.ones(n)
    x = (s * np.exp(1j * (rng.uniform(-3, 3) + 0.003 * np.arange(n))) +
         rng.uniform(0.02, 0.3) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    gaps = np.concatenate([rng.integers(1, 30, int(rng.integers(0, 40))), rng.integers(30, 3000, int(rng.integers(3, 80))),
                           rng.integers(3000, 20000, int(rng.integers(1, 10)))])
    rng.shuffle(gaps)
    idx = np.cumsum(gaps)
    idx = np.concatenate([[0] if rng.integers(0, 2) else [], idx[idx < n]]).astype(np.uint64)
    ph = rng.uniform(-3.14, 3.14, idx.size).astype(np.float32)
    bw = float(rng.choice([0.005, 0.01, 0.02]))
    want = orc.costas_loop(x, const, bw, idx, ph)
    cl = pkg.CostasLoop(bw, const)
    cuts = [0] + np.sort(rng.choice(np.arange(1, n), int(rng.integers(0, 5)), replace=False)).tolist() + [n]
    outs = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sel = (idx >= lo) & (idx < hi)
        t = np.zeros(int(sel.sum()), dtype=pkg.TAG_DTYPE)
        t["index"], t["phase"], t["flags"] = idx[sel] - lo, ph[sel], pkg.TAG_SYNCWORD
        outs.append(tp.host(cl.process_bulk(tp.dev(x[lo:hi]), t)))
    y = np.concatenate(outs)
    ok = np.array_equal(tp.bits(y), tp.bits(want))
    bad += not ok
    print(f"case {case}: {const} {n} symbols, {idx.size} tags, bw {bw}, {len(cuts) - 1} calls: {'ok' if ok else 'MISMATCH'}")
print("fuzz:", cases - bad, "of", cases, "cases agree")
sys.exit(1 if bad else 0)
