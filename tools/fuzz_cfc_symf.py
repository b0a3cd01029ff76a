#!/usr/bin/env python3
"""Randomised differential test of the fused CoarseFrequencyCorrection + SymbolFilter call (k_rot_checkpoints,
k_symf_wg_plan, k_symbol_filter_fast) against the CPU oracle: hundreds of detection tags at ragged distances (shorter
than a checkpoint chunk, around the tile size, longer than the renormalisation period), random frequencies and time
estimates, random call boundaries.  Symbols bit-exact, re-timed tag indices identical.
tools/fuzz_cfc_symf.py [cases=10] [seed=1]"""
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
rrc, pfb = tp._receiver_pfb()
bad = 0
for case in range(cases):
    n = int(rng.integers(60000, 300000))
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    kinds = [rng.integers(1, 40, int(rng.integers(0, 30))), rng.integers(40, 1500, int(rng.integers(5, 120))),
             rng.integers(900, 1100, int(rng.integers(0, 40))), rng.integers(1500, 30000, int(rng.integers(2, 30)))]
    gaps = np.concatenate(kinds)
    rng.shuffle(gaps)
    idx = np.cumsum(gaps)
    idx = idx[idx < n - 10].astype(np.uint64)
    tags = np.zeros(idx.size, dtype=pkg.TAG_DTYPE)
    tags["index"] = idx
    tags["amplitude"] = rng.uniform(0.5, 2.0, idx.size)
    tags["time_est"] = rng.uniform(-0.5, 0.5, idx.size)
    tags["phase"] = rng.uniform(-3, 3, idx.size)
    tags["freq"] = rng.uniform(-0.04, 0.04, idx.size)
    tags["flags"] = pkg.TAG_SYNCWORD
    delay = int(rng.choice([0, 26]))
    z = orc.coarse_frequency_correction(x, tags["index"], tags["freq"], delay=delay)
    want, want_tags, wc = orc.symbol_filter(z, pfb, 32, 4, 44, tags=tags.astype(orc.TAG_DTYPE), out_cap=n // 4 + idx.size + 16)
    assert wc == n
    cuts = np.sort(rng.choice(np.arange(1, n), int(rng.integers(0, 6)), replace=False)).tolist()
    cuts = [0] + cuts + [n]
    cfc, sf = pkg.CoarseFrequencyCorrection(delay), pkg.SymbolFilter(pfb, 32, 4, 44)
    ys, ts, off = [], [], 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        t = tags[(tags["index"] >= lo) & (tags["index"] < hi)].copy()
        t["index"] -= lo
        y, tt, c = pkg.cfc_symbol_filter(cfc, sf, tp.dev(x[lo:hi]), t)
        assert c == hi - lo
        tt = tt.copy()
        tt["index"] += off
        off += y.numel()
        ys.append(tp.host(y))
        ts.append(tt)
    y, t = np.concatenate(ys), np.concatenate(ts)
    ok = y.size == want.size and np.array_equal(tp.bits(y), tp.bits(want)) and np.array_equal(t["index"], want_tags["index"])
    bad += not ok
    print(f"case {case}: {n} items, {idx.size} tags, delay {delay}, {len(cuts) - 1} calls: {'ok' if ok else 'MISMATCH'}")
    if not ok:
        m = min(y.size, want.size)
        d = np.nonzero(tp.bits(y[:m]) != tp.bits(want[:m]))[0]
        print(f"   sizes {y.size} / {want.size}; differing symbols {d.size}, first at {d[:8].tolist()}; "
              f"tag indices equal {np.array_equal(t['index'], want_tags['index'])}")
        if d.size:
            k = int(d[0])
            near = want_tags["index"][np.searchsorted(want_tags["index"], k) - 1: np.searchsorted(want_tags["index"], k) + 1]
            print(f"   got {y[k]} want {want[k]}; output tags around it {near.tolist()}; max |diff| {np.max(np.abs(y[:m] - want[:m]))}")
print("fuzz:", cases - bad, "of", cases, "cases agree")
sys.exit(1 if bad else 0)
