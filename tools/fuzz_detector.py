#!/usr/bin/env python3
"""Randomised differential test of SyncwordDetection against the CPU oracle: random settings (samples per symbol, tap
count, time threshold, bin range, power threshold), random syncword positions, carrier offset and noise, the whole
stream in one call and in random chunks.  Output items and tag indices must be identical, tag values inside the bands
of tests/test_gpu_parity.py.  tools/fuzz_detector.py [cases=20] [seed=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import _oracle as orc
import _signals as sig
import test_gpu_parity as tp
pkg = ge.load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    sps = int(rng.choice([2, 4]))
    ntaps_req = int(rng.choice([10, 22, 44])) * (sps // 2)
    tthr = int(rng.integers(64, 1100))
    lo = int(rng.integers(-6, 3))
    hi = int(rng.integers(lo, min(lo + 9, 7)))
    thr = float(rng.uniform(6.0, 14.0))
    rrc = orc.rrc_taps(1.0, float(sps), 1.0, 0.35, ntaps_req)
    rrc = (rrc / np.sqrt(np.sum(rrc.astype(np.float64) ** 2))).astype(np.float32)
    L = 63 * sps + rrc.size
    nsym = int(rng.integers(20000, 60000))
    symbols = rng.integers(0, 2, nsym).astype(np.uint8)
    locs = np.sort(rng.choice(np.arange(100, nsym - 200), int(rng.integers(3, 12)), replace=False))
    for loc in locs:
        symbols[loc:loc + 64] = sig.SYNCWORD
    f = float(rng.uniform(lo - 0.4, hi + 0.4)) * np.pi / L
    x = orc.rotator(orc.interpolating_fir(sig.BPSK[symbols], sps, rrc), np.float32(f))
    x = (x + sig.awgn(x.size, float(rng.uniform(0.02, 0.4)), 1000 + case)).astype(np.complex64)
    kw = dict(samples_per_symbol=sps, time_threshold=tthr, power_threshold=thr)
    ref = orc.SyncwordDetection(rrc, sig.SYNCWORD, sig.BPSK, lo, hi, **kw)
    _, ref_out, ref_tags = ref.process(x)
    try:
        sd = pkg.SyncwordDetection(rrc, sig.SYNCWORD, sig.BPSK, lo, hi, max_items=x.size, **kw)
        cap = max(1024, 2 * ref_tags.size + 64)  # T = 64 in noise at a low threshold: thousands of tags
        st, out, tags, n = sd.process_bulk(tp.dev(x), tags_cap=cap)
        assert n == ref_out.size and np.array_equal(tp.bits(tp.host(out)), tp.bits(ref_out)), "items"
        tp.assert_tags_match(tags, ref_tags, rtol=3e-4)
        sd2 = pkg.SyncwordDetection(rrc, sig.SYNCWORD, sig.BPSK, lo, hi, max_items=x.size, **kw)
        pos, got = 0, []
        while pos + 2048 <= x.size:
            size = int(rng.integers(2048, 30000))
            _, _, t, d = sd2.process_bulk(tp.dev(x[pos:pos + size]), want_output=False, tags_cap=cap)
            t = t.copy()
            t["index"] += pos
            got.append(t)
            if d == 0:
                break
            pos += d
        got = np.concatenate(got) if got else tags[:0]
        assert np.array_equal(got["index"], tags["index"][: got.size]) and got.size >= tags.size - 2, "chunked indices"
        print(f"case {case}: sps {sps} taps {rrc.size} T {tthr} bins [{lo}, {hi}] thr {thr:.2f}: {tags.size} tags ok")
    except AssertionError as e:
        bad += 1
        print(f"case {case}: sps {sps} taps {rrc.size} T {tthr} bins [{lo}, {hi}] thr {thr:.2f}: MISMATCH {e}")
print("fuzz:", cases - bad, "of", cases, "cases agree")
sys.exit(1 if bad else 0)
