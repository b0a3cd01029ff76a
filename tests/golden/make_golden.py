#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ that come from the reference checkout.

Run in the build container (needs /root/reference and oracle/_ref/ref_taps_dump built by
`make -C oracle`).  Only DATA is written: filter taps produced by the reference's own
standalone headers compiled unchanged, and the 65-tap literal held by test/qa_firdes.cpp.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as orc  # noqa: E402

REF = "/root/reference"


def main():
    # G1a: literal known-answer vector of test/qa_firdes.cpp:11-34
    src = open(os.path.join(REF, "test", "qa_firdes.cpp")).read()
    body = src[src.index("expected_taps = {") : src.index("};", src.index("expected_taps = {"))]
    vals = [float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", body)]
    assert len(vals) == 65
    np.save(os.path.join(HERE, "qa_firdes_rrc65.npy"), np.array(vals, dtype=np.float32))
    # G1b: taps from the reference headers themselves (oracle/_ref/ref_taps_dump)
    jobs = {
        "ref_rrc_1_4_1_0.35_44": ("rrc", 1.0, 4.0, 1.0, 0.35, 44),
        "ref_rrc_1_4_1_0.35_65": ("rrc", 1.0, 4.0, 1.0, 0.35, 65),
        "ref_rrc_32_128_1_0.35_1408": ("rrc", 32.0, 128.0, 1.0, 0.35, 1408),
        "ref_rrc_1_4_1_0.35_1024": ("rrc", 1.0, 4.0, 1.0, 0.35, 1024),
        "ref_txrrc_4": ("txrrc", 4),
        "ref_pfb_arb_taps": ("pfbarb",),
    }
    for name, args in jobs.items():
        taps = orc.ref_taps_dump(*args)
        assert taps is not None, "build oracle/_ref first (make -C oracle)"
        np.save(os.path.join(HERE, name + ".npy"), taps)
        print(name, taps.size)


if __name__ == "__main__":
    main()
