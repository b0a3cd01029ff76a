#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ that come from the reference checkout.

Run in the build container (needs /root/reference and oracle/_ref/ref_taps_dump built by
`make -C oracle`).  Only DATA is written: filter taps produced by the reference's own
standalone headers compiled unchanged, the 65-tap literal held by test/qa_firdes.cpp, and the
numeric definition of the header code -- the (128, 32) LDPC parity-check matrix in alist form
(header_fec_decoder.hpp:31-258), its dense generator (header_fec_encoder.hpp:28-45) and the two
byte vectors test/qa_header_fec_decoder.cpp decodes.
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


def header_code():
    inc = os.path.join(REF, "blocks", "include", "gnuradio-4.0", "packet-modem")
    src = open(os.path.join(inc, "header_fec_decoder.hpp")).read()
    a = src.index('alist[] = R""(') + len('alist[] = R""(')
    alist = src[a:src.index(')"";', a)]
    nums = [[int(v) for v in line.split()] for line in alist.strip().splitlines()]
    assert nums[0] == [128, 96]
    data_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)), "gr4-packet-modem_amd", "data")
    with open(os.path.join(data_dir, "header_ldpc_128_32.alist"), "w") as f:
        for row in nums:
            f.write(" ".join(str(v) for v in row) + "\n")
    src = open(os.path.join(inc, "header_fec_encoder.hpp")).read()
    a = src.index("_generator[] = {")
    gen = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]{8}", src[a:src.index("};", a)])]
    assert len(gen) == 96
    np.save(os.path.join(HERE, "header_ldpc_generator.npy"), np.array(gen, dtype=np.uint32))
    np.array(gen, dtype="<u4").tofile(os.path.join(data_dir, "header_ldpc_generator.u32"))  # BurstGenerator
    tx = orc.ref_taps_dump("txrrc", 4)  # packet_transmitter_rrc_taps(4), from the reference's own header
    assert tx is not None, "build oracle/_ref first (make -C oracle)"
    np.save(os.path.join(HERE, "ref_txrrc_4.npy"), tx.astype("<f4"))  # fixture only: the product computes these taps itself
    src = open(os.path.join(REF, "test", "qa_header_fec_decoder.cpp")).read()
    vecs = []
    pos = 0
    while True:
        a = src.find("std::vector<uint8_t> v = {", pos)
        if a < 0:
            break
        b = src.index("};", a)
        vecs.append(np.array([int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]{2}", src[a:b])], dtype=np.uint8))
        pos = b
    assert [v.size for v in vecs] == [8, 256]
    np.save(os.path.join(HERE, "qa_header_fec_valid_bytes.npy"), vecs[0])
    np.save(os.path.join(HERE, "qa_header_fec_random_bytes.npy"), vecs[1])
    print("header code: alist", len(nums), "lines, generator 96 rows, qa vectors", [v.size for v in vecs])
    # known-answer sequences of test/qa_additive_scrambler.cpp:13-24,59-61
    src = open(os.path.join(REF, "test", "qa_additive_scrambler.cpp")).read()
    a = src.index("ccsds_scrambling_sequence = {")
    seq = [int(v) for v in re.findall(r"[01]", src[src.index("{", a):src.index("};", a)])]
    assert len(seq) == 255
    np.save(os.path.join(HERE, "qa_ccsds_scrambling_sequence.npy"), np.array(seq, dtype=np.uint8))
    a = src.index("std::vector<uint8_t> expected = {")
    seq = [int(v) for v in re.findall(r"[01]", src[src.index("{", a):src.index("};", a)])]
    assert len(seq) == 40
    np.save(os.path.join(HERE, "qa_ccsds_2023_first40.npy"), np.array(seq, dtype=np.uint8))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "header_code":
        header_code()
        sys.exit(0)
    main()
