#!/usr/bin/env python3
"""One CPU-baseline worker of bench.py (no torch, no GPU): runs the CPU oracle over a sample file until the time
is up and prints the number of samples it got through.
usage: cpu_baseline_worker.py <piece.npy> <front_end|detector> <seconds>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc  # noqa: E402

SPS, BINS = 4, 4
SYNCWORD = np.array(
    [0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 1, 1, 1,
     0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0], dtype=np.uint8)


def main():
    piece = np.load(sys.argv[1], mmap_mode="r")
    piece = np.ascontiguousarray(piece)
    leg, seconds = sys.argv[2], float(sys.argv[3])
    rrc, norm = orc.unit_norm_rrc(SPS)
    bpsk = np.array([1, -1], dtype=np.complex64)
    pfb = orc.rrc_taps(32.0 / float(norm), 32.0 * SPS, 1.0, 0.35, 32 * SPS * 11)[:-1]
    bipolar = np.where(SYNCWORD == 1, -1.0, 1.0).astype(np.float32)
    sd = orc.SyncwordDetection(rrc, SYNCWORD, bpsk, -BINS, BINS, power_threshold=9.5)
    done = 0
    t0 = time.perf_counter()
    while True:
        _, out, tg = sd.process(piece, tags_cap=1 << 16)
        if leg == "front_end":
            # packet_receiver.hpp:76-127 block after block; the tag gate is a pure copy that accepts every tag
            z = orc.coarse_frequency_correction(out, tg["index"], tg["freq"], delay=26)
            sym, sym_tags, _ = orc.symbol_filter(z, pfb, 32, SPS, 44, tags=tg.astype(orc.TAG_DTYPE))
            w = orc.syncword_wipeoff(sym, bipolar, sym_tags["index"])
            orc.costas_loop(w, "QPSK", 0.01, sym_tags["index"], sym_tags["phase"])
        done += out.size
        if time.perf_counter() - t0 >= seconds:
            break
    print(done, time.perf_counter() - t0)


if __name__ == "__main__":
    main()
