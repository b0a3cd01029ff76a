"""Seeded stimulus generators shared by the oracle tests and the GPU parity tests.

They rebuild, with numpy RNG seeds, the stimuli of the reference's own tests
(test/qa_*.cpp) -- the reference seeds from std::random_device, so the *procedure* is
reproduced, not the bits."""
import numpy as np

import _oracle as orc

# CCSDS 64-bit syncword, packet_receiver.hpp:45-59 / qa_syncword_detection.cpp:35-49
SYNCWORD = np.array(
    [0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 1, 1, 1,
     0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0],
    dtype=np.uint8,
)
BPSK = np.array([1.0 + 0j, -1.0 + 0j], dtype=np.complex64)
QA_SYNCWORD_LOCATIONS = [100, 1000, 1250, 10000, 13721, 43124, 58000, 127018, 525178, 893251]


def qa_syncword_stream(num_symbols, locations, freq_error, seed=1234, sps=4):
    """qa_syncword_detection.cpp:21-90: random BPSK symbols with the syncword inserted at
    `locations`, unit-norm 45-tap RRC interpolation by sps, rotator at freq_error."""
    rng = np.random.default_rng(seed)
    symbols = rng.integers(0, 2, size=num_symbols, dtype=np.uint8)
    for loc in locations:
        symbols[loc:loc + SYNCWORD.size] = SYNCWORD
    rrc, _ = orc.unit_norm_rrc(sps)
    x = orc.interpolating_fir(BPSK[symbols], sps, rrc)
    x = orc.rotator(x, np.float32(freq_error))
    return x, rrc


def awgn(n, sigma, seed):
    rng = np.random.default_rng(seed)
    return (sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.0)).astype(np.complex64)
