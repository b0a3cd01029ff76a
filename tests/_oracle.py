"""ctypes binding of the CPU oracle (oracle/libgr4pm_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libgr4pm_oracle.so")


class OrcTag(C.Structure):
    _fields_ = [
        ("index", C.c_uint64),
        ("amplitude", C.c_float),
        ("phase", C.c_float),
        ("freq", C.c_double),
        ("freq_bin", C.c_int32),
        ("noise_power", C.c_float),
        ("esn0_db", C.c_float),
        ("time_est", C.c_float),
        ("flags", C.c_int32),
        ("user", C.c_int32),
    ]


class OrcCrcParams(C.Structure):
    _fields_ = [("num_bits", C.c_uint), ("poly", C.c_uint64), ("initial_value", C.c_uint64),
                ("final_xor", C.c_uint64), ("input_reflected", C.c_int), ("result_reflected", C.c_int)]


CRC32 = dict(num_bits=32, poly=0x4C11DB7, initial_value=0xFFFFFFFF, final_xor=0xFFFFFFFF, input_reflected=True,
             result_reflected=True)  # crc_check.hpp:61-66 defaults


TAG_DTYPE = np.dtype(
    [
        ("index", "<u8"),
        ("amplitude", "<f4"),
        ("phase", "<f4"),
        ("freq", "<f8"),
        ("freq_bin", "<i4"),
        ("noise_power", "<f4"),
        ("esn0_db", "<f4"),
        ("time_est", "<f4"),
        ("flags", "<i4"),
        ("user", "<i4"),
    ],
    align=True,
)
assert TAG_DTYPE.itemsize == C.sizeof(OrcTag) == 48

# symbol-rate control tags (orc_ptag == gr4pm_packet_tag): kind 1 syncword / 2 header / 3 payload
PTAG_DTYPE = np.dtype(
    [
        ("index", "<u8"),
        ("kind", "<i4"),
        ("constellation", "<i4"),
        ("loop_bandwidth", "<f8"),
        ("packet_length", "<u8"),
        ("payload_symbols", "<u8"),
        ("payload_bits", "<u8"),
        ("syncword", TAG_DTYPE),
    ],
    align=True,
)
assert PTAG_DTYPE.itemsize == 48 + TAG_DTYPE.itemsize


def build(force=False):
    src = os.path.join(ORACLE_DIR, "gr4pm_oracle.cpp")
    hdr = os.path.join(ORACLE_DIR, "gr4pm_oracle.h")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libgr4pm_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        vp, sz, u64p, f32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        L.orc_rrc_taps.restype = sz
        L.orc_rrc_taps.argtypes = [C.c_double] * 4 + [sz, vp]
        L.orc_tx_rrc_taps.restype = sz
        L.orc_tx_rrc_taps.argtypes = [sz, vp]
        L.orc_fft.restype = None
        L.orc_fft.argtypes = [vp, vp, sz]
        L.orc_sd_create.restype = vp
        L.orc_sd_create.argtypes = [sz, sz, vp, sz, vp, sz, vp, sz, C.c_int, C.c_int, C.c_uint64, C.c_float]
        L.orc_sd_destroy.argtypes = [vp]
        L.orc_sd_syncword_samples_size.restype = sz
        L.orc_sd_syncword_samples_size.argtypes = [vp]
        L.orc_sd_self_corr.restype = C.c_float
        L.orc_sd_self_corr.argtypes = [vp]
        L.orc_sd_template.argtypes = [vp, sz, vp]
        L.orc_sd_process.restype = C.c_int
        L.orc_sd_process.argtypes = [vp, vp, sz, vp, C.POINTER(sz), vp, sz, C.POINTER(sz), vp, vp]
        L.orc_sdf_create.restype = vp
        L.orc_sdf_create.argtypes = [sz, sz, sz]
        L.orc_sdf_destroy.argtypes = [vp]
        L.orc_sdf_process.restype = C.c_int
        L.orc_sdf_process.argtypes = [vp, vp, sz, vp, sz, C.c_int, sz, vp, vp, sz,
                                      C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_int)]
        L.orc_cfc_create.restype = vp
        L.orc_cfc_create.argtypes = [sz]
        L.orc_cfc_destroy.argtypes = [vp]
        L.orc_cfc_process.argtypes = [vp, vp, sz, vp, vp, vp, sz]
        L.orc_rot_create.restype = vp
        L.orc_rot_create.argtypes = [C.c_float]
        L.orc_rot_destroy.argtypes = [vp]
        L.orc_rot_process.argtypes = [vp, vp, sz, vp]
        L.orc_costas_create.restype = vp
        L.orc_costas_create.argtypes = [C.c_double, C.c_int]
        L.orc_costas_destroy.argtypes = [vp]
        L.orc_sincosf.argtypes = [vp, sz, vp, vp]
        L.orc_costas_coeffs.argtypes = [vp, f32p, f32p]
        L.orc_costas_process.argtypes = [vp, vp, sz, vp, vp, vp, sz]
        L.orc_wipe_create.restype = vp
        L.orc_wipe_create.argtypes = [vp, sz]
        L.orc_wipe_destroy.argtypes = [vp]
        L.orc_wipe_process.argtypes = [vp, vp, sz, vp, vp, sz]
        L.orc_ifir_create.restype = vp
        L.orc_ifir_create.argtypes = [sz, vp, sz]
        L.orc_ifir_destroy.argtypes = [vp]
        L.orc_ifir_process_c64.argtypes = [vp, vp, sz, vp]
        L.orc_ifir_process_f32.argtypes = [vp, vp, sz, vp]
        L.orc_ifir_int.argtypes = [sz, vp, sz, vp, sz, vp]
        L.orc_symf_create.restype = vp
        L.orc_symf_create.argtypes = [sz, vp, sz, sz, sz]
        L.orc_symf_destroy.argtypes = [vp]
        L.orc_symf_process_c64.restype = sz
        L.orc_symf_process_c64.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]
        L.orc_symf_process_f32.restype = sz
        L.orc_symf_process_f32.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
        L.orc_arb_create.restype = vp
        L.orc_arb_create.argtypes = [C.c_double, C.c_int, vp, sz, sz]
        L.orc_arb_destroy.argtypes = [vp]
        L.orc_arb_process.restype = sz
        L.orc_arb_process.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
        szp = C.POINTER(sz)
        L.orc_pmi_create.restype = vp
        L.orc_pmi_create.argtypes = [sz, sz, C.c_double, C.c_double, C.c_double]
        L.orc_pmi_destroy.argtypes = [vp]
        L.orc_pmi_process.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp, sz, vp, sz, szp, szp, szp, szp, szp]
        L.orc_costas_process_packets.argtypes = [vp, vp, sz, vp, vp, sz]
        L.orc_sr_create.restype = vp
        L.orc_sr_create.argtypes = [sz]
        L.orc_sr_destroy.argtypes = [vp]
        L.orc_sr_process.restype = sz
        L.orc_sr_process.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, szp]
        L.orc_sr_process_int.restype = sz
        L.orc_sr_process_int.argtypes = [vp, vp, sz, vp, vp, sz]
        L.orc_llr_create.restype = vp
        L.orc_llr_create.argtypes = [C.c_float, C.c_int]
        L.orc_llr_destroy.argtypes = [vp]
        L.orc_llr_process.restype = sz
        L.orc_llr_process.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, szp]
        u64 = C.c_uint64
        L.orc_scr_create.restype = vp
        L.orc_scr_create.argtypes = [u64, u64, u64, u64]
        L.orc_scr_destroy.argtypes = [vp]
        L.orc_scr_process_f32.argtypes = [vp, vp, sz, vp, vp, sz]
        L.orc_scr_process_u8.argtypes = [vp, vp, sz, vp, vp, sz]
        L.orc_hps_create.restype = vp
        L.orc_hps_create.argtypes = [sz]
        L.orc_hps_destroy.argtypes = [vp]
        L.orc_hps_process.argtypes = [vp, vp, sz, vp, szp, vp, szp, vp, sz, vp, szp, vp, szp, sz]
        L.orc_header_fec_encode.argtypes = [vp, vp, sz, vp]
        L.orc_ldpc_create.restype = vp
        L.orc_ldpc_create.argtypes = [C.c_char_p]
        L.orc_ldpc_destroy.argtypes = [vp]
        L.orc_ldpc_decode.argtypes = [vp, vp, vp, C.c_uint]
        L.orc_header_fec_decode.argtypes = [vp, vp, sz, vp, vp]
        L.orc_header_fec_decode_q8.argtypes = [vp, vp, sz, vp, vp]
        L.orc_ldpc_decode_q8.argtypes = [vp, vp, vp, C.c_uint]
        L.orc_crc_compute.restype = C.c_uint64
        L.orc_crc_compute.argtypes = [C.POINTER(OrcCrcParams), vp, sz]
        L.orc_crc_check.restype = sz
        L.orc_crc_check.argtypes = [C.POINTER(OrcCrcParams), C.c_int, C.c_int, C.c_uint64, vp, vp, sz, vp, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- helpers
def rrc_taps(gain, fs, symbol_rate, alpha, ntaps):
    out = np.zeros(ntaps | 1, dtype=np.float32)
    n = lib().orc_rrc_taps(gain, fs, symbol_rate, alpha, ntaps, _p(out))
    return out[:n]


def tx_rrc_taps(sps):
    out = np.zeros((sps * 11) | 1, dtype=np.float32)
    n = lib().orc_tx_rrc_taps(sps, _p(out))
    return out[:n]


def unit_norm_rrc(sps=4):
    """packet_receiver.hpp:60-74 / qa_syncword_detection.cpp:64-75 (float accumulate)."""
    t = rrc_taps(1.0, float(sps), 1.0, 0.35, sps * 11)
    norm = np.float32(0.0)
    for x in t:
        norm = np.float32(norm + np.float32(x * x))
    norm = np.float32(np.sqrt(norm))
    return (t / norm).astype(np.float32), norm


def fft(x):
    x = _c64(x)
    out = np.empty_like(x)
    lib().orc_fft(_p(x), _p(out), x.size)
    return out


class SyncwordDetection:
    def __init__(self, rrc_taps, syncword, constellation, min_freq_bin=0, max_freq_bin=0,
                 fft_size=2048, samples_per_symbol=4, time_threshold=768, power_threshold=9.5):
        self.rrc = _f32(rrc_taps)
        self.sw = np.ascontiguousarray(syncword, dtype=np.uint8)
        self.const = _c64(constellation)
        self.fft_size = fft_size
        self.time_threshold = time_threshold
        self.nbins = max_freq_bin - min_freq_bin + 1
        self.h = lib().orc_sd_create(fft_size, samples_per_symbol, _p(self.rrc), self.rrc.size,
                                     _p(self.sw), self.sw.size, _p(self.const), self.const.size,
                                     min_freq_bin, max_freq_bin, time_threshold, power_threshold)
        if not self.h:
            raise ValueError("orc_sd_create failed")
        self._syncword_samples_size = lib().orc_sd_syncword_samples_size(self.h)
        self._syncword_self_corr = lib().orc_sd_self_corr(self.h)

    def template(self, b):
        out = np.empty(self.fft_size, dtype=np.complex64)
        lib().orc_sd_template(self.h, b, _p(out))
        return out

    def process(self, x, debug=False, tags_cap=4096):
        x = _c64(x)
        out = np.zeros(x.size, dtype=np.complex64)
        tags = np.zeros(tags_cap, dtype=TAG_DTYPE)
        n_done, n_tags = C.c_size_t(0), C.c_size_t(0)
        zpow = np.zeros(x.size, dtype=np.float32) if debug else None
        bins = np.zeros(x.size, dtype=np.int32) if debug else None
        st = lib().orc_sd_process(self.h, _p(x), x.size, _p(out), C.byref(n_done), _p(tags), tags_cap,
                                  C.byref(n_tags), _p(zpow), _p(bins))
        n = n_done.value
        assert n_tags.value <= tags_cap
        res = (st, out[:n], tags[: n_tags.value].copy())
        if debug:
            res += (zpow[:n], bins[:n])
        return res

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sd_destroy(self.h)
            self.h = None


class SyncwordDetectionFilter:
    def __init__(self, samples_per_symbol=4, syncword_size=64, header_size=128):
        self.h = lib().orc_sdf_create(samples_per_symbol, syncword_size, header_size)

    def process(self, x, out_cap, tag_flags=0, headers=(), n_ignored=0):
        """one processBulk call; headers: list of packet_length or None (invalid_header)"""
        x = _c64(x)
        out = np.zeros(max(out_cap, 1), dtype=np.complex64)
        plen = np.array([h if h is not None else 0 for h in headers] + [0], dtype=np.uint64)
        inval = np.array([h is None for h in headers] + [0], dtype=np.uint8)
        c, hc, ic, tf = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        st = lib().orc_sdf_process(self.h, _p(x), x.size, _p(out), out_cap, tag_flags, len(headers),
                                   _p(plen), _p(inval), n_ignored, C.byref(c), C.byref(hc), C.byref(ic),
                                   C.byref(tf))
        return st, out[: c.value], c.value, hc.value, ic.value, tf.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sdf_destroy(self.h)
            self.h = None


def coarse_frequency_correction(x, tag_index, tag_freq, delay=0, state=None):
    x = _c64(x)
    out = np.empty_like(x)
    ti = np.ascontiguousarray(tag_index, dtype=np.uint64)
    tf = np.ascontiguousarray(tag_freq, dtype=np.float64)
    h = state or lib().orc_cfc_create(delay)
    lib().orc_cfc_process(h, _p(x), x.size, _p(out), _p(ti), _p(tf), ti.size)
    if state is None:
        lib().orc_cfc_destroy(h)
    return out


def rotator(x, phase_incr):
    x = _c64(x)
    out = np.empty_like(x)
    h = lib().orc_rot_create(phase_incr)
    lib().orc_rot_process(h, _p(x), x.size, _p(out))
    lib().orc_rot_destroy(h)
    return out


CONSTELLATIONS = {"PILOT": 0, "BPSK": 1, "QPSK": 2}


def sincosf(x):
    """the host libm's sinf / cosf (what the reference's CostasLoop calls)"""
    x = _f32(x)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_sincosf(_p(x), x.size, _p(s), _p(c))
    return s, c


def costas_coeffs(loop_bandwidth, constellation):
    h = lib().orc_costas_create(loop_bandwidth, CONSTELLATIONS[constellation])
    k1, k2 = C.c_float(0), C.c_float(0)
    lib().orc_costas_coeffs(h, C.byref(k1), C.byref(k2))
    lib().orc_costas_destroy(h)
    return k1.value, k2.value


def costas_loop(x, constellation="BPSK", loop_bandwidth=0.01, tag_index=(), tag_phase=()):
    x = _c64(x)
    out = np.empty_like(x)
    ti = np.ascontiguousarray(tag_index, dtype=np.uint64)
    tp = np.ascontiguousarray(tag_phase, dtype=np.float32)
    h = lib().orc_costas_create(loop_bandwidth, CONSTELLATIONS[constellation])
    lib().orc_costas_process(h, _p(x), x.size, _p(out), _p(ti), _p(tp), ti.size)
    lib().orc_costas_destroy(h)
    return out


class CostasLoop:
    """stateful CostasLoop driven by control tags (PTAG_DTYPE), costas_loop.hpp:52-148"""

    def __init__(self, loop_bandwidth=0.01, constellation="BPSK"):
        self._h = lib().orc_costas_create(loop_bandwidth, CONSTELLATIONS[constellation])

    def process(self, x, tags):
        x = _c64(x)
        out = np.empty_like(x)
        tags = np.ascontiguousarray(tags, dtype=PTAG_DTYPE)
        lib().orc_costas_process_packets(self._h, _p(x), x.size, _p(out), _p(tags), tags.size)
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_costas_destroy(self._h)
            self._h = None


class PayloadMetadataInsert:
    """payload_metadata_insert.hpp:77-307; headers: packet_length per packet, None = invalid"""

    def __init__(self, syncword_size=64, header_size=128, syncword_bw=0.02, header_bw=0.01, payload_bw=0.005):
        self._h = lib().orc_pmi_create(syncword_size, header_size, syncword_bw, header_bw, payload_bw)

    def process(self, x, tags, headers=(), out_cap=None, tags_cap=None):
        x = _c64(x)
        tags = np.ascontiguousarray(tags, dtype=TAG_DTYPE)
        out_cap = x.size if out_cap is None else out_cap
        out = np.zeros(max(out_cap, 1), dtype=np.complex64)
        tags_cap = 3 * tags.size + 8 if tags_cap is None else tags_cap
        tout = np.zeros(tags_cap, dtype=PTAG_DTYPE)
        hl = np.array([0 if h is None else int(h) for h in headers], dtype=np.uint64)
        hi = np.array([1 if h is None else 0 for h in headers], dtype=np.uint8)
        vals = [C.c_size_t(0) for _ in range(5)]
        rc = lib().orc_pmi_process(self._h, _p(x), x.size, _p(out), out_cap, _p(tags), tags.size, _p(hl), _p(hi),
                                   hl.size, _p(tout), tags_cap, *[C.byref(v) for v in vals])
        n_tags, consumed, produced, used, ignored = [v.value for v in vals]
        assert rc == 0, "tags_cap too small"
        return {"out": out[:produced], "tags": tout[:n_tags], "consumed": consumed, "headers_used": used,
                "ignored": ignored}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pmi_destroy(self._h)
            self._h = None


class SyncwordRemove:
    """syncword_remove.hpp:39-105"""

    def __init__(self, syncword_size=64):
        self._h = lib().orc_sr_create(syncword_size)

    def process(self, x, tags):
        x = _c64(x)
        tags = np.ascontiguousarray(tags, dtype=PTAG_DTYPE)
        out = np.empty_like(x)
        tout = np.zeros(tags.size + 1, dtype=PTAG_DTYPE)
        nt = C.c_size_t(0)
        n = lib().orc_sr_process(self._h, _p(x), x.size, _p(out), _p(tags), tags.size, _p(tout), tout.size,
                                 C.byref(nt))
        return out[:n], tout[: nt.value]

    def process_int(self, x, tag_index):
        x = np.ascontiguousarray(x, dtype=np.int32)
        ti = np.ascontiguousarray(tag_index, dtype=np.uint64)
        out = np.empty_like(x)
        n = lib().orc_sr_process_int(self._h, _p(x), x.size, _p(out), _p(ti), ti.size)
        return out[:n]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_sr_destroy(self._h)
            self._h = None


class ConstellationLLRDecoder:
    """constellation_llr_decoder.hpp:55-134"""

    def __init__(self, noise_sigma=1.0, constellation="BPSK"):
        self._h = lib().orc_llr_create(noise_sigma, CONSTELLATIONS[constellation])

    def process(self, x, tags=None):
        x = _c64(x)
        tags = np.zeros(0, dtype=PTAG_DTYPE) if tags is None else np.ascontiguousarray(tags, dtype=PTAG_DTYPE)
        out = np.empty(2 * x.size, dtype=np.float32)
        tout = np.zeros(tags.size + 1, dtype=PTAG_DTYPE)
        nt = C.c_size_t(0)
        n = lib().orc_llr_process(self._h, _p(x), x.size, _p(out), _p(tags), tags.size, _p(tout), tout.size,
                                  C.byref(nt))
        if n == C.c_size_t(-1).value:
            raise ValueError("constellation not supported")
        return out[:n], tout[: nt.value]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_llr_destroy(self._h)
            self._h = None


class AdditiveScrambler:
    """additive_scrambler.hpp:58-100 (float soft symbols or uint8 hard symbols)"""

    def __init__(self, mask=0x8A, seed=0x7F, length=7, count=0):
        self._h = lib().orc_scr_create(mask, seed, length, count)

    def process(self, x, reset_index=()):
        ri = np.ascontiguousarray(reset_index, dtype=np.uint64)
        if np.asarray(x).dtype == np.uint8:
            x = np.ascontiguousarray(x, dtype=np.uint8)
            out = np.empty_like(x)
            lib().orc_scr_process_u8(self._h, _p(x), x.size, _p(out), _p(ri), ri.size)
        else:
            x = _f32(x)
            out = np.empty_like(x)
            lib().orc_scr_process_f32(self._h, _p(x), x.size, _p(out), _p(ri), ri.size)
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scr_destroy(self._h)
            self._h = None


class HeaderPayloadSplit:
    """header_payload_split.hpp:38-135 on float items; tags: PTAG_DTYPE (kind 3 = payload_bits)"""

    def __init__(self, header_size=256):
        self._h = lib().orc_hps_create(header_size)

    def process(self, x, tags):
        x = _f32(x)
        tags = np.ascontiguousarray(tags, dtype=PTAG_DTYPE)
        header, payload = np.empty_like(x), np.empty_like(x)
        ht, pt = np.zeros(tags.size + 1, dtype=PTAG_DTYPE), np.zeros(tags.size + 1, dtype=PTAG_DTYPE)
        v = [C.c_size_t(0) for _ in range(4)]
        rc = lib().orc_hps_process(self._h, _p(x), x.size, _p(header), C.byref(v[0]), _p(payload), C.byref(v[1]),
                                   _p(tags), tags.size, _p(ht), C.byref(v[2]), _p(pt), C.byref(v[3]), tags.size + 1)
        if rc != 0:
            raise ValueError("received unexpected payload_bits tag")
        return header[: v[0].value], payload[: v[1].value], ht[: v[2].value], pt[: v[3].value]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_hps_destroy(self._h)
            self._h = None


def header_fec_encode(header_bytes, generator):
    """header_fec_encoder.hpp:60-107: [n, 4] bytes -> [n, 32] bytes"""
    h = np.ascontiguousarray(header_bytes, dtype=np.uint8).reshape(-1, 4)
    g = np.ascontiguousarray(generator, dtype=np.uint32)
    out = np.empty((h.shape[0], 32), dtype=np.uint8)
    lib().orc_header_fec_encode(_p(g), _p(h), h.shape[0], _p(out))
    return out


class HeaderFecDecoder:
    """header_fec_decoder.hpp:290-347 with the LDPC decoder restated (see gr4pm_oracle.h)"""

    def __init__(self, alist):
        self._h = lib().orc_ldpc_create(alist.encode())
        assert self._h

    def process(self, llrs, arithmetic=0):
        """arithmetic 1: the product's 8-bit message form (gr4pm_header_fec_decoder_params::arithmetic)"""
        x = _f32(llrs)
        n = x.size // 256
        out = np.empty((n, 4), dtype=np.uint8)
        inval = np.empty(n, dtype=np.uint8)
        (lib().orc_header_fec_decode_q8 if arithmetic else lib().orc_header_fec_decode)(self._h, _p(x), n, _p(out), _p(inval))
        return out, inval.astype(bool)

    def decode(self, llrs, max_iterations=25):
        x = _f32(llrs)
        bits = np.empty(32, dtype=np.uint8)
        it = lib().orc_ldpc_decode(self._h, _p(x), _p(bits), max_iterations)
        return bits, it

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ldpc_destroy(self._h)
            self._h = None


def header_format(packet_length, packet_type=0):
    """header_formatter.hpp:104-107: big-endian length, type (0 user data, 1 idle), spare 0x55"""
    return np.array([(packet_length >> 8) & 0xFF, packet_length & 0xFF, packet_type, 0x55], dtype=np.uint8)


def header_parse(header, invalid=False):
    """header_parser.hpp:56-85: packet_length or None (invalid_header)"""
    length = (int(header[0]) << 8) | int(header[1])
    if invalid or length == 0 or header[2] not in (0, 1):
        return None
    return length


def crc_compute(data, **params):
    """Crc<uint64_t>::compute, crc.hpp:119-156"""
    pr = OrcCrcParams(*[int(params[k]) for k in ("num_bits", "poly", "initial_value", "final_xor",
                                                  "input_reflected", "result_reflected")])
    d = np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().orc_crc_compute(C.byref(pr), _p(d), d.size))


def crc_check(data, packet_len, swap_endianness=False, discard_crc=False, skip_header_bytes=0, **params):
    """CrcCheck over packets laid back to back, crc_check.hpp:152-208.  Returns (out bytes, out_len per packet)"""
    pr = OrcCrcParams(*[int(params[k]) for k in ("num_bits", "poly", "initial_value", "final_xor",
                                                  "input_reflected", "result_reflected")])
    d = np.ascontiguousarray(data, dtype=np.uint8)
    pl = np.ascontiguousarray(packet_len, dtype=np.uint64)
    out = np.empty(max(d.size, 1), dtype=np.uint8)
    ol = np.zeros(max(pl.size, 1), dtype=np.uint64)
    n = lib().orc_crc_check(C.byref(pr), int(swap_endianness), int(discard_crc), skip_header_bytes, _p(d), _p(pl),
                            pl.size, _p(out), _p(ol))
    return out[:n], ol[: pl.size]


def syncword_wipeoff(x, syncword, tag_index):
    x = _c64(x)
    out = np.empty_like(x)
    sw = _f32(syncword)
    ti = np.ascontiguousarray(tag_index, dtype=np.uint64)
    h = lib().orc_wipe_create(_p(sw), sw.size)
    lib().orc_wipe_process(h, _p(x), x.size, _p(out), _p(ti), ti.size)
    lib().orc_wipe_destroy(h)
    return out


def interpolating_fir(x, interpolation, taps):
    taps = _f32(taps)
    h = lib().orc_ifir_create(interpolation, _p(taps), taps.size)
    if np.iscomplexobj(x):
        x = _c64(x)
        out = np.empty(x.size * interpolation, dtype=np.complex64)
        lib().orc_ifir_process_c64(h, _p(x), x.size, _p(out))
    else:
        x = _f32(x)
        out = np.empty(x.size * interpolation, dtype=np.float32)
        lib().orc_ifir_process_f32(h, _p(x), x.size, _p(out))
    lib().orc_ifir_destroy(h)
    return out


def interpolating_fir_int(x, interpolation, taps):
    x = np.ascontiguousarray(x, dtype=np.int32)
    taps = np.ascontiguousarray(taps, dtype=np.int32)
    out = np.empty(x.size * interpolation, dtype=np.int32)
    lib().orc_ifir_int(interpolation, _p(taps), taps.size, _p(x), x.size, _p(out))
    return out


def symbol_filter(x, taps, num_arms, samples_per_symbol, delay, tags=None, out_cap=None):
    """returns (symbols, tags_out, consumed)"""
    taps = _f32(taps)
    h = lib().orc_symf_create(samples_per_symbol, _p(taps), taps.size, num_arms, delay)
    if out_cap is None:
        out_cap = len(x) // samples_per_symbol + 8
    consumed = C.c_size_t(0)
    if np.iscomplexobj(x):
        x = _c64(x)
        out = np.zeros(out_cap, dtype=np.complex64)
        tin = np.ascontiguousarray(tags if tags is not None else np.zeros(0, dtype=TAG_DTYPE), dtype=TAG_DTYPE)
        tout = np.zeros(tin.size + 8, dtype=TAG_DTYPE)
        nto = C.c_size_t(0)
        n = lib().orc_symf_process_c64(h, _p(x), x.size, _p(out), out_cap, _p(tin), tin.size, _p(tout),
                                       tout.size, C.byref(nto), C.byref(consumed))
        res = out[:n], tout[: nto.value].copy(), consumed.value
    else:
        x = _f32(x)
        out = np.zeros(out_cap, dtype=np.float32)
        n = lib().orc_symf_process_f32(h, _p(x), x.size, _p(out), out_cap, C.byref(consumed))
        res = out[:n], np.zeros(0, dtype=TAG_DTYPE), consumed.value
    lib().orc_symf_destroy(h)
    return res


def pfb_arb_resampler(x, rate, taps, filter_size=32, rate_is_double=True, out_cap=None):
    x = _c64(x)
    taps = _f32(taps)
    if out_cap is None:
        out_cap = int(x.size * rate) + 64
    out = np.zeros(out_cap, dtype=np.complex64)
    consumed = C.c_size_t(0)
    h = lib().orc_arb_create(rate, 1 if rate_is_double else 0, _p(taps), taps.size, filter_size)
    n = lib().orc_arb_process(h, _p(x), x.size, _p(out), out_cap, C.byref(consumed))
    lib().orc_arb_destroy(h)
    return out[:n], consumed.value


class PfbArbResampler:
    """the same restatement with its state kept across calls (pfb_arb_resampler.hpp:122-182): call-by-call
    comparisons with the device block under identical input / output span sizes"""

    def __init__(self, rate, taps, filter_size=32, rate_is_double=True):
        taps = _f32(taps)
        self._h = lib().orc_arb_create(rate, 1 if rate_is_double else 0, _p(taps), taps.size, filter_size)

    def process(self, x, out_cap):
        x = _c64(x)
        out = np.zeros(max(out_cap, 1), dtype=np.complex64)
        consumed = C.c_size_t(0)
        n = lib().orc_arb_process(self._h, _p(x), x.size, _p(out), out_cap, C.byref(consumed))
        return out[:n], consumed.value

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_arb_destroy(self._h)
            self._h = None


class InterpolatingFir:
    """interpolating_fir_filter.hpp:76-102 with the history kept across calls"""

    def __init__(self, interpolation, taps):
        taps = _f32(taps)
        self.interpolation = interpolation
        self._h = lib().orc_ifir_create(interpolation, _p(taps), taps.size)

    def process(self, x):
        if np.iscomplexobj(x):
            x = _c64(x)
            out = np.empty(x.size * self.interpolation, dtype=np.complex64)
            lib().orc_ifir_process_c64(self._h, _p(x), x.size, _p(out))
        else:
            x = _f32(x)
            out = np.empty(x.size * self.interpolation, dtype=np.float32)
            lib().orc_ifir_process_f32(self._h, _p(x), x.size, _p(out))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ifir_destroy(self._h)
            self._h = None


# ---------------------------------------------------------------- reference-built taps
def ref_taps_dump(*args):
    """runs oracle/_ref/ref_taps_dump (built from the reference's own headers) if present"""
    exe = os.path.join(ORACLE_DIR, "_ref", "ref_taps_dump")
    if not os.path.exists(exe):
        return None
    raw = subprocess.check_output([exe] + [str(a) for a in args])
    return np.frombuffer(raw, dtype=np.float32).copy()
