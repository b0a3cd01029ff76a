"""Host-side mirror of the reference's block interface for the RX hot path, over the C-ABI.

Same block names, setting names, defaults and error behaviour as the reference's
gr::Block<T> classes (cited per class; paths relative to
/root/reference/blocks/include/gnuradio-4.0/packet-modem/).  `start()` mirrors
gr::Block::start(), `process_bulk()` mirrors processBulk(); samples are torch tensors
resident in HBM (torch is only plumbing: device memory + the current HIP stream), tags are
numpy records (TAG_DTYPE) with explicit indices instead of the runtime's chunk-head tags.
The C++ gr::Block wrappers in host/ are the drop-in for the reference's flowgraphs; this
module is what the parity tests and bench.py drive."""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import PACKET_TAG_DTYPE, TAG_DTYPE, TAG_OTHER, TAG_SYNCWORD, Gr4pmError, check, lib

CONSTELLATIONS = {"PILOT": 0, "BPSK": 1, "QPSK": 2}  # constellation.hpp:6


def _torch():
    import torch
    return torch


def _stream_handle():
    torch = _torch()
    if not torch.cuda.is_available():
        raise Gr4pmError("no HIP device: the gr4pm blocks have no CPU fallback")
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _release(name, h):
    """destroy a C handle; silent when the interpreter is already tearing the module down"""
    try:
        getattr(lib(), name)(h)
    except Exception:
        pass


def _dev_c64(x, what="in"):
    torch = _torch()
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.complex64 and x.is_contiguous()):
        raise TypeError(f"{what} must be a contiguous complex64 CUDA tensor")
    return x


def _inputs_ready(x):
    """The receivers inside the library run on streams of their own (non-blocking: they do not order themselves against
    torch's streams).  An input that a torch kernel is still writing on the caller's current stream would be read too
    early: wait for that stream (a few microseconds when it is idle)."""
    _torch().cuda.current_stream(x.device).synchronize()


def _dev_c64_rows(x, what="in"):
    """[channels, n] with contiguous rows; the row stride is free (a window of a wider ring)"""
    torch = _torch()
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.complex64 and x.dim() == 2 and
            (x.shape[1] <= 1 or x.stride(1) == 1) and x.stride(0) >= x.shape[1]):
        raise TypeError(f"{what} must be a [channels, n] complex64 CUDA tensor with contiguous rows")
    return x


def _tags_array(tags):
    if tags is None:
        return np.zeros(0, dtype=TAG_DTYPE)
    return np.ascontiguousarray(tags, dtype=TAG_DTYPE)


def _ptags_array(tags):
    if tags is None:
        return np.zeros(0, dtype=PACKET_TAG_DTYPE)
    return np.ascontiguousarray(tags, dtype=PACKET_TAG_DTYPE)


def _header_msgs(headers):
    """parsed_header messages: packet_length, or None for an "invalid_header" message; returns
    (HEADER_MSG_DTYPE array with at least one slot, number of messages)"""
    if isinstance(headers, np.ndarray) and headers.dtype == _abi.HEADER_MSG_DTYPE:
        return (headers if headers.size else np.zeros(1, dtype=_abi.HEADER_MSG_DTYPE)), headers.size
    n = len(headers)
    msgs = np.zeros(max(n, 1), dtype=_abi.HEADER_MSG_DTYPE)
    if isinstance(headers, np.ndarray):  # all valid
        msgs["packet_length"][:n] = headers
        return msgs, n
    for i, h in enumerate(headers):
        if h is None:
            msgs[i]["invalid_header"] = 1
        else:
            msgs[i]["packet_length"] = int(h)
    return msgs, n


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def sincosf(x):
    """sinf / cosf as the CostasLoop kernels evaluate them (glibc's algorithm on the device), for the parity suite"""
    torch = _torch()
    xd = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    s, c = torch.empty_like(xd), torch.empty_like(xd)
    check(lib().gr4pm_sincosf(xd.data_ptr(), xd.numel(), s.data_ptr(), c.data_ptr()), "sincosf")
    return s.cpu().numpy(), c.cpu().numpy()


def costas_phase_wrap(x):
    """the phase wrap of a CostasLoop iteration (costas_loop.hpp:141-145) as the kernels evaluate it, for the parity suite"""
    torch = _torch()
    xd = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    out = torch.empty_like(xd)
    check(lib().gr4pm_costas_phase_wrap(xd.data_ptr(), xd.numel(), out.data_ptr()), "costas_phase_wrap")
    return out.cpu().numpy()


def packet_transmitter_rrc_taps(samples_per_symbol):
    """packet_transmitter_rrc_taps.hpp:8-28: the transmitter's RRC (root_raised_cosine(1, sps, 1, 0.35, 11 sps))
    scaled so that the largest polyphase |tap| sum is 0.9 (DAC head-room), in the reference's float32 arithmetic
    and summation order"""
    sps = int(samples_per_symbol)
    taps = root_raised_cosine(1.0, float(sps), 1.0, 0.35, sps * 11)
    worst = np.float32(0.0)
    for j in range(sps):
        acc = np.float32(0.0)
        for k in range(j, taps.size, sps):
            acc = np.float32(acc + np.abs(taps[k]))
        worst = max(worst, acc)
    return (taps * np.float32(np.float32(0.9) / worst)).astype(np.float32)


def root_raised_cosine(gain, sampling_freq, symbol_rate, alpha, ntaps):
    """firdes::root_raised_cosine<float>, firdes.hpp:29-76"""
    out = np.zeros(ntaps | 1, dtype=np.float32)
    n = lib().gr4pm_firdes_root_raised_cosine(gain, sampling_freq, symbol_rate, alpha, ntaps, _np_ptr(out))
    return out[:n]


class SyncwordDetection:
    """syncword_detection.hpp:32-357.  Settings :133-141 keep their names and defaults."""

    def __init__(self, rrc_taps, syncword, constellation, min_freq_bin=0, max_freq_bin=0, fft_size=2048,
                 samples_per_symbol=4, time_threshold=768, power_threshold=9.5, n_channels=1,
                 max_items=1 << 22):
        self.fft_size = fft_size
        self.samples_per_symbol = samples_per_symbol
        self.rrc_taps = np.ascontiguousarray(rrc_taps, dtype=np.float32)
        self.syncword = np.ascontiguousarray(syncword, dtype=np.uint8)
        self.constellation = np.ascontiguousarray(constellation, dtype=np.complex64)
        self.min_freq_bin = min_freq_bin
        self.max_freq_bin = max_freq_bin
        self.time_threshold = time_threshold
        self.power_threshold = power_threshold
        self.n_channels = n_channels
        self.max_items = max_items
        self._h = None
        self.start()

    def start(self):
        """start(), :143-202; raises Gr4pmError where the reference throws gr::exception"""
        self._destroy()
        p = _abi.SyncwordDetectionParams(
            self.fft_size, self.samples_per_symbol, _np_ptr(self.rrc_taps), self.rrc_taps.size,
            _np_ptr(self.syncword), self.syncword.size, _np_ptr(self.constellation), self.constellation.size,
            self.min_freq_bin, self.max_freq_bin, self.time_threshold, self.power_threshold, self.n_channels,
            self.max_items, _stream_handle())
        h = C.c_void_p()
        check(lib().gr4pm_syncword_detection_create(C.byref(p), C.byref(h)), "SyncwordDetection.start")
        self._h = h
        self._syncword_samples_size = lib().gr4pm_syncword_detection_syncword_samples_size(h)
        self._syncword_self_corr = lib().gr4pm_syncword_detection_self_corr(h)

    @property
    def _items_consumed(self):
        return lib().gr4pm_syncword_detection_items_consumed(self._h)

    def scan_counts(self, channel=0):
        """(candidates the scan visited, of which tested by the separate pass over the powers) in the last call"""
        v, m = C.c_uint64(0), C.c_uint64(0)
        lib().gr4pm_syncword_detection_scan_counts(self._h, channel, C.byref(v), C.byref(m))
        return v.value, m.value

    def reset(self):
        """back to the state right after start() without rebuilding the templates"""
        check(lib().gr4pm_syncword_detection_reset(self._h), "SyncwordDetection.reset")

    def announce(self, x):
        """look-ahead: names the input of a later call (after the ones already announced, at most
        two calls ahead); everything of that call that does not depend on the scan state runs on
        other streams behind the calls before it.  The caller keeps x alive and unchanged."""
        x = _dev_c64(x)
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        assert x2.shape[0] == self.n_channels
        check(lib().gr4pm_syncword_detection_announce(self._h, x2.data_ptr(), x2.stride(0), x2.shape[1]),
              "SyncwordDetection.announce")

    def process_bulk(self, x, want_output=True, tags_cap=1024, next_x=None):
        """processBulk(), :204-356.  x: [n] or [n_channels, n] complex64 on the GPU.
        Returns (status, out, tags): out holds the n_done published items (delayed input),
        tags a list (one per channel) of TAG_DTYPE records, index relative to out[0].
        next_x (optional look-ahead): the tensor the NEXT call will be given; its correlator
        then runs on a second stream behind this call's.  The caller keeps next_x alive and
        unchanged until that call (a device ring does)."""
        torch = _torch()
        x = _dev_c64(x)
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        assert x2.shape[0] == self.n_channels
        n_in = x2.shape[1]
        if next_x is not None:
            nx = _dev_c64(next_x)
            nx2 = nx.reshape(1, -1) if nx.dim() == 1 else nx
            assert nx2.shape[0] == self.n_channels
            check(lib().gr4pm_syncword_detection_hint_next(self._h, nx2.data_ptr(), nx2.stride(0), nx2.shape[1]),
                  "SyncwordDetection.hint_next")
        out = torch.empty_like(x2) if want_output else None
        n_done = C.c_size_t(0)
        tags = np.zeros((self.n_channels, tags_cap), dtype=TAG_DTYPE)
        n_tags = (C.c_size_t * self.n_channels)()
        st = lib().gr4pm_syncword_detection_process(
            self._h, x2.data_ptr(), x2.stride(0), n_in, out.data_ptr() if want_output else None,
            out.stride(0) if want_output else 0, C.byref(n_done), _np_ptr(tags), tags_cap, n_tags)
        check(st, "SyncwordDetection.processBulk")
        n = n_done.value
        tag_list = [tags[c, : n_tags[c]].copy() for c in range(self.n_channels)]
        if want_output:
            out = out[:, :n]
            if x.dim() == 1:
                out = out[0]
        if x.dim() == 1:
            tag_list = tag_list[0]
        return st, out, tag_list, n

    def correlate_only(self, x):
        """measurement hook: only the overlap-save correlator kernel (bench.py roofline leg)"""
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        check(lib().gr4pm_syncword_detection_correlate_only(self._h, x2.data_ptr(), x2.stride(0), x2.shape[1]),
              "correlate_only")

    def last_zpow(self, n_done):
        torch = _torch()
        z = torch.empty((self.n_channels, n_done), dtype=torch.float32, device="cuda")
        check(lib().gr4pm_syncword_detection_last_zpow(self._h, z.data_ptr(), z.stride(0)), "last_zpow")
        return z

    def _destroy(self):
        if getattr(self, "_h", None):
            _release("gr4pm_syncword_detection_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


class SyncwordDetectionFilter:
    """syncword_detection_filter.hpp:10-211"""

    def __init__(self, samples_per_symbol=4, syncword_size=64, header_size=128):
        self.samples_per_symbol, self.syncword_size, self.header_size = samples_per_symbol, syncword_size, header_size
        p = _abi.SdfParams(samples_per_symbol, syncword_size, header_size, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_syncword_detection_filter_create(C.byref(p), C.byref(self._h)), "SyncwordDetectionFilter")

    def start(self):
        check(lib().gr4pm_syncword_detection_filter_reset(self._h), "start")

    def process_bulk(self, x, out, head_tag_flags=0, headers=(), n_ignored=0):
        """one processBulk (:54-210). headers: sequence of packet_length or None (invalid_header).
        Returns (consumed, headers_consumed, ignored_consumed, tag_out_flags)."""
        x = _dev_c64(x)
        msgs = (_abi.HeaderMsg * max(len(headers), 1))()
        for i, hm in enumerate(headers):
            msgs[i].packet_length = 0 if hm is None else hm
            msgs[i].invalid_header = 1 if hm is None else 0
        c, hc, ic, tf = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        check(lib().gr4pm_syncword_detection_filter_process(
            self._h, x.data_ptr(), x.numel(), out.data_ptr(), out.numel(), head_tag_flags, msgs, len(headers),
            n_ignored, C.byref(c), C.byref(hc), C.byref(ic), C.byref(tf)), "SyncwordDetectionFilter.processBulk")
        return c.value, hc.value, ic.value, tf.value

    def gate(self, tag_index, headers=(), per_tag=False):
        """tag gate without the copy (device-resident chains): which syncword tags pass.
        headers[k] answers the k-th accepted tag (per_tag: tag k if accepted): packet_length, or
        None for invalid_header."""
        idx = np.ascontiguousarray(tag_index, dtype=np.uint64)
        msgs = np.zeros(max(len(headers), 1), dtype=_abi.HEADER_MSG_DTYPE)
        if isinstance(headers, np.ndarray) and headers.dtype == _abi.HEADER_MSG_DTYPE:
            msgs = _header_msgs(headers)[0]
        elif len(headers):
            if isinstance(headers, np.ndarray) and headers.dtype.kind in "iu":
                msgs["packet_length"][: len(headers)] = headers
            else:
                inval = np.fromiter((h is None for h in headers), dtype=bool, count=len(headers))
                msgs["invalid_header"][: len(headers)] = inval
                msgs["packet_length"][: len(headers)] = np.fromiter((0 if h is None else h for h in headers),
                                                                    dtype=np.uint64, count=len(headers))
        acc = np.zeros(max(idx.size, 1), dtype=np.uint8)
        used = C.c_size_t(0)
        check(lib().gr4pm_syncword_detection_filter_gate(self._h, _np_ptr(idx), idx.size, _np_ptr(msgs), len(headers),
                                                         1 if per_tag else 0, _np_ptr(acc), C.byref(used)),
              "SyncwordDetectionFilter.gate")
        return acc[: idx.size].astype(bool), used.value

    def gate_resolve(self, msg):
        """deliver the message of the packet the gate left open with a pending header"""
        m = np.ascontiguousarray(msg, dtype=_abi.HEADER_MSG_DTYPE).reshape(1)
        check(lib().gr4pm_syncword_detection_filter_gate_resolve(self._h, _np_ptr(m)), "SyncwordDetectionFilter.gate")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_syncword_detection_filter_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class _RotatorBase:
    def _create(self, mode, phase_incr, delay, n_channels):
        p = _abi.RotatorParams(mode, phase_incr, delay, n_channels, _stream_handle())
        self._h = C.c_void_p()
        self.n_channels = n_channels
        check(lib().gr4pm_rotator_create(C.byref(p), C.byref(self._h)), type(self).__name__)

    def start(self):
        check(lib().gr4pm_rotator_reset(self._h), "start")

    def process_bulk(self, x, tags=None, tag_channel=None):
        torch = _torch()
        x = _dev_c64(x)
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        out = torch.empty_like(x2)
        t = _tags_array(tags)
        tc = None if tag_channel is None else np.ascontiguousarray(tag_channel, dtype=np.uint32)
        check(lib().gr4pm_rotator_process(self._h, x2.data_ptr(), x2.stride(0), x2.shape[1], out.data_ptr(),
                                          _np_ptr(t), None if tc is None else _np_ptr(tc), t.size),
              type(self).__name__ + ".processBulk")
        return out.reshape(x.shape)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_rotator_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class Rotator(_RotatorBase):
    """rotator.hpp:20-65"""

    def __init__(self, phase_incr=0.0, n_channels=1):
        self.phase_incr = np.float32(phase_incr)
        self._create(0, float(self.phase_incr), 0, n_channels)


class CoarseFrequencyCorrection(_RotatorBase):
    """coarse_frequency_correction.hpp:20-99"""

    def __init__(self, delay=0, n_channels=1):
        self.delay = delay
        self._create(1, 0.0, delay, n_channels)


class CostasLoop:
    """costas_loop.hpp:15-149"""

    def __init__(self, loop_bandwidth=0.01, constellation="BPSK", n_channels=1):
        self.loop_bandwidth = loop_bandwidth
        self.constellation = constellation
        if constellation.upper() not in CONSTELLATIONS:
            raise Gr4pmError(f"unknown constellation {constellation}")  # enum_cast(...).value() throws
        p = _abi.CostasParams(loop_bandwidth, CONSTELLATIONS[constellation.upper()], n_channels, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_costas_loop_create(C.byref(p), C.byref(self._h)), "CostasLoop")

    @property
    def coeffs(self):
        k1, k2 = C.c_float(0), C.c_float(0)
        lib().gr4pm_costas_loop_coeffs(self._h, C.byref(k1), C.byref(k2))
        return k1.value, k2.value

    def settings_changed(self, loop_bandwidth=None, constellation=None):
        """settingsChanged(), :52-88"""
        if loop_bandwidth is not None:
            self.loop_bandwidth = loop_bandwidth
        if constellation is not None:
            self.constellation = constellation
        check(lib().gr4pm_costas_loop_set(self._h, self.loop_bandwidth, CONSTELLATIONS[self.constellation.upper()]),
              "settingsChanged")

    def process_bulk(self, x, tags=None, tag_channel=None):
        torch = _torch()
        x = _dev_c64(x)
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        out = torch.empty_like(x2)
        t = _tags_array(tags)
        tc = None if tag_channel is None else np.ascontiguousarray(tag_channel, dtype=np.uint32)
        check(lib().gr4pm_costas_loop_process(self._h, x2.data_ptr(), x2.stride(0), x2.shape[1], out.data_ptr(),
                                              _np_ptr(t), None if tc is None else _np_ptr(tc), t.size),
              "CostasLoop.processBulk")
        return out.reshape(x.shape)

    def process_packets(self, x, tags):
        """processBulk() behind PayloadMetadataInsert: tags are PACKET_TAG_DTYPE records whose
        "constellation" / "loop_bandwidth" keys update the settings (:52-88) from the tagged item
        on and whose syncword_phase sets the phase (:101-106)"""
        torch = _torch()
        x = _dev_c64(x)
        out = torch.empty_like(x)
        t = _ptags_array(tags)
        check(lib().gr4pm_costas_loop_process_packets(self._h, x.data_ptr(), x.numel(), out.data_ptr(), _np_ptr(t),
                                                      t.size), "CostasLoop.processBulk")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_costas_loop_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class SyncwordWipeoff:
    """syncword_wipeoff.hpp:12-91"""

    def __init__(self, syncword):
        self.syncword = np.ascontiguousarray(syncword, dtype=np.float32)
        p = _abi.WipeoffParams(_np_ptr(self.syncword), self.syncword.size, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_syncword_wipeoff_create(C.byref(p), C.byref(self._h)), "SyncwordWipeoff")

    def process_bulk(self, x, tags=None, in_place=False):
        """in_place: the syncword items of x itself are multiplied, nothing is copied (x is returned)"""
        torch = _torch()
        x = _dev_c64(x)
        out = x if in_place else torch.empty_like(x)
        t = _tags_array(tags)
        check(lib().gr4pm_syncword_wipeoff_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), _np_ptr(t),
                                                   t.size), "SyncwordWipeoff.processBulk")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_syncword_wipeoff_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class PayloadMetadataInsert:
    """payload_metadata_insert.hpp:12-324"""

    def __init__(self, syncword_size=64, header_size=128, syncword_costas_loop_bandwidth=0.02,
                 header_costas_loop_bandwidth=0.01, payload_costas_loop_bandwidth=0.005):
        self.syncword_size, self.header_size = syncword_size, header_size
        p = _abi.PmiParams(syncword_size, header_size, syncword_costas_loop_bandwidth, header_costas_loop_bandwidth,
                           payload_costas_loop_bandwidth, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_payload_metadata_insert_create(C.byref(p), C.byref(self._h)), "PayloadMetadataInsert")

    def start(self):
        check(lib().gr4pm_payload_metadata_insert_reset(self._h), "PayloadMetadataInsert.start")

    def resolve(self, msg):
        """per_tag mode: the message of the packet that was opened with a pending header"""
        m = np.ascontiguousarray(msg, dtype=_abi.HEADER_MSG_DTYPE).reshape(1)
        check(lib().gr4pm_payload_metadata_insert_resolve(self._h, _np_ptr(m)), "PayloadMetadataInsert.resolve")

    def process_bulk(self, x, tags=None, headers=(), out_cap=None, tags_cap=None, per_tag=False):
        """x: symbols with syncword tags (TAG_DTYPE); headers: the pending parsed_header messages
        (packet_length, None = invalid header), or with per_tag one message per tag.  Returns
        dict(out, tags, consumed, headers_used, ignored): consumed < len(x) where the block waits
        for a header message."""
        torch = _torch()
        x = _dev_c64(x)
        t = _tags_array(tags)
        out_cap = x.numel() if out_cap is None else out_cap
        out = torch.empty(max(out_cap, 1), dtype=x.dtype, device=x.device)
        tags_cap = 3 * t.size + 8 if tags_cap is None else tags_cap
        tout = np.empty(tags_cap, dtype=PACKET_TAG_DTYPE)  # the first n_tags records are written whole
        msgs, n_msgs = _header_msgs(headers)
        v = [C.c_size_t(0) for _ in range(5)]
        check(lib().gr4pm_payload_metadata_insert_process(
            self._h, x.data_ptr(), x.numel(), out.data_ptr(), out_cap, _np_ptr(t), t.size, _np_ptr(msgs),
            n_msgs, 1 if per_tag else 0, _np_ptr(tout), tags_cap, *[C.byref(c) for c in v]),
            "PayloadMetadataInsert.processBulk")
        n_tags, consumed, produced, used, ignored = [c.value for c in v]
        return {"out": out[:produced], "tags": tout[:n_tags], "consumed": consumed, "headers_used": used,
                "ignored": ignored}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_payload_metadata_insert_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class SyncwordRemove:
    """syncword_remove.hpp:11-112"""

    def __init__(self, syncword_size=64):
        self.syncword_size = syncword_size
        p = _abi.SyncwordRemoveParams(syncword_size, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_syncword_remove_create(C.byref(p), C.byref(self._h)), "SyncwordRemove")

    def process_bulk(self, x, tags=None):
        torch = _torch()
        x = _dev_c64(x)
        t = _ptags_array(tags)
        out = torch.empty(max(x.numel(), 1), dtype=x.dtype, device=x.device)
        tout = np.empty(t.size + 1, dtype=PACKET_TAG_DTYPE)
        nt, produced = C.c_size_t(0), C.c_size_t(0)
        check(lib().gr4pm_syncword_remove_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), _np_ptr(t),
                                                  t.size, _np_ptr(tout), tout.size, C.byref(nt), C.byref(produced)),
              "SyncwordRemove.processBulk")
        return out[: produced.value], tout[: nt.value]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_syncword_remove_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class ConstellationLLRDecoder:
    """constellation_llr_decoder.hpp:13-142"""

    def __init__(self, noise_sigma=1.0, constellation="BPSK"):
        self.noise_sigma, self.constellation = noise_sigma, constellation
        if constellation.upper() not in CONSTELLATIONS:
            raise Gr4pmError(f"unknown constellation {constellation}")
        p = _abi.LlrParams(noise_sigma, CONSTELLATIONS[constellation.upper()], _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_constellation_llr_decoder_create(C.byref(p), C.byref(self._h)),
              "ConstellationLLRDecoder")  # PILOT: "constellation not supported", :72-74

    def process_bulk(self, x, tags=None):
        """returns (llr float32 tensor, tags re-indexed to LLR positions)"""
        torch = _torch()
        x = _dev_c64(x)
        t = _ptags_array(tags)
        out = torch.empty(max(2 * x.numel(), 1), dtype=torch.float32, device=x.device)
        tout = np.empty(t.size + 1, dtype=PACKET_TAG_DTYPE)
        nt, produced = C.c_size_t(0), C.c_size_t(0)
        check(lib().gr4pm_constellation_llr_decoder_process(
            self._h, x.data_ptr(), x.numel(), out.data_ptr(), out.numel(), _np_ptr(t), t.size, _np_ptr(tout),
            tout.size, C.byref(nt), C.byref(produced)), "ConstellationLLRDecoder.processBulk")
        return out[: produced.value], tout[: nt.value]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_constellation_llr_decoder_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class AdditiveScrambler:
    """additive_scrambler.hpp:24-100 on float32 soft symbols or uint8 hard symbols (by the
    dtype of the first tensor it is given, or `dtype=`)"""

    def __init__(self, mask=0x8A, seed=0x7F, length=7, count=0, dtype="float32"):
        self.mask, self.seed, self.length, self.count = mask, seed, length, count
        kind = {"float32": 1, "uint8": 2}[str(dtype).replace("torch.", "")]
        p = _abi.ScramblerParams(mask, seed, length, count, kind, _stream_handle())
        self._kind = kind
        self._h = C.c_void_p()
        check(lib().gr4pm_additive_scrambler_create(C.byref(p), C.byref(self._h)), "AdditiveScrambler")

    def start(self):
        check(lib().gr4pm_additive_scrambler_reset(self._h), "AdditiveScrambler.start")

    def process_bulk(self, x, reset_index=()):
        """reset_index: items that carry the reset_tag_key (:78-80)"""
        torch = _torch()
        want = torch.float32 if self._kind == 1 else torch.uint8
        if not x.is_cuda or x.dtype != want:
            raise Gr4pmError(f"AdditiveScrambler expects a CUDA {want} tensor")
        x = x.contiguous()
        out = torch.empty_like(x)
        ri = np.ascontiguousarray(reset_index, dtype=np.uint64)
        check(lib().gr4pm_additive_scrambler_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), _np_ptr(ri),
                                                     ri.size), "AdditiveScrambler.processBulk")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_additive_scrambler_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class HeaderPayloadSplit:
    """header_payload_split.hpp:9-147 on float32 items (the header loop, packet_receiver.hpp:136-137) or complex64
    items (the symbol tap of zmq_output, :159-162: header_size 128, the payload tags carry payload_symbols)"""

    def __init__(self, header_size=256):
        self.header_size = header_size
        p = _abi.HeaderPayloadSplitParams(header_size, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_header_payload_split_create(C.byref(p), C.byref(self._h)), "HeaderPayloadSplit")

    def process_bulk(self, x, tags=None):
        """returns (header items, payload items, header tags, payload tags)"""
        torch = _torch()
        x = x.contiguous()
        assert x.is_cuda and x.dtype in (torch.float32, torch.complex64)
        t = _ptags_array(tags)
        hdr, pay = torch.empty(max(x.numel(), 1), dtype=x.dtype, device=x.device), torch.empty(
            max(x.numel(), 1), dtype=x.dtype, device=x.device)
        ht, pt = np.empty(t.size + 1, dtype=PACKET_TAG_DTYPE), np.empty(t.size + 1, dtype=PACKET_TAG_DTYPE)
        v = [C.c_size_t(0) for _ in range(4)]
        fn = (lib().gr4pm_header_payload_split_process if x.dtype == torch.float32
              else lib().gr4pm_header_payload_split_process_c64)
        check(fn(
            self._h, x.data_ptr(), x.numel(), hdr.data_ptr(), C.byref(v[0]), pay.data_ptr(), C.byref(v[1]),
            _np_ptr(t), t.size, _np_ptr(ht), C.byref(v[2]), _np_ptr(pt), C.byref(v[3]), t.size + 1),
            "HeaderPayloadSplit.processBulk")
        return hdr[: v[0].value], pay[: v[1].value], ht[: v[2].value], pt[: v[3].value]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_header_payload_split_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


def header_ldpc_alist():
    """the (128, 32) parity-check matrix of the header code in alist form
    (header_fec_decoder.hpp:31-258), shipped as data/header_ldpc_128_32.alist"""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return open(os.path.join(here, "data", "header_ldpc_128_32.alist")).read()


class HeaderFecDecoder:
    """header_fec_decoder.hpp:13-359: 256 LLRs -> 4 header bytes, or invalid"""

    def __init__(self, alist=None, max_iterations=25, arithmetic=0):
        """arithmetic: 0 float32 messages (default), 1 8-bit messages (include/gr4pm_hip.h)"""
        self._alist = (alist or header_ldpc_alist()).encode()
        p = _abi.HeaderFecDecoderParams(self._alist, max_iterations, _stream_handle(), arithmetic)
        self._h = C.c_void_p()
        check(lib().gr4pm_header_fec_decoder_create(C.byref(p), C.byref(self._h)), "HeaderFecDecoder.start")

    def process_bulk(self, llrs):
        """llrs: CUDA float32, 256 per codeword.  Returns (headers [n, 4] uint8, invalid [n] bool)"""
        torch = _torch()
        llrs = llrs.contiguous()
        assert llrs.is_cuda and llrs.dtype == torch.float32
        n = llrs.numel() // 256
        out = np.empty((max(n, 1), 4), dtype=np.uint8)
        inval = np.empty(max(n, 1), dtype=np.uint8)
        check(lib().gr4pm_header_fec_decoder_process(self._h, llrs.data_ptr(), n, _np_ptr(out), _np_ptr(inval)),
              "HeaderFecDecoder.processBulk")
        return out[:n], inval[:n].astype(bool)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_header_fec_decoder_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


def header_parse(headers, invalid=None):
    """HeaderParser, header_parser.hpp:46-95: [n, 4] bytes (+ the decoder's verdicts) -> the
    parsed_header messages as a HEADER_MSG_DTYPE array, and the packet types (-1 invalid)"""
    hd = np.ascontiguousarray(headers, dtype=np.uint8).reshape(-1, 4)
    inv = np.zeros(hd.shape[0], dtype=np.uint8) if invalid is None else np.ascontiguousarray(invalid, dtype=np.uint8)
    msgs = np.zeros(max(hd.shape[0], 1), dtype=_abi.HEADER_MSG_DTYPE)
    ptype = np.zeros(max(hd.shape[0], 1), dtype=np.int32)
    lib().gr4pm_header_parse(_np_ptr(hd), _np_ptr(inv), hd.shape[0], _np_ptr(msgs), _np_ptr(ptype))
    return msgs[: hd.shape[0]], ptype[: hd.shape[0]]


def _item_kind(x):
    torch = _torch()
    if x.dtype == torch.complex64:
        return 0
    if x.dtype == torch.float32:
        return 1
    raise TypeError("items must be complex64 or float32")


class InterpolatingFirFilter:
    """interpolating_fir_filter.hpp:14-103 (TIn = TOut in {c64, float}, TTaps = float)"""

    def __init__(self, interpolation, taps, item_dtype="complex64"):
        self.interpolation = interpolation
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.item_kind = 0 if item_dtype == "complex64" else 1
        p = _abi.InterpFirParams(interpolation, _np_ptr(self.taps), self.taps.size, self.item_kind, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_interp_fir_create(C.byref(p), C.byref(self._h)), "InterpolatingFirFilter")

    def process_bulk(self, x):
        torch = _torch()
        assert x.is_cuda and x.is_contiguous() and _item_kind(x) == self.item_kind
        out = torch.empty(x.numel() * self.interpolation, dtype=x.dtype, device=x.device)
        check(lib().gr4pm_interp_fir_process(self._h, x.data_ptr(), x.numel(), out.data_ptr()),
              "InterpolatingFirFilter.processBulk")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_interp_fir_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class SymbolFilter:
    """symbol_filter.hpp:13-253"""

    def __init__(self, taps, num_arms, samples_per_symbol, delay=0, item_dtype="complex64"):
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.num_arms, self.samples_per_symbol, self.delay = num_arms, samples_per_symbol, delay
        self.item_kind = 0 if item_dtype == "complex64" else 1
        p = _abi.SymbolFilterParams(samples_per_symbol, _np_ptr(self.taps), self.taps.size, num_arms, delay,
                                    self.item_kind, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_symbol_filter_create(C.byref(p), C.byref(self._h)), "SymbolFilter")

    def start(self):
        check(lib().gr4pm_symbol_filter_reset(self._h), "start")

    def process_bulk(self, x, tags=None, out_cap=None):
        """returns (symbols, tags_out, consumed)"""
        torch = _torch()
        assert x.is_cuda and x.is_contiguous() and _item_kind(x) == self.item_kind
        t = _tags_array(tags)
        if out_cap is None:
            out_cap = x.numel() // self.samples_per_symbol + t.size + 2
        out = torch.empty(max(out_cap, 1), dtype=x.dtype, device=x.device)
        tout = np.zeros(t.size + 64, dtype=TAG_DTYPE)
        nto, cons, prod = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        check(lib().gr4pm_symbol_filter_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), out_cap, _np_ptr(t),
                                                t.size, _np_ptr(tout), tout.size, C.byref(nto), C.byref(cons),
                                                C.byref(prod)), "SymbolFilter.processBulk")
        return out[: prod.value], tout[: nto.value].copy(), cons.value

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_symbol_filter_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


def cfc_symbol_filter(cfc, symf, x, tags=None, out_cap=None):
    """CoarseFrequencyCorrection -> SymbolFilter in one pass (gr4pm_cfc_symbol_filter_process):
    identical results and state updates, the rotated stream is not materialised.
    Returns (symbols, tags_out, consumed)."""
    torch = _torch()
    x = _dev_c64(x)
    t = _tags_array(tags)
    if out_cap is None:
        out_cap = x.numel() // symf.samples_per_symbol + t.size + 2
    out = torch.empty(max(out_cap, 1), dtype=x.dtype, device=x.device)
    tout = np.zeros(t.size + 64, dtype=TAG_DTYPE)
    nto, cons, prod = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    check(lib().gr4pm_cfc_symbol_filter_process(cfc._h, symf._h, x.data_ptr(), x.numel(), out.data_ptr(), out_cap,
                                                _np_ptr(t), t.size, _np_ptr(tout), tout.size, C.byref(nto),
                                                C.byref(cons), C.byref(prod)), "cfc_symbol_filter")
    return out[: prod.value], tout[: nto.value].copy(), cons.value


def cfc_symbol_filter_plan(cfc, n_in, tags=None):
    """first half of cfc_symbol_filter (gr4pm_cfc_symbol_filter_plan): the CFC's tag handling and the
    phasor checkpoints of a call of n_in items; returns the plan id for cfc_symbol_filter_run"""
    t = _tags_array(tags)
    plan = C.c_int(-1)
    check(lib().gr4pm_cfc_symbol_filter_plan(cfc._h, n_in, _np_ptr(t), t.size, C.byref(plan)), "cfc_symbol_filter_plan")
    return plan.value


def cfc_symbol_filter_run(cfc, plan, symf, x, tags=None, out_cap=None):
    """second half (gr4pm_cfc_symbol_filter_run): the filter itself, on the SymbolFilter's stream;
    the caller makes sure the plan has completed.  Returns (symbols, tags_out, consumed)."""
    torch = _torch()
    x = _dev_c64(x)
    t = _tags_array(tags)
    if out_cap is None:
        out_cap = x.numel() // symf.samples_per_symbol + t.size + 2
    out = torch.empty(max(out_cap, 1), dtype=x.dtype, device=x.device)
    tout = np.zeros(t.size + 64, dtype=TAG_DTYPE)
    nto, cons, prod = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    check(lib().gr4pm_cfc_symbol_filter_run(cfc._h, plan, symf._h, x.data_ptr(), x.numel(), out.data_ptr(), out_cap,
                                            _np_ptr(t), t.size, _np_ptr(tout), tout.size, C.byref(nto),
                                            C.byref(cons), C.byref(prod)), "cfc_symbol_filter_run")
    return out[: prod.value], tout[: nto.value].copy(), cons.value


class PfbArbResampler:
    """pfb_arb_resampler.hpp:23-183 (<c64, c64, float, TRate>)"""

    def __init__(self, rate=1.0, taps=None, filter_size=32, rate_dtype="float32"):
        if taps is None:
            from . import default_pfb_arb_taps
            taps = default_pfb_arb_taps()
        self.rate, self.filter_size = rate, filter_size
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        p = _abi.PfbArbParams(rate, 1 if rate_dtype == "float64" else 0, _np_ptr(self.taps), self.taps.size,
                              filter_size, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_pfb_arb_resampler_create(C.byref(p), C.byref(self._h)), "PfbArbResampler")

    def process_bulk(self, x, out_cap=None):
        """returns (out, consumed)"""
        torch = _torch()
        x = _dev_c64(x)
        if out_cap is None:
            out_cap = int(x.numel() * self.rate) + 64
        out = torch.empty(out_cap, dtype=x.dtype, device=x.device)
        cons, prod = C.c_size_t(0), C.c_size_t(0)
        check(lib().gr4pm_pfb_arb_resampler_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), out_cap,
                                                    C.byref(cons), C.byref(prod)), "PfbArbResampler.processBulk")
        return out[: prod.value], cons.value

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_pfb_arb_resampler_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


# CCSDS syncword of the modem, packet_receiver.hpp:45-59
SYNCWORD = np.array(
    [0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 1, 1, 1,
     0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0],
    dtype=np.uint8)


def binary_slicer(x, invert=False):
    """BinarySlicer<invert>, binary_slicer.hpp:10-35: float32 soft symbols -> uint8 hard symbols"""
    torch = _torch()
    x = x.contiguous()
    assert x.is_cuda and x.dtype == torch.float32
    out = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
    check(lib().gr4pm_binary_slicer_process(x.data_ptr(), x.numel(), out.data_ptr(), 1 if invert else 0,
                                            _stream_handle()), "BinarySlicer")
    return out


def pack_bits(x, inputs_per_output=8, bits_per_input=1, msb_first=True):
    """PackBits<MSB|LSB, uint8_t, uint8_t>, pack_bits.hpp"""
    torch = _torch()
    x = x.contiguous()
    assert x.is_cuda and x.dtype == torch.uint8
    if x.numel() % inputs_per_output:
        raise Gr4pmError("input size not divisible by inputs_per_output")
    n_out = x.numel() // inputs_per_output
    out = torch.empty(n_out, dtype=torch.uint8, device=x.device)
    check(lib().gr4pm_pack_bits_process(x.data_ptr(), n_out, out.data_ptr(), inputs_per_output, bits_per_input,
                                        1 if msb_first else 0, _stream_handle()), "PackBits")
    return out


def slice_pack(llr):
    """BinarySlicer<true> + PackBits<>(8, 1) in one kernel (packet_receiver.hpp:140-144)"""
    torch = _torch()
    llr = llr.contiguous()
    assert llr.is_cuda and llr.dtype == torch.float32 and llr.numel() % 8 == 0
    out = torch.empty(llr.numel() // 8, dtype=torch.uint8, device=llr.device)
    check(lib().gr4pm_slice_pack_process(llr.data_ptr(), out.numel(), out.data_ptr(), _stream_handle()), "slice_pack")
    return out


class CrcCheck:
    """crc_check.hpp:22-239 (defaults = CRC-32, :61-66)"""

    def __init__(self, num_bits=32, poly=0x4C11DB7, initial_value=0xFFFFFFFF, final_xor=0xFFFFFFFF,
                 input_reflected=True, result_reflected=True, swap_endianness=False, discard_crc=False,
                 skip_header_bytes=0):
        p = _abi.CrcCheckParams(num_bits, poly, initial_value, final_xor, int(input_reflected), int(result_reflected),
                                int(swap_endianness), int(discard_crc), skip_header_bytes, _stream_handle())
        self._h = C.c_void_p()
        check(lib().gr4pm_crc_check_create(C.byref(p), C.byref(self._h)), "CrcCheck")

    def compute(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint8)
        return int(lib().gr4pm_crc_check_compute(self._h, _np_ptr(d), d.size))

    def process_bulk(self, x, packet_len, packet_offset=None):
        """x: CUDA uint8 packets ("packet_len" tags = packet_len[], laid back to back unless
        packet_offset is given).  Returns (bytes of the packets that pass, out_len per packet)"""
        torch = _torch()
        x = x.contiguous()
        assert x.is_cuda and x.dtype == torch.uint8
        pl = np.ascontiguousarray(packet_len, dtype=np.uint64)
        po = (np.concatenate([[0], np.cumsum(pl)[:-1]]).astype(np.uint64) if packet_offset is None
              else np.ascontiguousarray(packet_offset, dtype=np.uint64))
        out = torch.empty(max(x.numel(), 1), dtype=torch.uint8, device=x.device)
        ol = np.zeros(max(pl.size, 1), dtype=np.uint64)
        n = C.c_size_t(0)
        check(lib().gr4pm_crc_check_process(self._h, x.data_ptr(), _np_ptr(po), _np_ptr(pl), pl.size, out.data_ptr(),
                                            _np_ptr(ol), C.byref(n)), "CrcCheck.processBulk")
        return out[: n.value], ol[: pl.size]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_crc_check_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class HeaderDecoder:
    """The header decode loop of packet_receiver.hpp:131-139 as one unit: AdditiveScrambler<float>
    (CCSDS 131.0-B-5 polynomial, reset at every "header_start") -> HeaderPayloadSplit ->
    HeaderFecDecoder -> HeaderParser.  Fed with the LLR stream and its tags
    (ConstellationLLRDecoder output); a header whose 256 LLRs end in a later call is finished
    there."""

    def __init__(self):
        self.descrambler = AdditiveScrambler(0x4001, 0x18E38, 16)  # :131-135
        self.header_payload_split = HeaderPayloadSplit(256)        # :136-137
        self.header_fec_decoder = HeaderFecDecoder()               # :138
        self._partial = None

    def process_bulk(self, llr, llr_tags):
        """returns dict(messages: HEADER_MSG_DTYPE per finished header, packet_type, header_bytes,
        invalid, payload_llr: descrambled payload LLRs, payload_tags)"""
        torch = _torch()
        t = _ptags_array(llr_tags)
        resets = t["index"][t["kind"] == _abi.PKT_HEADER_START]
        d = self.descrambler.process_bulk(llr, resets)
        hdr, pay, _, pay_tags = self.header_payload_split.process_bulk(d, t)
        if self._partial is not None:
            hdr = torch.cat([self._partial, hdr])
        n = hdr.numel() // 256
        self._partial = hdr[n * 256:].clone() if hdr.numel() % 256 else None
        hb, inval = self.header_fec_decoder.process_bulk(hdr[: n * 256])
        msgs, ptype = header_parse(hb, inval)
        return {"messages": msgs, "packet_type": ptype, "header_bytes": hb, "invalid": inval, "payload_llr": pay,
                "payload_tags": pay_tags}


class PacketReceiver:
    """The sample-rate / symbol-rate front half of gr::packet_modem::PacketReceiver
    (packet_receiver.hpp:34-127,191-232): SyncwordDetection -> SyncwordDetectionFilter ->
    CoarseFrequencyCorrection -> SymbolFilter -> SyncwordWipeoff -> CostasLoop, with every
    constant the reference constructor fixes.  Everything after CostasLoop (header decode,
    LDPC, CRC) is outside the hot path; the `parsed_header` feedback that the filter waits for
    is supplied by the caller (`header_fn(tag) -> packet_length | None`, or a constant).

    pipelined=True runs the chain as the reference's multi-threaded scheduler does (one worker
    per block, benchmarks/README.md:8-26), here as three stages on three HIP streams --
    detector | frequency correction + symbol filter + wipe-off | Costas loop -- so that the
    per-packet serial kernels (phasor checkpoints, PLL), which occupy only a few CUs, overlap
    the detector of the following batches.  process_bulk() then returns the result of an
    EARLIER batch (None while the pipeline fills); flush() drains it.

    soft_bits=True continues as packet_receiver.hpp:123-131 does: SyncwordWipeoff ->
    PayloadMetadataInsert (drops everything between packets, marks syncword / header / payload
    with "constellation" and "loop_bandwidth" tags) -> CostasLoop (those tags drive its settings)
    -> SyncwordRemove -> ConstellationLLRDecoder (noise_sigma 0.7, QPSK): the result then also
    carries "llr" (float32, two per symbol) and "llr_tags".

    decode_headers=True (implies soft_bits) closes the header feedback loop of
    packet_receiver.hpp:131-139,233-247 on the device instead of asking the caller for
    `header_fn`: descrambler -> HeaderPayloadSplit -> HeaderFecDecoder -> HeaderParser supply the
    parsed_header messages.  The reference resolves that loop packet by packet; a batch resolves
    it in two passes: (A) every detection is taken through a second set of the same blocks up to
    its header LLRs (its payload is dropped by answering "invalid_header"), which yields every
    candidate's header; (B) the real
    chain runs with those messages, and the headers it decodes itself are compared with the ones
    it was given (`header_mismatches` in the result; its payload LLRs leave descrambled as
    "payload_llr" / "payload_tags").  Pass A runs on a compact stream of one 912-sample window
    per detection, so it costs a few percent of the real pass.  A header whose symbols continue in
    the next batch stays pending until then.  The payload tail follows (:140-147): BinarySlicer<true>,
    PackBits, CrcCheck(discard_crc) -- "packets" holds the bytes of the packets whose CRC-32 matches,
    "packet_lengths" their lengths (0 = dropped)."""

    def __init__(self, samples_per_symbol=4, syncword_freq_bins=4, syncword_threshold=9.5,
                 costas_constellation="QPSK", max_items=1 << 22, pipelined=False, fused=True,
                 soft_bits=False, decode_headers=False, detector=True):
        """detector=False: only the blocks behind SyncwordDetection are created (the caller feeds
        _stage1 / _stage2 from a detector of its own: MultiChannelPacketReceiver)"""
        torch = _torch()
        self.fused = fused  # CFC applied while the symbol filter stages its input
        self.decode_headers = decode_headers
        soft_bits = soft_bits or decode_headers
        self.soft_bits = soft_bits
        sps = samples_per_symbol
        self.samples_per_symbol = sps
        rrc = root_raised_cosine(1.0, float(sps), 1.0, 0.35, sps * 11)            # :60-65
        norm = np.float32(0.0)
        for v in rrc:                                                            # :67-74 (float accumulate)
            norm = np.float32(norm + np.float32(v * v))
        norm = np.float32(np.sqrt(norm))
        self.rrc_taps = (rrc / norm).astype(np.float32)
        bpsk = np.array([1, -1], dtype=np.complex64)
        self.pipelined = pipelined
        cur = torch.cuda.current_stream()
        self._streams = [torch.cuda.Stream() for _ in range(3)] if pipelined else [cur, cur, cur]
        self.syncword_detection = None
        if detector:
            with torch.cuda.stream(self._streams[0]):
                self.syncword_detection = SyncwordDetection(                      # :76-83
                    self.rrc_taps, SYNCWORD, bpsk, -syncword_freq_bins, syncword_freq_bins,
                    samples_per_symbol=sps, power_threshold=syncword_threshold, max_items=max_items)
        with torch.cuda.stream(self._streams[1]):
            self.syncword_detection_filter = SyncwordDetectionFilter(sps)         # :84-85
            self.freq_correction = CoarseFrequencyCorrection((self.rrc_taps.size - 1) // 2 + sps)  # :94-95
            arms = 32                                                             # :96
            pfb = root_raised_cosine(float(arms) / float(norm), float(arms * sps), 1.0, 0.35, arms * sps * 11)[:-1]  # :100-110
            self.symbol_filter = SymbolFilter(pfb, arms, sps, self.rrc_taps.size - 1)  # :111-115
            self.syncword_wipeoff = SyncwordWipeoff(np.where(SYNCWORD == 1, -1.0, 1.0).astype(np.float32))  # :117-122
            if decode_headers:  # pass A: the same blocks again, up to the header LLRs
                self._spec = {
                    "cfc": CoarseFrequencyCorrection((self.rrc_taps.size - 1) // 2 + sps),
                    "symf": SymbolFilter(pfb, arms, sps, self.rrc_taps.size - 1),
                    "wipeoff": SyncwordWipeoff(np.where(SYNCWORD == 1, -1.0, 1.0).astype(np.float32)),
                    "pmi": PayloadMetadataInsert(),
                    "costas": CostasLoop(),
                    "remove": SyncwordRemove(),
                    "llr": ConstellationLLRDecoder(0.7, "QPSK"),
                    "headers": HeaderDecoder(),
                }
                self._awaiting = np.zeros(0, dtype=np.uint64)   # detections whose window continues next batch
                self._awaiting_tags = np.zeros(0, dtype=TAG_DTYPE)
                self._spec_order = np.zeros(0, dtype=np.uint64)  # detections in pass A, header not out yet
                self._spec_fifo = np.zeros(0, dtype=_abi.HEADER_MSG_DTYPE)
                self._tail = torch.zeros(self._SPEC_W + self._SPEC_PRE, dtype=torch.complex64, device="cuda")
                self._known_idx = np.zeros(0, dtype=np.uint64)  # candidates with a decoded header ...
                self._known_msg = np.zeros(0, dtype=_abi.HEADER_MSG_DTYPE)  # ... and the message
                self._pending_real = None                       # accepted packet waiting for its header
        with torch.cuda.stream(self._streams[2]):
            self.costas_loop = CostasLoop(0.01, costas_constellation)             # :125
            if soft_bits:
                self.payload_metadata_insert = PayloadMetadataInsert()            # :123-124
                self.costas_loop = CostasLoop()                                   # :125 (BPSK until the first tag)
                self.syncword_remove = SyncwordRemove()                           # :126
                self.constellation_decoder = ConstellationLLRDecoder(0.7, "QPSK")  # :129-130
            if decode_headers:
                self.header_decoder = HeaderDecoder()                             # :131-139
                self.payload_crc_check = CrcCheck(discard_crc=True)               # :145-147
                self._payload_carry = None   # soft bits of a payload that continues in the next batch
                self._payload_lens = np.zeros(0, dtype=np.uint64)  # bits of the payloads not finished yet
                self._used_msgs = np.zeros(0, dtype=_abi.HEADER_MSG_DTYPE)        # given to pass B, not yet verified
        # messages of accepted tags on their way to PayloadMetadataInsert: the symbol filter can
        # hold a tag of the last few samples back until the next call (symbol_filter.hpp:204-228)
        self._hdr_fifo = np.zeros(0, dtype=_abi.HEADER_MSG_DTYPE)
        self._workers = None
        self._inflight = []
        if pipelined:
            import concurrent.futures
            # the current device is a per-thread setting (torch and HIP alike): the workers take the creator's
            dev = _torch().cuda.current_device()
            self._workers = [concurrent.futures.ThreadPoolExecutor(max_workers=1, initializer=_torch().cuda.set_device,
                                                                   initargs=(dev,)) for _ in range(2)]

    # ---- the three stages
    def _stage0(self, x, tags_cap, history, next_x=None):
        torch = _torch()
        with torch.cuda.stream(self._streams[0]):
            st, y, det_tags, n = self.syncword_detection.process_bulk(x, want_output=history is None,
                                                                      tags_cap=tags_cap, next_x=next_x)
        if history is not None:
            # device-ring input: the delayed stream (hpp:318-319: out[i] = in[i - (2T+1)]) is read in
            # place from the ring instead of being copied
            d = 2 * self.syncword_detection.time_threshold + 1
            assert history.numel() >= d and history.data_ptr() + history.numel() * 8 == x.data_ptr(), \
                "history must be the ring contents that directly precede x"
            # as_strided takes the offset inside the STORAGE, not inside the view
            y = torch.as_strided(history, (n,), (1,), history.storage_offset() + history.numel() - d)
        base = self.syncword_detection._items_consumed - n        # absolute index of y[0]
        return st, y, det_tags, n, base

    def _stage1(self, st, y, det_tags, n, base, header_fn):
        torch = _torch()
        if st != 0:
            return {"status": st, "consumed": 0, "symbols": None, "tags": det_tags, "detector_tags": det_tags}
        if self.decode_headers:
            return self._stage1_decode(y, det_tags, n, base)
        with torch.cuda.stream(self._streams[1]):
            # SyncwordDetectionFilter: gate the tags; the samples pass unchanged
            if callable(header_fn):
                headers = [header_fn(t) for t in det_tags]
            elif header_fn is None:  # every header invalid
                headers = [None] * det_tags.size
            else:  # a constant packet_length
                headers = np.full(det_tags.size, int(header_fn), dtype=np.uint64)
            acc, _ = self.syncword_detection_filter.gate(base + det_tags["index"], headers, per_tag=True)
            tags = det_tags[acc]
            if self.fused:
                sym, sym_tags, consumed = cfc_symbol_filter(self.freq_correction, self.symbol_filter, y, tags)
            else:
                z = self.freq_correction.process_bulk(y, tags)
                sym, sym_tags, consumed = self.symbol_filter.process_bulk(z, tags)
            w = self.syncword_wipeoff.process_bulk(sym, sym_tags)
            self._hdr_fifo = np.concatenate([self._hdr_fifo, _header_msgs(headers)[0][: det_tags.size][acc]])
            hdrs, self._hdr_fifo = self._hdr_fifo[: sym_tags.size], self._hdr_fifo[sym_tags.size:]
        return {"status": 0, "consumed": n, "symbols": w, "tags": sym_tags, "detector_tags": det_tags,
                "accepted": acc, "headers": hdrs}

    # pass A works on a compact stream: one window of _SPEC_W samples per detection, starting
    # _SPEC_PRE samples before the tagged one (room for the matched filter's history) and long
    # enough for syncword + header behind the filter delay: 16 + 11 + 192 < 228 symbols
    _SPEC_PRE = 64
    _SPEC_W = 912

    def _predecode(self, y, det_tags, idx_abs, base):
        """pass A: the header of every detection (see the class docstring); updates the table of
        known headers.  A detection whose window runs past the end of this batch waits in
        self._awaiting and is decoded with the next batch (its first samples come from the saved
        tail of this one)."""
        torch = _torch()
        sp = self._spec
        n, W, pre = y.numel(), self._SPEC_W, self._SPEC_PRE
        # window starts relative to y[0]: waiting detections of the last batch first (negative)
        old = self._awaiting.astype(np.int64) - np.int64(base) - pre
        new = det_tags["index"].astype(np.int64) - pre
        fits = new + W <= n
        starts = np.concatenate([old, new[fits]])
        tags = np.concatenate([self._awaiting_tags, det_tags[fits]])
        self._awaiting, self._awaiting_tags = idx_abs[~fits], det_tags[~fits].copy()
        k = starts.size
        if k:
            st = torch.from_numpy(starts).to(y.device)
            ar = torch.arange(W, device=y.device)
            inside = st >= 0
            compact = torch.empty((k, W), dtype=y.dtype, device=y.device)
            if bool(inside.all()):
                compact = y[st[:, None] + ar]
            else:  # a few windows begin in the previous batch: read them from [saved tail | head of y]
                head = torch.cat([self._tail, y[: W]])
                compact[inside] = y[st[inside][:, None] + ar]
                compact[~inside] = head[(st[~inside] + self._tail.numel())[:, None] + ar]
            ctags = tags.copy()
            ctags["index"] = np.arange(k, dtype=np.uint64) * np.uint64(W) + np.uint64(pre)
            inv = np.zeros(k, dtype=_abi.HEADER_MSG_DTYPE)
            inv["invalid_header"] = 1
            sym, sym_tags, _ = cfc_symbol_filter(sp["cfc"], sp["symf"], compact.reshape(-1), ctags)
            w = sp["wipeoff"].process_bulk(sym, sym_tags)
            self._spec_fifo = np.concatenate([self._spec_fifo, inv])
            hdrs, self._spec_fifo = self._spec_fifo[: sym_tags.size], self._spec_fifo[sym_tags.size:]
            pm = sp["pmi"].process_bulk(w, sym_tags, hdrs, per_tag=True)
            z = sp["costas"].process_packets(pm["out"], pm["tags"])
            d, dt = sp["remove"].process_bulk(z, pm["tags"])
            llr, lt = sp["llr"].process_bulk(d, dt)
            done = sp["headers"].process_bulk(llr, lt)["messages"]
            self._spec_order = np.concatenate([self._spec_order, np.concatenate([
                (old + np.int64(base) + pre).astype(np.uint64), idx_abs[fits]])])
            m = done.size
            self._known_idx = np.concatenate([self._known_idx, self._spec_order[:m]])
            self._known_msg = np.concatenate([self._known_msg, done])
            self._spec_order = self._spec_order[m:]
            order = np.argsort(self._known_idx, kind="stable")
            self._known_idx, self._known_msg = self._known_idx[order], self._known_msg[order]
        self._tail = y[-(W + pre):].clone()

    def _stage1_decode(self, y, det_tags, n, base):
        torch = _torch()
        with torch.cuda.stream(self._streams[1]):
            idx_abs = (base + det_tags["index"]).astype(np.uint64)
            self._predecode(y, det_tags, idx_abs, base)
            resolve = None
            if self._pending_real is not None:
                j = np.searchsorted(self._known_idx, self._pending_real)
                if j < self._known_idx.size and self._known_idx[j] == self._pending_real:
                    resolve = self._known_msg[j].copy()
                    self.syncword_detection_filter.gate_resolve(resolve)
                    self._pending_real = None
                    waiting = np.nonzero(self._hdr_fifo["invalid_header"] == 2)[0]
                    if waiting.size:  # its tag has not even reached PayloadMetadataInsert yet
                        self._hdr_fifo[waiting[0]] = resolve
                        resolve = None
            # per-tag messages: decoded / still on its way (2) / never decoded (1)
            msgs = np.zeros(det_tags.size, dtype=_abi.HEADER_MSG_DTYPE)
            msgs["invalid_header"] = 1
            j = np.searchsorted(self._known_idx, idx_abs)
            jc = np.minimum(j, max(self._known_idx.size - 1, 0))
            hit = (j < self._known_idx.size) & (self._known_idx[jc] == idx_abs) if self._known_idx.size else \
                np.zeros(det_tags.size, dtype=bool)
            msgs[hit] = self._known_msg[jc[hit]]
            msgs["invalid_header"][np.isin(idx_abs, self._awaiting) | np.isin(idx_abs, self._spec_order)] = 2
            acc, _ = self.syncword_detection_filter.gate(idx_abs, msgs, per_tag=True)
            tags, headers = det_tags[acc], msgs[acc]
            if headers.size and headers["invalid_header"][-1] == 2:
                self._pending_real = idx_abs[acc][-1]
            keep = self._known_idx + np.uint64(1 << 22) >= np.uint64(base)  # forget old entries
            self._known_idx, self._known_msg = self._known_idx[keep], self._known_msg[keep]
            sym, sym_tags, consumed = cfc_symbol_filter(self.freq_correction, self.symbol_filter, y, tags)
            w = self.syncword_wipeoff.process_bulk(sym, sym_tags)
            self._hdr_fifo = np.concatenate([self._hdr_fifo, headers])
            hdrs, self._hdr_fifo = self._hdr_fifo[: sym_tags.size], self._hdr_fifo[sym_tags.size:]
        return {"status": 0, "consumed": n, "symbols": w, "tags": sym_tags, "detector_tags": det_tags,
                "accepted": acc, "headers": hdrs, "resolve": resolve}

    def _stage2(self, res):
        torch = _torch()
        if res["status"] != 0:
            return res
        with torch.cuda.stream(self._streams[2]):
            if not self.soft_bits:
                res["symbols"] = self.costas_loop.process_bulk(res["symbols"], res["tags"])
                return res
            early_mismatch = 0
            if res.get("resolve") is not None:
                self.payload_metadata_insert.resolve(res["resolve"])
                if self.decode_headers:
                    waiting = np.nonzero(self._used_msgs["invalid_header"] == 2)[0]
                    early = getattr(self, "_early_hdrs", [])
                    if waiting.size:
                        self._used_msgs[waiting[0]] = res["resolve"]
                    elif early:  # the chain's own decode of that header came first: checked now
                        got0, r = early.pop(0), res["resolve"]
                        if not (r["invalid_header"] == got0["invalid_header"] and
                                (got0["invalid_header"] == 1 or r["packet_length"] == got0["packet_length"])):
                            early_mismatch = 1
            sym_in, tags_in, hdrs_in = res["symbols"], res["tags"], res["headers"]
            carry = getattr(self, "_pm_carry", None)
            if carry is not None:  # symbols PayloadMetadataInsert could not take in the batch before (below)
                csym, ctags, chdrs = carry
                tags_in = tags_in.copy()
                tags_in["index"] += csym.numel()
                sym_in = torch.cat([csym, sym_in])
                tags_in = np.concatenate([ctags, tags_in])
                hdrs_in = np.concatenate([chdrs, hdrs_in])
                self._pm_carry = None
            pm = self.payload_metadata_insert.process_bulk(sym_in, tags_in, hdrs_in, per_tag=True)
            if pm["consumed"] != sym_in.numel():
                # the block waits for a header that pass A delivers with the next batch (a batch that ends 816 .. 848 items
                # behind a syncword, payload_metadata_insert.hpp:243-247): the rest is carried, as in the native receiver
                rest = sym_in.numel() - pm["consumed"]
                if rest > 1024:
                    raise Gr4pmError(f"PayloadMetadataInsert stalled at symbol {pm['consumed']} of {sym_in.numel()}: "
                                     "a header message is missing")
                keep = tags_in["index"] >= pm["consumed"]
                ctags = tags_in[keep].copy()
                ctags["index"] -= pm["consumed"]
                self._pm_carry = (sym_in[pm["consumed"]:].clone(), ctags, hdrs_in[keep].copy())
            res["tags"], res["headers"] = tags_in, hdrs_in  # (what the bookkeeping below refers to)
            z = self.costas_loop.process_packets(pm["out"], pm["tags"])
            data, data_tags = self.syncword_remove.process_bulk(z, pm["tags"])
            llr, llr_tags = self.constellation_decoder.process_bulk(data, data_tags)
            res.update(symbols=z, packet_tags=pm["tags"], llr=llr, llr_tags=llr_tags,
                       ignored_syncwords=pm["ignored"])
            if self.decode_headers:
                # the chain's own header decode (packet_receiver.hpp:131-139): descrambled payload
                # LLRs for the consumers behind, and the check that pass A told the truth
                hd = self.header_decoder.process_bulk(llr, llr_tags)
                # messages of the packets PayloadMetadataInsert opened (it ignores syncwords inside a packet)
                opened_at = pm["tags"]["syncword"]["index"][pm["tags"]["kind"] == _abi.PKT_SYNCWORD]
                opened = res["headers"][np.isin(res["tags"]["index"], opened_at)]
                self._used_msgs = np.concatenate([self._used_msgs, opened])
                k = hd["messages"].size
                given, self._used_msgs = self._used_msgs[:k], self._used_msgs[k:]
                got = hd["messages"]
                same = (given["invalid_header"] == got["invalid_header"]) & \
                    ((given["packet_length"] == got["packet_length"]) | (got["invalid_header"] == 1))
                pending = given["invalid_header"] == 2  # pass A's message not there yet: checked when it arrives
                if pending.any():
                    self._early_hdrs = getattr(self, "_early_hdrs", []) + [got[i] for i in np.nonzero(pending)[0]]
                res.update(header_messages=got, header_bytes=hd["header_bytes"], packet_type=hd["packet_type"],
                           payload_llr=hd["payload_llr"], payload_tags=hd["payload_tags"],
                           header_mismatches=int(np.sum(~same & ~pending)) + early_mismatch)
                # payload tail, packet_receiver.hpp:140-147: BinarySlicer<true> -> PackBits -> CrcCheck
                # (which needs whole packets: an unfinished one waits for the next batch)
                soft = hd["payload_llr"] if self._payload_carry is None else torch.cat([self._payload_carry,
                                                                                       hd["payload_llr"]])
                lens = np.concatenate([self._payload_lens, hd["payload_tags"]["payload_bits"].astype(np.uint64)])
                ends = np.cumsum(lens)
                whole = int(np.searchsorted(ends, soft.numel(), side="right"))
                used = int(ends[whole - 1]) if whole else 0
                packed = slice_pack(soft[:used])
                data, out_len = self.payload_crc_check.process_bulk(packed, lens[:whole] // 8)
                self._payload_carry = soft[used:].clone() if used < soft.numel() else None
                self._payload_lens = lens[whole:]
                res.update(packets=data, packet_lengths=out_len, crc_ok=out_len > 0)
        return res

    def _stage12(self, fut1):
        return self._stage2(fut1.result())

    def announce(self, x):
        """the detector's look-ahead (SyncwordDetection.announce)"""
        self.syncword_detection.announce(x)

    def process_bulk(self, x, header_fn=None, tags_cap=4096, history=None, next_x=None):
        """x: complex64 CUDA tensor.  Returns dict(consumed, symbols, tags, detector_tags):
        symbols = CostasLoop output (one per symbol), tags = symbol-rate tags.  With
        pipelined=True the dict belongs to an earlier batch (None while the pipeline fills).
        history: when x is a window of a device ring buffer, the view of the >= 2T+1 items that
        precede x in the ring; the chain then reads the delayed stream in place (no copy).
        next_x: the window the next call will present (SyncwordDetection look-ahead)."""
        front = self._stage0(x, tags_cap, history, next_x)
        if not self.pipelined:
            return self._stage2(self._stage1(*front, header_fn))
        f1 = self._workers[0].submit(self._stage1, *front, header_fn)
        f2 = self._workers[1].submit(self._stage12, f1)
        self._inflight.append(f2)
        if len(self._inflight) > 2:  # keep at most two batches behind the detector
            return self._inflight.pop(0).result()
        return None

    def flush(self):
        """pipelined mode: results of the batches still in flight (oldest first)"""
        out = [f.result() for f in self._inflight]
        self._inflight = []
        return out


def mapper(x, map_values):
    """Mapper<uint8_t, c64 | float>, mapper.hpp:13-51: out[i] = map[x[i] & (len(map) - 1)]"""
    torch = _torch()
    x = x.contiguous()
    assert x.is_cuda and x.dtype == torch.uint8
    m = np.ascontiguousarray(map_values)
    kind = 0 if np.iscomplexobj(m) else 1
    m = m.astype(np.complex64 if kind == 0 else np.float32)
    out = torch.empty(x.numel(), dtype=torch.complex64 if kind == 0 else torch.float32, device=x.device)
    check(lib().gr4pm_mapper_process(x.data_ptr(), x.numel(), out.data_ptr(), _np_ptr(m), m.size, kind,
                                     _stream_handle()), "Mapper")
    return out


def burst_shaper(x, leading_shape, trailing_shape, packet_len, packet_offset=None):
    """BurstShaper, burst_shaper.hpp:47-126 over whole packets ("packet_len" tags = packet_len[],
    back to back unless packet_offset is given)"""
    torch = _torch()
    x = x.contiguous()
    assert x.is_cuda and x.dtype in (torch.complex64, torch.float32)
    lead = np.ascontiguousarray(leading_shape, dtype=np.float32)
    trail = np.ascontiguousarray(trailing_shape, dtype=np.float32)
    pl = np.ascontiguousarray(packet_len, dtype=np.uint64)
    po = (np.concatenate([[0], np.cumsum(pl)[:-1]]).astype(np.uint64) if packet_offset is None
          else np.ascontiguousarray(packet_offset, dtype=np.uint64))
    out = torch.empty_like(x)
    check(lib().gr4pm_burst_shaper_process(x.data_ptr(), x.numel(), out.data_ptr(), 0 if x.dtype == torch.complex64 else 1,
                                           _np_ptr(lead), lead.size, _np_ptr(trail), trail.size, _np_ptr(po), _np_ptr(pl),
                                           pl.size, _stream_handle()), "BurstShaper")
    return out


class BurstGenerator:
    """The burst path of PacketTransmitterPdu (packet_transmitter_pdu.hpp:84-337, stream_mode = false)
    plus the channel of apps/packet_transceiver.cpp:48-78, on the device, as a test-signal source:

        payload bytes + CRC-32  ->  [header (formatter, (128,32) LDPC + repetition) | payload] bits
        -> AdditiveScrambler (CCSDS 131.0-B-5, restarted per packet) -> PackBits(2) -> Mapper(QPSK)
        -> [syncword (BPSK) | header | payload | 9 ramp-down symbols | 11 zero symbols]
        -> InterpolatingFirFilter (4 samples/symbol, the transmitter's RRC) -> BurstShaper (sine ramps)
        -> gaps, Rotator (carrier offset), AWGN.

    Header formatting and its FEC encoding (32 bits per packet) and the CRC run on the host."""

    RAMP_DOWN, FLUSH = 9, 11   # :213-216, :209

    def __init__(self, samples_per_symbol=4, generator=None):
        import os
        self.sps = samples_per_symbol
        here = os.path.dirname(os.path.abspath(__file__))
        self.generator = np.fromfile(os.path.join(here, "data", "header_ldpc_generator.u32"), dtype="<u4") \
            if generator is None else np.ascontiguousarray(generator, dtype=np.uint32)
        taps = packet_transmitter_rrc_taps(samples_per_symbol)  # computed here, not shipped as data
        self.rrc_taps = taps
        self.fir = InterpolatingFirFilter(samples_per_symbol, taps)
        self.scrambler = AdditiveScrambler(0x4001, 0x18E38, 16, dtype="uint8")   # :118-122
        a = np.float32(np.sqrt(np.float32(2.0)) / np.float32(2.0))
        self.qpsk = np.array([a + 1j * a, a - 1j * a, -a + 1j * a, -a - 1j * a], dtype=np.complex64)  # :131-134
        ramp, offset = 4 * samples_per_symbol, 4 * samples_per_symbol                              # :296-300
        n_lead = offset + ramp
        self.leading = np.sin((np.arange(n_lead) + 1.0) / n_lead * 0.5 * np.pi).astype(np.float32)  # :301-306
        n_trail = self.FLUSH * samples_per_symbol - offset + ramp
        self.trailing = np.sin((np.arange(n_trail)[::-1] + 1.0) / n_trail * 0.5 * np.pi).astype(np.float32)  # :307-313

    def header_bits(self, packet_length, packet_type=0):
        """header_formatter.hpp:104-107 + header_fec_encoder.hpp:60-107 -> 256 bits"""
        info = ((packet_length >> 8) & 0xFF) << 24 | (packet_length & 0xFF) << 16 | (packet_type & 0xFF) << 8 | 0x55
        bits = [(info >> (31 - i)) & 1 for i in range(32)]
        bits += [bin(info & int(g)).count("1") & 1 for g in self.generator]
        return np.array(bits + bits, dtype=np.uint8)

    def symbols(self, payloads, packet_types=None):
        """packet symbols (device complex64) back to back, and the length of each packet in symbols"""
        import zlib
        torch = _torch()
        bits, resets, n_syms, pos = [], [], [], 0
        for k, data in enumerate(payloads):
            data = bytes(data)
            crc = zlib.crc32(data)
            body = np.frombuffer(data + crc.to_bytes(4, "big"), dtype=np.uint8)   # crc_append, :60-69
            b = np.concatenate([self.header_bits(len(data), 0 if packet_types is None else packet_types[k]),
                                np.unpackbits(body)])
            resets.append(pos)
            bits.append(b)
            pos += b.size
            n_syms.append(64 + b.size // 2 + self.RAMP_DOWN + self.FLUSH)
        allbits = torch.from_numpy(np.concatenate(bits)).cuda()
        scr = self.scrambler.process_bulk(allbits, np.array(resets, dtype=np.uint64))
        sym = mapper(pack_bits(scr, 2, 1), self.qpsk)                                           # :135-147
        sw = torch.from_numpy(np.where(SYNCWORD == 1, -1.0, 1.0).astype(np.complex64)).cuda()   # :158-181
        rng = np.random.default_rng(12345)
        parts, at = [], 0
        for k, b in enumerate(bits):
            n = b.size // 2
            ramp = torch.from_numpy(self.qpsk[rng.integers(0, 4, self.RAMP_DOWN)]).cuda()        # :211-247 (any data)
            parts += [sw, sym[at:at + n], ramp, torch.zeros(self.FLUSH, dtype=torch.complex64, device="cuda")]
            at += n
        return torch.cat(parts), np.array(n_syms, dtype=np.uint64)

    def bursts(self, payloads, packet_types=None):
        """shaped bursts back to back (device complex64) and their lengths in samples"""
        sym, n_syms = self.symbols(payloads, packet_types)
        self.fir = InterpolatingFirFilter(self.sps, self.rrc_taps)  # every call starts from a silent filter
        x = self.fir.process_bulk(sym)                                                           # :286-288
        lens = n_syms * np.uint64(self.sps)
        return burst_shaper(x, self.leading, self.trailing, lens), lens                         # :314-316

    def stream(self, payloads, gaps, freq_error=0.0, esn0_db=None, seed=1, tail=4000, packet_types=None,
               sfo_ppm=None, carrier="rotator"):
        """bursts separated by `gaps` (samples of silence before each burst) through the channel model of
        apps/packet_transceiver.cpp:48-78: PfbArbResampler<c64, c64, float> at rate 1 + 1e-6 sfo_ppm (sampling
        frequency offset; sfo_ppm=None leaves the block out), Rotator at freq_error rad/sample, AWGN for the
        given Es/N0 (tx power 0.32).  carrier="rotator" is the reference's Rotator (phasor recurrence, one serial
        lane for an untagged stream: 50 Msamples/s); carrier="closed_form" multiplies by exp(j freq_error n) from
        a double-precision phase instead -- a generator-only mode, not the reference's rounding, at HBM speed."""
        torch = _torch()
        x, lens = self.bursts(payloads, packet_types)
        total = int(np.sum(lens)) + int(np.sum(gaps)) + tail
        out = torch.zeros(total, dtype=torch.complex64, device="cuda")
        src = dst = 0
        for n, g in zip(lens, gaps):
            dst += int(g)
            out[dst:dst + int(n)] = x[src:src + int(n)]
            src += int(n)
            dst += int(n)
        if sfo_ppm is not None:                                                                  # :69-71
            rate = float(np.float32(1.0) + np.float32(1e-6) * np.float32(sfo_ppm))
            out, _ = PfbArbResampler(rate, rate_dtype="float32").process_bulk(out)
            total = out.numel()
        if freq_error and carrier == "closed_form":
            k = torch.arange(total, device="cuda", dtype=torch.float64)
            ph = torch.remainder(k * float(np.float32(freq_error)), 2.0 * np.pi)
            out = (out * torch.polar(torch.ones_like(ph), ph).to(torch.complex64)).contiguous()
        elif freq_error:
            out = Rotator(np.float32(freq_error)).process_bulk(out)                              # :72-73
        if esn0_db is not None:
            n0 = 0.32 * self.sps * 10.0 ** (-0.1 * esn0_db)                                      # :48-52
            g = torch.Generator(device="cuda")
            g.manual_seed(seed)
            # NoiseSource "amplitude" = sqrt(n0) is the standard deviation of the complex noise
            noise = torch.complex(torch.randn(total, generator=g, device="cuda"),
                                  torch.randn(total, generator=g, device="cuda")) * np.float32(np.sqrt(n0 / 2.0))
            out = (out + noise).to(torch.complex64)
        return out


def _hip_memcpy_d2d(dst, src, nbytes):
    """device-to-device copy through the HIP runtime the library is linked against"""
    import ctypes.util
    global _hiprt
    try:
        _hiprt
    except NameError:
        _hiprt = C.CDLL("libamdhip64.so")
        _hiprt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return 0 if _hiprt.hipMemcpy(dst, src, nbytes, 3) == 0 else -3


class MultiChannelPacketReceiver:
    """BASELINE config 3: `n_channels` independent receive chains on one GPU
    (packet_receiver.hpp:191-265 couples nothing across receivers).  The detector is ONE batched
    SyncwordDetection handle (every launch covers all channels, blockIdx.y = channel); behind
    it every channel has its own SyncwordDetectionFilter / CoarseFrequencyCorrection /
    SymbolFilter / SyncwordWipeoff / CostasLoop with their carried state, and the channels'
    chains are spread over `workers` host threads, each with a HIP stream of its own, whose
    process() calls queue up without waiting in between (gr4pm_set_deferred_sync).
    process_bulk(x[n_channels, n]) returns one PacketReceiver-style result per channel."""

    def __init__(self, n_channels, samples_per_symbol=4, syncword_freq_bins=4, syncword_threshold=9.5,
                 costas_constellation="QPSK", max_items=1 << 22, workers=8, soft_bits=False):
        import concurrent.futures
        torch = _torch()
        self.n_channels = n_channels
        workers = max(1, min(workers, n_channels))
        self._streams = [torch.cuda.Stream() for _ in range(workers)]
        self.chains = []
        for c in range(n_channels):
            with torch.cuda.stream(self._streams[c % workers]):
                self.chains.append(PacketReceiver(samples_per_symbol, syncword_freq_bins, syncword_threshold,
                                                  costas_constellation, soft_bits=soft_bits, detector=False))
        bpsk = np.array([1, -1], dtype=np.complex64)
        self.syncword_detection = SyncwordDetection(
            self.chains[0].rrc_taps, SYNCWORD, bpsk, -syncword_freq_bins, syncword_freq_bins,
            samples_per_symbol=samples_per_symbol, power_threshold=syncword_threshold, n_channels=n_channels,
            max_items=max_items)
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers, initializer=torch.cuda.set_device,
                                                           initargs=(torch.cuda.current_device(),))

    def announce(self, x):
        self.syncword_detection.announce(x)

    def _run_worker(self, w, y, det_tags, n, base, header_fn):
        torch = _torch()
        lib().gr4pm_set_deferred_sync(1)  # per thread: the calls below only queue their kernels
        out = {}
        try:
            with torch.cuda.stream(self._streams[w]):
                for c in range(w, self.n_channels, len(self._streams)):
                    chain = self.chains[c]
                    out[c] = chain._stage2(chain._stage1(0, y[c], det_tags[c], n, base, header_fn))
            self._streams[w].synchronize()
        finally:
            lib().gr4pm_set_deferred_sync(0)
        return out

    def process_bulk(self, x, header_fn=None, tags_cap=4096):
        """x: [n_channels, n] complex64 on the GPU.  Returns a list (one per channel) of
        dict(consumed, symbols, tags, detector_tags, ...) as PacketReceiver.process_bulk does."""
        st, y, det_tags, n = self.syncword_detection.process_bulk(x, want_output=True, tags_cap=tags_cap)
        if st != 0:
            return [{"status": st, "consumed": 0, "symbols": None, "tags": t, "detector_tags": t} for t in det_tags]
        base = self.syncword_detection._items_consumed - n
        _torch().cuda.current_stream().synchronize()  # y is read on the workers' streams
        futs = [self._pool.submit(self._run_worker, w, y, det_tags, n, base, header_fn)
                for w in range(len(self._streams))]
        res = {}
        for f in futs:
            res.update(f.result())
        return [res[c] for c in range(self.n_channels)]


class NativeMultiChannelReceiver:
    """gr4pm_multichannel_receiver: what MultiChannelPacketReceiver composes in Python, inside the
    library (one batched detector, every channel's own chain on worker threads with a stream each).
    process_bulk(x[n_channels, n], packet_length) -> one dict per channel."""

    def __init__(self, n_channels, samples_per_symbol=4, syncword_freq_bins=4, syncword_threshold=9.5,
                 costas_constellation="QPSK", max_items=1 << 22, tags_cap=4096, workers=12, output_ring=False):
        self.n_channels = n_channels
        # output_ring (streaming callers, bench.py): submit() takes its symbol buffer from a ring of 6 (more than the
        # batches in flight) instead of allocating one per batch -- a device allocation in the middle of a stream stalls
        # every stage; a result's "symbols" then stay valid until five further batches have been submitted
        self.output_ring = output_ring
        self._ring, self._ring_next = [], 0
        self.samples_per_symbol = samples_per_symbol
        self.tags_cap = max(int(tags_cap), 64)
        p = _abi.MultiChannelReceiverParams(n_channels, samples_per_symbol, syncword_freq_bins, syncword_threshold,
                                            {"PILOT": 0, "BPSK": 1, "QPSK": 2}[costas_constellation], max_items,
                                            self.tags_cap, workers)
        self._h = C.c_void_p()
        self._pending = []
        check(lib().gr4pm_multichannel_receiver_create(C.byref(p), C.byref(self._h)), "MultiChannelReceiver")

    def announce(self, x):
        x = _dev_c64_rows(x)
        assert x.dim() == 2 and x.shape[0] == self.n_channels
        _inputs_ready(x)
        check(lib().gr4pm_multichannel_receiver_announce(self._h, x.data_ptr(), x.stride(0), x.shape[1]),
              "MultiChannelReceiver.announce")

    def submit(self, x, packet_length=None):
        """pipelined form: returns the items consumed per channel as soon as the detector has taken the batch;
        the stages behind it keep working on up to four batches.  Results come from collect(), in order."""
        torch = _torch()
        x = _dev_c64_rows(x)
        assert x.dim() == 2 and x.shape[0] == self.n_channels
        _inputs_ready(x)
        Cn, n = self.n_channels, x.shape[1]
        stride = n // self.samples_per_symbol + self.tags_cap + 64
        if self.output_ring:
            if not self._ring or self._ring[0].shape[1] < stride:
                self._ring = [torch.empty((Cn, stride), dtype=torch.complex64, device=x.device) for _ in range(6)]
            sym = self._ring[self._ring_next][:, :stride]
            stride = self._ring[self._ring_next].stride(0)
            self._ring_next = (self._ring_next + 1) % 6
        else:
            sym = torch.empty((Cn, stride), dtype=torch.complex64, device=x.device)
        consumed = C.c_size_t(0)
        check(lib().gr4pm_multichannel_receiver_submit(
            self._h, x.data_ptr(), x.stride(0), n, 0 if packet_length is None else int(packet_length),
            sym.data_ptr(), stride, C.byref(consumed)), "MultiChannelReceiver.submit")
        self._pending.append((sym, x))  # keep the buffers alive until collected
        return consumed.value

    def in_flight(self):
        return int(lib().gr4pm_multichannel_receiver_in_flight(self._h))

    def set_input_in_place(self, on=True):
        """the caller keeps every submitted input unchanged until it has been collected (a device ring): the receiver
        reads the detector's delayed stream in place instead of writing a delayed copy per batch.  Same results."""
        check(lib().gr4pm_multichannel_receiver_set_input_in_place(self._h, 1 if on else 0),
              "MultiChannelReceiver.set_input_in_place")

    def collect(self):
        Cn = self.n_channels
        sym, _x = self._pending.pop(0)
        consumed = C.c_size_t(0)
        n_sym, n_tags, n_det = (C.c_size_t * Cn)(), (C.c_size_t * Cn)(), (C.c_size_t * Cn)()
        tags = np.zeros((Cn, self.tags_cap), dtype=TAG_DTYPE)
        det = np.zeros((Cn, self.tags_cap), dtype=TAG_DTYPE)
        check(lib().gr4pm_multichannel_receiver_collect(self._h, C.byref(consumed), n_sym, _np_ptr(tags), n_tags,
                                                        _np_ptr(det), n_det), "MultiChannelReceiver.collect")
        return [{"status": 0, "consumed": consumed.value, "symbols": sym[c, : n_sym[c]],
                 "tags": tags[c, : n_tags[c]].copy(), "detector_tags": det[c, : n_det[c]].copy()} for c in range(Cn)]

    def process_bulk(self, x, packet_length=None):
        torch = _torch()
        x = _dev_c64_rows(x)
        assert x.dim() == 2 and x.shape[0] == self.n_channels
        _inputs_ready(x)
        Cn, n = self.n_channels, x.shape[1]
        stride = n // self.samples_per_symbol + self.tags_cap + 64
        sym = torch.empty((Cn, stride), dtype=torch.complex64, device=x.device)
        consumed = C.c_size_t(0)
        n_sym, n_tags, n_det = (C.c_size_t * Cn)(), (C.c_size_t * Cn)(), (C.c_size_t * Cn)()
        tags = np.zeros((Cn, self.tags_cap), dtype=TAG_DTYPE)
        det = np.zeros((Cn, self.tags_cap), dtype=TAG_DTYPE)
        check(lib().gr4pm_multichannel_receiver_process(
            self._h, x.data_ptr(), x.stride(0), n, 0 if packet_length is None else int(packet_length),
            sym.data_ptr(), stride, C.byref(consumed), n_sym, _np_ptr(tags), n_tags, _np_ptr(det), n_det),
            "MultiChannelReceiver.process")
        return [{"status": 0, "consumed": consumed.value, "symbols": sym[c, : n_sym[c]],
                 "tags": tags[c, : n_tags[c]].copy(), "detector_tags": det[c, : n_det[c]].copy()} for c in range(Cn)]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_multichannel_receiver_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class NativePacketReceiver:
    """gr4pm_packet_receiver: the same chain as PacketReceiver (front end, or soft_bits up to the
    LLR decoder) composed and pipelined in the C++ library -- stage threads, streams and
    intermediate buffers live there, so no Python runs between the kernels of a batch.
    Identical outputs (tests compare bit for bit).  parsed_header feedback: a constant
    packet_length per call (None: every header invalid)."""

    def __init__(self, samples_per_symbol=4, syncword_freq_bins=4, syncword_threshold=9.5,
                 costas_constellation="QPSK", max_items=1 << 22, tags_cap=4096, pipelined=False, soft_bits=False,
                 decode_headers=False, output_ring=False, packets_only=False, result_fields="all", packets_cap=None):
        """result_fields="packets" (decode_headers): collect() marshals only what a packet sink needs -- consumed, the accepted
        detections' tags, header messages, packet lengths and bytes -- and leaves the per-symbol tag lists (tens of thousands
        of 96-byte records per batch on a packet-dense stream) in the library; "all": everything the C result holds.
        packets_only (a form of decode_headers; include/gr4pm_hip.h): IQ in, CRC-checked packets out, the stream between
        the Costas loop and the packer never written to memory -- the same packets / header messages / tags; the result
        has no "llr", "payload_llr" and "pdu_symbols" arrays (None)"""
        soft_bits = soft_bits or decode_headers
        self.packets_only = bool(packets_only)
        assert result_fields in ("all", "packets")
        self.result_fields = result_fields
        self._packets_cap = packets_cap  # (tests: a packet output buffer that is too small for a batch)
        self.output_ring = output_ring
        self.samples_per_symbol, self.pipelined, self.soft_bits = samples_per_symbol, pipelined, soft_bits
        self.decode_headers = decode_headers
        self.time_threshold = 768
        self._alist = header_ldpc_alist().encode() if decode_headers else None
        p = _abi.PacketReceiverParams(samples_per_symbol, syncword_freq_bins, syncword_threshold,
                                      CONSTELLATIONS[costas_constellation.upper()], max_items, tags_cap,
                                      1 if pipelined else 0, 1 if soft_bits else 0, 1 if decode_headers else 0,
                                      self._alist, 1 if packets_only else 0)
        self._h = C.c_void_p()
        check(lib().gr4pm_packet_receiver_create(C.byref(p), C.byref(self._h)), "PacketReceiver")
        self._keep = []  # (input tensors, output tensors) of the batches in flight

    def announce(self, x):
        """names the input of a later submit (after the ones already announced): the detector's
        look-ahead, up to two batches ahead.  The caller keeps x alive and unchanged until then."""
        x = _dev_c64(x)
        _inputs_ready(x)
        check(lib().gr4pm_packet_receiver_announce(self._h, x.data_ptr(), x.numel()), "PacketReceiver.announce")

    # output buffers.  output_ring=True (streaming callers, bench.py): a ring of _OUT_RING sets (more
    # than the batches the library keeps in flight), allocated at the first submit and whenever a
    # bigger batch arrives -- not once per call: a device allocation in the middle of a stream stalls
    # every stage.  A result's "symbols" / "llr" / "packets" then stay valid until _OUT_RING - 1
    # further batches have been submitted.
    _OUT_RING = 8

    def _outputs(self, n, device):
        torch = _torch()
        if not self.output_ring:  # fresh buffers for every batch: results stay valid as long as they are referenced
            n_sym = n // self.samples_per_symbol + 4160
            return (torch.empty(n_sym, dtype=torch.complex64, device=device),
                    torch.empty(2 * n_sym if self.soft_bits and not self.packets_only else 1, dtype=torch.float32, device=device),
                    torch.empty((self._packets_cap or n // 16 + 65536) if self.decode_headers else 1, dtype=torch.uint8, device=device))
        if getattr(self, "_out_n", -1) < n:
            n_sym = n // self.samples_per_symbol + 4160
            self._out_ring = [(torch.empty(n_sym, dtype=torch.complex64, device=device),
                               torch.empty(2 * n_sym if self.soft_bits and not self.packets_only else 1, dtype=torch.float32,
                                           device=device),
                               torch.empty(n // 16 + 65536 if self.decode_headers else 1, dtype=torch.uint8,
                                           device=device)) for _ in range(self._OUT_RING)]
            self._out_n, self._out_next = n, 0
        out = self._out_ring[self._out_next]
        self._out_next = (self._out_next + 1) % self._OUT_RING
        return out

    def submit(self, x, packet_length=None, history=None, next_x=None):
        x = _dev_c64(x)
        _inputs_ready(x)
        n = x.numel()
        sym, llr, pk = self._outputs(n, x.device)
        delayed = None
        if history is not None:
            d = 2 * self.time_threshold + 1
            assert history.numel() >= d and history.data_ptr() + history.numel() * 8 == x.data_ptr(), \
                "history must be the ring contents that directly precede x"
            delayed = x.data_ptr() - 8 * d
        nx = None if next_x is None else _dev_c64(next_x)
        check(lib().gr4pm_packet_receiver_submit(
            self._h, x.data_ptr(), n, delayed, None if nx is None else nx.data_ptr(), 0 if nx is None else nx.numel(),
            0 if packet_length is None else int(packet_length), sym.data_ptr(), sym.numel(),
            llr.data_ptr() if self.soft_bits and not self.packets_only else None, llr.numel(),
            pk.data_ptr() if self.decode_headers else None, pk.numel()), "PacketReceiver.submit")
        self._keep.append((x, history, nx, sym, llr, pk))

    def collect(self):
        r = _abi.PacketReceiverResult()
        st = lib().gr4pm_packet_receiver_collect(self._h, C.byref(r))
        x, history, nx, sym, llr, pk = self._keep.pop(0)
        check(st, "PacketReceiver")
        if st > 0:  # a gr::work::Status the batch ended with (INSUFFICIENT_OUTPUT_ITEMS ...): the batch is lost, the receiver goes on
            return {"status": int(st), "error": lib().gr4pm_last_error().decode(), "consumed": r.consumed}

        def records(ptr, n, dtype):
            if not n:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=n).copy()

        if self.result_fields == "packets" and self.decode_headers:
            res = {"status": 0, "consumed": r.consumed, "symbols": sym[: r.n_symbols], "tags": records(r.tags, r.n_tags, TAG_DTYPE),
                   "n_detector_tags": int(r.n_detector_tags), "n_packet_tags": int(r.n_packet_tags),
                   "header_messages": records(r.header_messages, r.n_header_messages, _abi.HEADER_MSG_DTYPE),
                   "header_mismatches": r.header_mismatches, "packets": pk[: r.n_packet_bytes],
                   "packet_lengths": records(r.packet_lengths, r.n_packets, np.dtype(np.uint64))}
            res["crc_ok"] = res["packet_lengths"] > 0
            return res
        res = {"status": 0, "consumed": r.consumed, "symbols": sym[: r.n_symbols],
               "tags": records(r.tags, r.n_tags, TAG_DTYPE),
               "detector_tags": records(r.detector_tags, r.n_detector_tags, TAG_DTYPE),
               "accepted": records(r.accepted, r.n_detector_tags, np.dtype(np.uint8)).astype(bool)}
        if self.soft_bits:
            torch = _torch()
            pdu_sym = None
            if not self.packets_only:
                pdu_sym = torch.empty(r.n_pdu_symbols, dtype=torch.complex64, device=sym.device)
                if r.n_pdu_symbols:  # the library's buffer is recycled: take a copy
                    check(_hip_memcpy_d2d(pdu_sym.data_ptr(), r.pdu_symbols, 8 * r.n_pdu_symbols), "pdu_symbols")
            res.update(pdu_symbols=pdu_sym, symbol_pdus=records(r.symbol_pdus, r.n_symbol_pdus, _abi.SYMBOL_PDU_DTYPE))
            res.update(llr=None if self.packets_only else llr[: r.n_llr], n_llr=int(r.n_llr),
                       llr_tags=records(r.llr_tags, r.n_llr_tags, PACKET_TAG_DTYPE),
                       packet_tags=records(r.packet_tags, r.n_packet_tags, PACKET_TAG_DTYPE),
                       ignored_syncwords=r.ignored_syncwords)
        if self.decode_headers:
            torch = _torch()
            pay = None
            if not self.packets_only:
                pay = torch.empty(r.n_payload_llr, dtype=torch.float32, device=llr.device)
                if r.n_payload_llr:  # the library's buffer is recycled at the next collect(): take a copy
                    check(_hip_memcpy_d2d(pay.data_ptr(), r.payload_llr, 4 * r.n_payload_llr), "payload_llr")
            res.update(header_messages=records(r.header_messages, r.n_header_messages, _abi.HEADER_MSG_DTYPE),
                       packet_type=records(r.packet_type, r.n_header_messages, np.dtype(np.int32)),
                       header_mismatches=r.header_mismatches, payload_llr=pay, n_payload_llr=int(r.n_payload_llr),
                       payload_tags=records(r.payload_tags, r.n_payload_tags, PACKET_TAG_DTYPE),
                       packets=pk[: r.n_packet_bytes],
                       packet_lengths=records(r.packet_lengths, r.n_packets, np.dtype(np.uint64)))
            res["crc_ok"] = res["packet_lengths"] > 0
        return res

    def set_symbol_pdu_callback(self, fn):
        """the symbol PDU tap (packet_receiver.hpp:159-189): fn(kind, symbols) is called from collect() once per
        complete header (kind 0, 128 symbols) / payload (kind 1) PDU with a numpy copy of its symbols; None removes it"""
        if fn is None:
            self._pdu_cb = None
            check(lib().gr4pm_packet_receiver_set_symbol_pdu_callback(self._h, None, None), "set_symbol_pdu_callback")
            return

        def tramp(user, kind, ptr, n):
            buf = (C.c_char * (8 * n)).from_address(ptr)
            fn(int(kind), np.frombuffer(buf, dtype=np.complex64, count=n).copy())
        self._pdu_cb = _abi.SYMBOL_PDU_FN(tramp)  # keep the trampoline alive
        check(lib().gr4pm_packet_receiver_set_symbol_pdu_callback(self._h, self._pdu_cb, None), "set_symbol_pdu_callback")

    def publish_symbol_pdus(self, header_endpoint="tcp://*:5000", payload_endpoint="tcp://*:5001"):
        """`zmq_output` of packet_receiver.hpp:159-189: collect() publishes every complete header PDU (128 symbols) on
        header_endpoint and every payload PDU on payload_endpoint, one ZeroMQ message of raw complex64 each
        (zmq_pdu_pub_sink.hpp:31-41; the library speaks ZMTP 3.0 itself).  Returns the two bound TCP ports;
        (None, None) stops publishing."""
        self._pdu_cb = None
        ports = (C.c_int * 2)(0, 0)
        enc = lambda e: None if e is None else e.encode()
        check(lib().gr4pm_packet_receiver_publish_symbol_pdus(self._h, enc(header_endpoint), enc(payload_endpoint), ports),
              "publish_symbol_pdus")
        return int(ports[0]), int(ports[1])

    def process_bulk(self, x, header_fn=None, tags_cap=None, history=None, next_x=None):
        """same calling convention as PacketReceiver.process_bulk (header_fn: None or a constant
        packet_length); pipelined: returns the result of an earlier batch, None while filling"""
        self.submit(x, header_fn, history, next_x)
        depth = 5 if self.decode_headers else (4 if self.soft_bits else 3)  # stages behind the detector
        if not self.pipelined or lib().gr4pm_packet_receiver_inflight(self._h) > depth:
            return self.collect()
        return None

    def flush(self):
        out = []
        while lib().gr4pm_packet_receiver_inflight(self._h):
            out.append(self.collect())
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _release("gr4pm_packet_receiver_destroy", self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class ZmqPduPubSink:
    """zmq_pdu_pub_sink.hpp:11-44: a ZeroMQ PUB socket bound to `endpoint`, one message per PDU holding its raw items
    (gr4pm_zmq_pub_*: ZMTP 3.0 in the library, no libzmq; host only -- works without a GPU)"""

    def __init__(self, endpoint="tcp://*:5555"):
        h = C.c_void_p()
        check(lib().gr4pm_zmq_pub_create(endpoint.encode(), C.byref(h)), "ZmqPduPubSink")
        self._h = h

    @property
    def port(self):
        return int(lib().gr4pm_zmq_pub_port(self._h))

    @property
    def subscribers(self):
        return int(lib().gr4pm_zmq_pub_subscribers(self._h))

    @property
    def dropped(self):
        return int(lib().gr4pm_zmq_pub_dropped(self._h))

    def process_one(self, data):
        """data: the PDU's items (any contiguous numpy array or bytes)"""
        buf = np.ascontiguousarray(data) if not isinstance(data, (bytes, bytearray)) else np.frombuffer(bytes(data), dtype=np.uint8)
        check(lib().gr4pm_zmq_pub_send(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes), "ZmqPduPubSink.process_one")

    def close(self):
        if getattr(self, "_h", None):
            lib().gr4pm_zmq_pub_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass
