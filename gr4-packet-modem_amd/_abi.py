"""ctypes declarations of include/gr4pm_hip.h (the C-ABI of libgr4pm_hip.so).

The library is HIP only: there is no CPU fallback anywhere in this package.  Importing works
without a GPU (so the symbol table can be checked), creating any block without one raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GR4PM_LIB: A/B builds of the same ABI (tools/); the default is the in-tree library
LIB_PATH = os.environ.get("GR4PM_LIB") or os.path.join(_HERE, "libgr4pm_hip.so")

OK = 0
INSUFFICIENT_INPUT_ITEMS = 1
INSUFFICIENT_OUTPUT_ITEMS = 2
TAG_SYNCWORD = 1
TAG_OTHER = 2

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
        ("user", "<i4"),  # the caller's cookie (never read by the library, copied with re-emitted tags)
    ],
    align=True,
)
assert TAG_DTYPE.itemsize == 48


# gr4pm_packet_tag: the control tags of the symbol-rate chain behind SyncwordWipeoff
PKT_SYNCWORD, PKT_HEADER_START, PKT_PAYLOAD = 1, 2, 3
PACKET_TAG_DTYPE = np.dtype(
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
assert PACKET_TAG_DTYPE.itemsize == 96


class SyncwordDetectionParams(C.Structure):
    _fields_ = [
        ("fft_size", C.c_size_t),
        ("samples_per_symbol", C.c_size_t),
        ("rrc_taps", C.c_void_p),
        ("n_rrc_taps", C.c_size_t),
        ("syncword", C.c_void_p),
        ("n_syncword", C.c_size_t),
        ("constellation", C.c_void_p),
        ("n_constellation", C.c_size_t),
        ("min_freq_bin", C.c_int),
        ("max_freq_bin", C.c_int),
        ("time_threshold", C.c_uint64),
        ("power_threshold", C.c_float),
        ("n_channels", C.c_size_t),
        ("max_items", C.c_size_t),
        ("stream", C.c_void_p),
    ]


class SdfParams(C.Structure):
    _fields_ = [("samples_per_symbol", C.c_size_t), ("syncword_size", C.c_size_t),
                ("header_size", C.c_size_t), ("stream", C.c_void_p)]


class HeaderMsg(C.Structure):
    _fields_ = [("packet_length", C.c_uint64), ("invalid_header", C.c_int32)]


HEADER_MSG_DTYPE = np.dtype([("packet_length", "<u8"), ("invalid_header", "<i4")], align=True)
assert HEADER_MSG_DTYPE.itemsize == C.sizeof(HeaderMsg)


class RotatorParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("phase_incr", C.c_float), ("delay", C.c_size_t),
                ("n_channels", C.c_size_t), ("stream", C.c_void_p)]


class CostasParams(C.Structure):
    _fields_ = [("loop_bandwidth", C.c_double), ("constellation", C.c_int), ("n_channels", C.c_size_t),
                ("stream", C.c_void_p)]


class WipeoffParams(C.Structure):
    _fields_ = [("syncword", C.c_void_p), ("n_syncword", C.c_size_t), ("stream", C.c_void_p)]


class InterpFirParams(C.Structure):
    _fields_ = [("interpolation", C.c_size_t), ("taps", C.c_void_p), ("n_taps", C.c_size_t),
                ("item_kind", C.c_int), ("stream", C.c_void_p)]


class SymbolFilterParams(C.Structure):
    _fields_ = [("samples_per_symbol", C.c_size_t), ("taps", C.c_void_p), ("n_taps", C.c_size_t),
                ("num_arms", C.c_size_t), ("delay", C.c_size_t), ("item_kind", C.c_int),
                ("stream", C.c_void_p)]


class PmiParams(C.Structure):
    _fields_ = [("syncword_size", C.c_size_t), ("header_size", C.c_size_t),
                ("syncword_costas_loop_bandwidth", C.c_double), ("header_costas_loop_bandwidth", C.c_double),
                ("payload_costas_loop_bandwidth", C.c_double), ("stream", C.c_void_p)]


class SyncwordRemoveParams(C.Structure):
    _fields_ = [("syncword_size", C.c_size_t), ("stream", C.c_void_p)]


class LlrParams(C.Structure):
    _fields_ = [("noise_sigma", C.c_float), ("constellation", C.c_int), ("stream", C.c_void_p)]


class ScramblerParams(C.Structure):
    _fields_ = [("mask", C.c_uint64), ("seed", C.c_uint64), ("length", C.c_uint64), ("count", C.c_uint64),
                ("item_kind", C.c_int), ("stream", C.c_void_p)]


class HeaderPayloadSplitParams(C.Structure):
    _fields_ = [("header_size", C.c_size_t), ("stream", C.c_void_p)]


class HeaderFecDecoderParams(C.Structure):
    _fields_ = [("alist", C.c_char_p), ("max_iterations", C.c_uint32), ("stream", C.c_void_p), ("arithmetic", C.c_int)]


class CrcCheckParams(C.Structure):
    _fields_ = [("num_bits", C.c_uint), ("poly", C.c_uint64), ("initial_value", C.c_uint64),
                ("final_xor", C.c_uint64), ("input_reflected", C.c_int), ("result_reflected", C.c_int),
                ("swap_endianness", C.c_int), ("discard_crc", C.c_int), ("skip_header_bytes", C.c_uint64),
                ("stream", C.c_void_p)]


class PacketReceiverParams(C.Structure):
    _fields_ = [("samples_per_symbol", C.c_size_t), ("syncword_freq_bins", C.c_int), ("syncword_threshold", C.c_float),
                ("costas_constellation", C.c_int), ("max_items", C.c_size_t), ("tags_cap", C.c_size_t),
                ("pipelined", C.c_int), ("soft_bits", C.c_int), ("decode_headers", C.c_int),
                ("header_alist", C.c_char_p), ("packets_only", C.c_int)]


class MultiChannelReceiverParams(C.Structure):
    _fields_ = [("n_channels", C.c_size_t), ("samples_per_symbol", C.c_size_t), ("syncword_freq_bins", C.c_int),
                ("syncword_threshold", C.c_float), ("costas_constellation", C.c_int), ("max_items", C.c_size_t),
                ("tags_cap", C.c_size_t), ("workers", C.c_int)]


class PacketReceiverResult(C.Structure):
    _fields_ = [("consumed", C.c_size_t), ("symbols", C.c_void_p), ("n_symbols", C.c_size_t), ("llr", C.c_void_p),
                ("n_llr", C.c_size_t), ("detector_tags", C.c_void_p), ("n_detector_tags", C.c_size_t),
                ("accepted", C.c_void_p), ("tags", C.c_void_p), ("n_tags", C.c_size_t),
                ("packet_tags", C.c_void_p), ("n_packet_tags", C.c_size_t), ("llr_tags", C.c_void_p),
                ("n_llr_tags", C.c_size_t), ("ignored_syncwords", C.c_size_t),
                ("header_messages", C.c_void_p), ("packet_type", C.c_void_p), ("n_header_messages", C.c_size_t),
                ("header_mismatches", C.c_size_t), ("payload_llr", C.c_void_p), ("n_payload_llr", C.c_size_t),
                ("payload_tags", C.c_void_p), ("n_payload_tags", C.c_size_t), ("packets", C.c_void_p),
                ("n_packet_bytes", C.c_size_t), ("packet_lengths", C.c_void_p), ("n_packets", C.c_size_t),
                ("pdu_symbols", C.c_void_p), ("n_pdu_symbols", C.c_size_t), ("symbol_pdus", C.c_void_p),
                ("n_symbol_pdus", C.c_size_t), ("symbol_pdu_resyncs", C.c_size_t)]


SYMBOL_PDU_DTYPE = np.dtype([("offset", "<u8"), ("length", "<u8"), ("kind", "<i4"), ("first", "<i4"), ("last", "<i4"),
                             ("pad", "<i4")])
SYMBOL_PDU_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)


class PfbArbParams(C.Structure):
    _fields_ = [("rate", C.c_double), ("rate_is_double", C.c_int), ("taps", C.c_void_p),
                ("n_taps", C.c_size_t), ("filter_size", C.c_size_t), ("stream", C.c_void_p)]


# every symbol include/gr4pm_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "gr4pm_last_error", "gr4pm_version", "gr4pm_device_count", "gr4pm_set_deferred_sync", "gr4pm_sincosf", "gr4pm_costas_phase_wrap",
    "gr4pm_packet_receiver_set_symbol_pdu_callback",
    "gr4pm_packet_receiver_publish_symbol_pdus",
    "gr4pm_zmq_pub_create",
    "gr4pm_zmq_pub_destroy",
    "gr4pm_zmq_pub_send",
    "gr4pm_zmq_pub_port",
    "gr4pm_zmq_pub_subscribers",
    "gr4pm_zmq_pub_dropped",
    "gr4pm_syncword_detection_create", "gr4pm_syncword_detection_destroy",
    "gr4pm_syncword_detection_reset", "gr4pm_syncword_detection_syncword_samples_size",
    "gr4pm_syncword_detection_self_corr", "gr4pm_syncword_detection_items_consumed", "gr4pm_syncword_detection_scan_counts",
    "gr4pm_syncword_detection_process", "gr4pm_syncword_detection_last_zpow",
    "gr4pm_syncword_detection_correlate_only",
    "gr4pm_syncword_detection_hint_next", "gr4pm_syncword_detection_announce",
    "gr4pm_syncword_detection_filter_create", "gr4pm_syncword_detection_filter_destroy",
    "gr4pm_syncword_detection_filter_reset", "gr4pm_syncword_detection_filter_process",
    "gr4pm_syncword_detection_filter_gate", "gr4pm_syncword_detection_filter_gate_resolve",
    "gr4pm_payload_metadata_insert_resolve",
    "gr4pm_rotator_create", "gr4pm_rotator_destroy", "gr4pm_rotator_reset", "gr4pm_rotator_process",
    "gr4pm_costas_loop_create", "gr4pm_costas_loop_destroy", "gr4pm_costas_loop_reset",
    "gr4pm_costas_loop_coeffs", "gr4pm_costas_loop_set", "gr4pm_costas_loop_process",
    "gr4pm_syncword_wipeoff_create", "gr4pm_syncword_wipeoff_destroy", "gr4pm_syncword_wipeoff_reset",
    "gr4pm_syncword_wipeoff_process",
    "gr4pm_interp_fir_create", "gr4pm_interp_fir_destroy", "gr4pm_interp_fir_reset",
    "gr4pm_interp_fir_process",
    "gr4pm_symbol_filter_create", "gr4pm_symbol_filter_destroy", "gr4pm_symbol_filter_reset",
    "gr4pm_symbol_filter_process", "gr4pm_cfc_symbol_filter_process",
    "gr4pm_cfc_symbol_filter_plan", "gr4pm_cfc_symbol_filter_run",
    "gr4pm_cfc_symbol_filter_plan_channels", "gr4pm_cfc_symbol_filter_run_channel",
    "gr4pm_cfc_symbol_filter_run_channels", "gr4pm_syncword_wipeoff_process_channels",
    "gr4pm_multichannel_receiver_set_input_in_place", "gr4pm_costas_loop_process_ragged", "gr4pm_costas_loop_set_small_footprint",
    "gr4pm_pfb_arb_resampler_create", "gr4pm_pfb_arb_resampler_destroy", "gr4pm_pfb_arb_resampler_reset",
    "gr4pm_pfb_arb_resampler_process",
    "gr4pm_firdes_root_raised_cosine",
    "gr4pm_payload_metadata_insert_create", "gr4pm_payload_metadata_insert_destroy",
    "gr4pm_payload_metadata_insert_reset", "gr4pm_payload_metadata_insert_process",
    "gr4pm_costas_loop_process_packets",
    "gr4pm_syncword_remove_create", "gr4pm_syncword_remove_destroy", "gr4pm_syncword_remove_reset",
    "gr4pm_syncword_remove_process",
    "gr4pm_constellation_llr_decoder_create", "gr4pm_constellation_llr_decoder_destroy",
    "gr4pm_constellation_llr_decoder_process",
    "gr4pm_additive_scrambler_create", "gr4pm_additive_scrambler_destroy", "gr4pm_additive_scrambler_reset",
    "gr4pm_additive_scrambler_process",
    "gr4pm_header_payload_split_create", "gr4pm_header_payload_split_destroy",
    "gr4pm_header_payload_split_reset", "gr4pm_header_payload_split_process", "gr4pm_header_payload_split_process_c64",
    "gr4pm_header_fec_decoder_create", "gr4pm_header_fec_decoder_destroy", "gr4pm_header_fec_decoder_process",
    "gr4pm_header_parse",
    "gr4pm_binary_slicer_process", "gr4pm_pack_bits_process", "gr4pm_slice_pack_process",
    "gr4pm_crc_check_create", "gr4pm_crc_check_destroy", "gr4pm_crc_check_compute", "gr4pm_crc_check_process",
    "gr4pm_mapper_process", "gr4pm_burst_shaper_process",
    "gr4pm_packet_receiver_create", "gr4pm_packet_receiver_destroy", "gr4pm_packet_receiver_submit", "gr4pm_packet_receiver_announce",
    "gr4pm_multichannel_receiver_create", "gr4pm_multichannel_receiver_destroy", "gr4pm_multichannel_receiver_announce",
    "gr4pm_multichannel_receiver_process", "gr4pm_multichannel_receiver_submit", "gr4pm_multichannel_receiver_collect",
    "gr4pm_multichannel_receiver_in_flight",
    "gr4pm_packet_receiver_collect", "gr4pm_packet_receiver_inflight",
]

_lib = None


class Gr4pmError(RuntimeError):
    """what the reference raises as gr::exception"""


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so); this library
    # links against the same soname.  Two copies in one process do not share the device: let
    # torch's be the one that is loaded (tensors come from there), also when the library is
    # loaded before anything has imported torch (build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, szp = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)
    L.gr4pm_last_error.restype = C.c_char_p
    L.gr4pm_version.restype = C.c_char_p
    L.gr4pm_device_count.restype = C.c_int
    if hasattr(L, "gr4pm_test_fail_allocations"):  # (the test build only: libgr4pm_hip_test.so)
        L.gr4pm_test_fail_allocations.argtypes = [C.c_long, C.c_long]
        L.gr4pm_test_fail_allocations.restype = None
        L.gr4pm_test_allocation_count.restype = C.c_ulonglong
    L.gr4pm_syncword_detection_create.argtypes = [C.POINTER(SyncwordDetectionParams), C.POINTER(vp)]
    L.gr4pm_syncword_detection_destroy.argtypes = [vp]
    L.gr4pm_syncword_detection_destroy.restype = None
    L.gr4pm_syncword_detection_reset.argtypes = [vp]
    L.gr4pm_syncword_detection_syncword_samples_size.argtypes = [vp]
    L.gr4pm_syncword_detection_syncword_samples_size.restype = sz
    L.gr4pm_syncword_detection_self_corr.argtypes = [vp]
    L.gr4pm_syncword_detection_self_corr.restype = C.c_float
    L.gr4pm_syncword_detection_items_consumed.argtypes = [vp]
    L.gr4pm_syncword_detection_items_consumed.restype = C.c_uint64
    L.gr4pm_syncword_detection_scan_counts.argtypes = [vp, sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.gr4pm_syncword_detection_scan_counts.restype = None
    L.gr4pm_syncword_detection_process.argtypes = [vp, vp, sz, sz, vp, sz, szp, vp, sz, vp]
    L.gr4pm_syncword_detection_last_zpow.argtypes = [vp, vp, sz]
    L.gr4pm_syncword_detection_correlate_only.argtypes = [vp, vp, sz, sz]
    L.gr4pm_syncword_detection_hint_next.argtypes = [vp, vp, sz, sz]
    L.gr4pm_syncword_detection_announce.argtypes = [vp, vp, sz, sz]
    L.gr4pm_syncword_detection_filter_create.argtypes = [C.POINTER(SdfParams), C.POINTER(vp)]
    L.gr4pm_syncword_detection_filter_destroy.argtypes = [vp]
    L.gr4pm_syncword_detection_filter_destroy.restype = None
    L.gr4pm_syncword_detection_filter_reset.argtypes = [vp]
    L.gr4pm_syncword_detection_filter_process.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, sz, sz, szp, szp, szp,
                                                          C.POINTER(C.c_int)]
    L.gr4pm_syncword_detection_filter_gate.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, szp]
    L.gr4pm_rotator_create.argtypes = [C.POINTER(RotatorParams), C.POINTER(vp)]
    L.gr4pm_rotator_destroy.argtypes = [vp]
    L.gr4pm_rotator_destroy.restype = None
    L.gr4pm_rotator_reset.argtypes = [vp]
    L.gr4pm_rotator_process.argtypes = [vp, vp, sz, sz, vp, vp, vp, sz]
    L.gr4pm_costas_loop_create.argtypes = [C.POINTER(CostasParams), C.POINTER(vp)]
    L.gr4pm_costas_loop_destroy.argtypes = [vp]
    L.gr4pm_costas_loop_destroy.restype = None
    L.gr4pm_costas_loop_reset.argtypes = [vp]
    L.gr4pm_costas_loop_coeffs.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.gr4pm_costas_loop_coeffs.restype = None
    L.gr4pm_costas_loop_set.argtypes = [vp, C.c_double, C.c_int]
    L.gr4pm_costas_loop_process.argtypes = [vp, vp, sz, sz, vp, vp, vp, sz]
    L.gr4pm_syncword_wipeoff_create.argtypes = [C.POINTER(WipeoffParams), C.POINTER(vp)]
    L.gr4pm_syncword_wipeoff_destroy.argtypes = [vp]
    L.gr4pm_syncword_wipeoff_destroy.restype = None
    L.gr4pm_syncword_wipeoff_reset.argtypes = [vp]
    L.gr4pm_syncword_wipeoff_process.argtypes = [vp, vp, sz, vp, vp, sz]
    L.gr4pm_syncword_wipeoff_process_channels.argtypes = [vp, sz, vp, sz, vp, vp, vp]
    L.gr4pm_cfc_symbol_filter_run_channels.argtypes = [vp, C.c_int, vp, sz, vp, sz, sz, vp, sz, vp, vp, vp, sz, vp, vp, vp, sz, sz]
    L.gr4pm_multichannel_receiver_set_input_in_place.argtypes = [vp, C.c_int]
    L.gr4pm_interp_fir_create.argtypes = [C.POINTER(InterpFirParams), C.POINTER(vp)]
    L.gr4pm_interp_fir_destroy.argtypes = [vp]
    L.gr4pm_interp_fir_destroy.restype = None
    L.gr4pm_interp_fir_reset.argtypes = [vp]
    L.gr4pm_interp_fir_process.argtypes = [vp, vp, sz, vp]
    L.gr4pm_symbol_filter_create.argtypes = [C.POINTER(SymbolFilterParams), C.POINTER(vp)]
    L.gr4pm_symbol_filter_destroy.argtypes = [vp]
    L.gr4pm_symbol_filter_destroy.restype = None
    L.gr4pm_symbol_filter_reset.argtypes = [vp]
    L.gr4pm_symbol_filter_process.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, szp, szp, szp]
    L.gr4pm_cfc_symbol_filter_process.argtypes = [vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, szp, szp, szp]
    L.gr4pm_cfc_symbol_filter_plan.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_int)]
    L.gr4pm_cfc_symbol_filter_run.argtypes = [vp, C.c_int, vp, vp, sz, vp, sz, vp, sz, vp, sz, szp, szp, szp]
    L.gr4pm_pfb_arb_resampler_create.argtypes = [C.POINTER(PfbArbParams), C.POINTER(vp)]
    L.gr4pm_pfb_arb_resampler_destroy.argtypes = [vp]
    L.gr4pm_pfb_arb_resampler_destroy.restype = None
    L.gr4pm_pfb_arb_resampler_reset.argtypes = [vp]
    L.gr4pm_pfb_arb_resampler_process.argtypes = [vp, vp, sz, vp, sz, szp, szp]
    L.gr4pm_payload_metadata_insert_create.argtypes = [C.POINTER(PmiParams), C.POINTER(vp)]
    L.gr4pm_payload_metadata_insert_destroy.argtypes = [vp]
    L.gr4pm_payload_metadata_insert_destroy.restype = None
    L.gr4pm_payload_metadata_insert_reset.argtypes = [vp]
    L.gr4pm_payload_metadata_insert_process.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, C.c_int, vp, sz,
                                                        szp, szp, szp, szp, szp]
    L.gr4pm_costas_loop_process_packets.argtypes = [vp, vp, sz, vp, vp, sz]
    L.gr4pm_syncword_detection_filter_gate_resolve.argtypes = [vp, vp]
    L.gr4pm_payload_metadata_insert_resolve.argtypes = [vp, vp]
    L.gr4pm_syncword_remove_create.argtypes = [C.POINTER(SyncwordRemoveParams), C.POINTER(vp)]
    L.gr4pm_syncword_remove_destroy.argtypes = [vp]
    L.gr4pm_syncword_remove_destroy.restype = None
    L.gr4pm_syncword_remove_reset.argtypes = [vp]
    L.gr4pm_syncword_remove_process.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, szp, szp]
    L.gr4pm_constellation_llr_decoder_create.argtypes = [C.POINTER(LlrParams), C.POINTER(vp)]
    L.gr4pm_constellation_llr_decoder_destroy.argtypes = [vp]
    L.gr4pm_constellation_llr_decoder_destroy.restype = None
    L.gr4pm_constellation_llr_decoder_process.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, szp, szp]
    L.gr4pm_additive_scrambler_create.argtypes = [C.POINTER(ScramblerParams), C.POINTER(vp)]
    L.gr4pm_additive_scrambler_destroy.argtypes = [vp]
    L.gr4pm_additive_scrambler_destroy.restype = None
    L.gr4pm_additive_scrambler_reset.argtypes = [vp]
    L.gr4pm_additive_scrambler_process.argtypes = [vp, vp, sz, vp, vp, sz]
    L.gr4pm_header_payload_split_create.argtypes = [C.POINTER(HeaderPayloadSplitParams), C.POINTER(vp)]
    L.gr4pm_header_payload_split_destroy.argtypes = [vp]
    L.gr4pm_header_payload_split_destroy.restype = None
    L.gr4pm_header_payload_split_reset.argtypes = [vp]
    L.gr4pm_header_payload_split_process.argtypes = [vp, vp, sz, vp, szp, vp, szp, vp, sz, vp, szp, vp, szp, sz]
    L.gr4pm_header_payload_split_process_c64.argtypes = [vp, vp, sz, vp, szp, vp, szp, vp, sz, vp, szp, vp, szp, sz]
    L.gr4pm_header_fec_decoder_create.argtypes = [C.POINTER(HeaderFecDecoderParams), C.POINTER(vp)]
    L.gr4pm_header_fec_decoder_destroy.argtypes = [vp]
    L.gr4pm_header_fec_decoder_destroy.restype = None
    L.gr4pm_header_fec_decoder_process.argtypes = [vp, vp, sz, vp, vp]
    L.gr4pm_header_parse.argtypes = [vp, vp, sz, vp, vp]
    L.gr4pm_header_parse.restype = None
    L.gr4pm_binary_slicer_process.argtypes = [vp, sz, vp, C.c_int, vp]
    L.gr4pm_pack_bits_process.argtypes = [vp, sz, vp, sz, C.c_uint, C.c_int, vp]
    L.gr4pm_slice_pack_process.argtypes = [vp, sz, vp, vp]
    L.gr4pm_crc_check_create.argtypes = [C.POINTER(CrcCheckParams), C.POINTER(vp)]
    L.gr4pm_crc_check_destroy.argtypes = [vp]
    L.gr4pm_crc_check_destroy.restype = None
    L.gr4pm_crc_check_compute.argtypes = [vp, vp, sz]
    L.gr4pm_crc_check_compute.restype = C.c_uint64
    L.gr4pm_crc_check_process.argtypes = [vp, vp, vp, vp, sz, vp, vp, szp]
    L.gr4pm_mapper_process.argtypes = [vp, sz, vp, vp, sz, C.c_int, vp]
    L.gr4pm_burst_shaper_process.argtypes = [vp, sz, vp, C.c_int, vp, sz, vp, sz, vp, vp, sz, vp]
    L.gr4pm_packet_receiver_create.argtypes = [C.POINTER(PacketReceiverParams), C.POINTER(vp)]
    L.gr4pm_packet_receiver_destroy.argtypes = [vp]
    L.gr4pm_packet_receiver_destroy.restype = None
    L.gr4pm_packet_receiver_submit.argtypes = [vp, vp, sz, vp, vp, sz, C.c_uint64, vp, sz, vp, sz, vp, sz]
    L.gr4pm_packet_receiver_announce.argtypes = [vp, vp, sz]
    L.gr4pm_multichannel_receiver_create.argtypes = [C.POINTER(MultiChannelReceiverParams), C.POINTER(vp)]
    L.gr4pm_multichannel_receiver_destroy.argtypes = [vp]
    L.gr4pm_multichannel_receiver_destroy.restype = None
    L.gr4pm_multichannel_receiver_announce.argtypes = [vp, vp, sz, sz]
    L.gr4pm_multichannel_receiver_process.argtypes = [vp, vp, sz, sz, C.c_uint64, vp, sz, szp, szp, vp, szp, vp, szp]
    L.gr4pm_multichannel_receiver_submit.argtypes = [vp, vp, sz, sz, C.c_uint64, vp, sz, szp]
    L.gr4pm_multichannel_receiver_collect.argtypes = [vp, szp, szp, vp, szp, vp, szp]
    L.gr4pm_multichannel_receiver_in_flight.argtypes = [vp]
    L.gr4pm_packet_receiver_collect.argtypes = [vp, C.POINTER(PacketReceiverResult)]
    L.gr4pm_packet_receiver_inflight.argtypes = [vp]
    L.gr4pm_packet_receiver_set_symbol_pdu_callback.argtypes = [vp, vp, vp]
    L.gr4pm_packet_receiver_publish_symbol_pdus.argtypes = [vp, C.c_char_p, C.c_char_p, vp]
    L.gr4pm_zmq_pub_create.argtypes = [C.c_char_p, vp]
    L.gr4pm_zmq_pub_destroy.argtypes = [vp]
    L.gr4pm_zmq_pub_destroy.restype = None
    L.gr4pm_zmq_pub_send.argtypes = [vp, vp, C.c_size_t]
    L.gr4pm_zmq_pub_port.argtypes = [vp]
    L.gr4pm_zmq_pub_port.restype = C.c_int
    L.gr4pm_zmq_pub_subscribers.argtypes = [vp]
    L.gr4pm_zmq_pub_subscribers.restype = C.c_size_t
    L.gr4pm_zmq_pub_dropped.argtypes = [vp]
    L.gr4pm_zmq_pub_dropped.restype = C.c_uint64
    L.gr4pm_packet_receiver_inflight.restype = sz
    L.gr4pm_sincosf.argtypes = [vp, sz, vp, vp]
    L.gr4pm_costas_phase_wrap.argtypes = [vp, sz, vp]
    L.gr4pm_firdes_root_raised_cosine.argtypes = [C.c_double] * 4 + [sz, vp]
    L.gr4pm_firdes_root_raised_cosine.restype = sz
    _lib = L
    return L


def check(status, what):
    """negative status == the reference's `throw gr::exception(...)`"""
    if status < 0:
        raise Gr4pmError(f"{what}: status {status}: {lib().gr4pm_last_error().decode()}")
    return status
