"""gr4-packet-modem RX hot path for MI355X (gfx950): hand-written HIP kernels behind the
C-ABI of include/gr4pm_hip.h, plus the host-side mirror of the reference's block interface.

The directory name carries a hyphen (it is the repo's prescribed name); import it through
`__graft_entry__.load_package()` which registers it as `gr4_packet_modem_amd`."""
import os

import numpy as np

from ._abi import (EXPORTS, LIB_PATH, PACKET_TAG_DTYPE, PKT_HEADER_START, PKT_PAYLOAD, PKT_SYNCWORD,  # noqa: F401
                   TAG_DTYPE, TAG_OTHER, TAG_SYNCWORD, Gr4pmError, lib)
from .blocks import (SYNCWORD, ZmqPduPubSink, AdditiveScrambler, BurstGenerator, CrcCheck, burst_shaper, mapper, binary_slicer, pack_bits, slice_pack, CoarseFrequencyCorrection, ConstellationLLRDecoder,  # noqa: F401
                     CostasLoop, HeaderDecoder, HeaderFecDecoder, HeaderPayloadSplit, InterpolatingFirFilter, MultiChannelPacketReceiver, NativeMultiChannelReceiver, NativePacketReceiver, PacketReceiver,
                     PayloadMetadataInsert, PfbArbResampler, Rotator, SymbolFilter, SyncwordDetection,
                     SyncwordDetectionFilter, SyncwordRemove, SyncwordWipeoff, cfc_symbol_filter, cfc_symbol_filter_plan, cfc_symbol_filter_run,
                     header_ldpc_alist,
                     header_parse, packet_transmitter_rrc_taps, root_raised_cosine, sincosf, costas_phase_wrap)

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def default_pfb_arb_taps():
    """the default prototype of PfbArbResampler (pfb_arb_taps.hpp:12): 1280 float32 taps,
    shipped as a data blob (data/pfb_arb_taps.f32, see tests/golden/make_golden.py)"""
    return np.fromfile(os.path.join(_DATA, "pfb_arb_taps.f32"), dtype="<f4")
