"""The GR4 block wrappers (gr4-packet-modem_amd/host/gr4pm_gr4_blocks.hpp) go through a compiler and
through processBulk(): tests/gr4_blocks_driver.cpp includes them by the REFERENCE's header names and
spellings (SyncwordDetectionFilter<>, SymbolFilter<c64, c64, float>, CostasLoop<> ...), wires the receiver
front end like packet_receiver.hpp:34-127 and runs it chunk by chunk against tests/gr4_stub/ (a test-only
stand-in for the GR4 block API; gnuradio4 itself is not in the image)."""
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
import _oracle as orc
import _signals as sig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "gr4_blocks_driver.bin")
REC = np.dtype([("index", "<u8"), ("amplitude", "<f4"), ("phase", "<f4"), ("freq", "<f8"), ("freq_bin", "<i4"),
                ("noise_power", "<f4"), ("esn0_db", "<f4"), ("time_est", "<f4"), ("has_syncword", "<i4")], align=True)


def test_wrapper_header_compiles_against_the_api_stub():
    """no GPU needed: the header, the drop-in headers under host/gnuradio-4.0/packet-modem/ and the driver compile"""
    ge.build_gr4_driver(force=True)
    assert os.path.exists(DRIVER)


def test_wrappers_compile_with_trace_prints(tmp_path):
    """-DTRACE (README.md:144-150 of the reference: cmake -D CMAKE_CXX_FLAGS=-DTRACE) turns on the entry / exit prints of
    every processBulk() in the wrappers; the build with it must compile (the prints name spans and counters)"""
    cmd = ge.gr4_compile_command(os.path.join(ROOT, "tests", "gr4_blocks_driver.cpp"), str(tmp_path / "trace.bin"))
    r = subprocess.run(cmd[:1] + ["-DTRACE"] + cmd[1:], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    hdr = open(os.path.join(ROOT, "gr4-packet-modem_amd", "host", "gr4pm_gr4_blocks.hpp")).read()
    assert hdr.count("GR4PM_TRACE_ENTRY(") - 2 == hdr.count("gr::work::Status processBulk(")  # (two in the macro definition)


def test_unsupported_instantiations_are_compile_errors(tmp_path):
    """the wrappers are built for the instantiations the reference uses, registers or tests (round 6: SymbolFilter<float,
    float, float> is one of them, test/qa_symbol_filter.cpp:41); anything else must not compile (no silent CPU path)"""
    src = tmp_path / "bad.cpp"
    src.write_text('#include <gnuradio-4.0/packet-modem/symbol_filter.hpp>\n'
                   'gr::packet_modem::SymbolFilter<double, double, double> f;\nint main() { return 0; }\n')
    r = subprocess.run(ge.gr4_compile_command(str(src), str(tmp_path / "bad")), capture_output=True, text=True)
    assert r.returncode != 0 and "gr4pm: SymbolFilter is built for" in r.stderr


def _packets(n_pkt, payload_len, seed):
    rng = np.random.default_rng(seed)
    sps = 4
    rrc, _ = orc.unit_norm_rrc(sps)
    a = np.float32(np.sqrt(0.5))
    syms, starts = [], []
    for _ in range(n_pkt):
        gap = np.zeros(int(rng.integers(300, 900)), dtype=np.complex64)
        nb = 128 + (payload_len + 4) * 4
        body = (np.where(rng.integers(0, 2, nb) == 0, a, -a) + 1j * np.where(rng.integers(0, 2, nb) == 0, a, -a)).astype(np.complex64)
        syms += [gap, sig.BPSK[sig.SYNCWORD], body]
        starts.append(sum(len(s) for s in syms[:-2]))
    syms.append(np.zeros(1500, dtype=np.complex64))
    x = orc.interpolating_fir(np.concatenate(syms), sps, rrc)
    x = (orc.rotator(x, np.float32(0.011)) + sig.awgn(x.size, 0.05, seed + 1)).astype(np.complex64)
    return x, rrc, starts


@pytest.mark.gpu
@pytest.mark.parametrize("host_output,max_chunk", [(1, 1 << 20), (0, 1 << 20), (0, 40000), (1, 9001)])
def test_receiver_front_end_through_processBulk(tmp_path, host_output, max_chunk):
    ge.build_gr4_driver()
    sps, payload_len = 4, 100
    x, rrc, starts = _packets(8, payload_len, seed=5)
    fin = tmp_path / "in.c64"
    x.tofile(fin)
    prefix = str(tmp_path / "out")
    r = subprocess.run([DRIVER, "chain", str(fin), prefix, str(host_output), str(max_chunk), str(payload_len)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, GR4PM_GR4_DEBUG="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    hits, misses = r.stderr.count("arena hit"), r.stderr.count("arena miss")
    print(r.stdout, "arena hits", hits, "misses", misses)
    if not host_output:
        # the chain stages at its entry; inner edges read the producers' device copies.  A consumer span that
        # straddles two producer calls (left-over items + new ones) falls back to the host copy for that call.
        assert hits > 2 * misses
    got = np.fromfile(prefix + ".symbols.c64", dtype=np.complex64)
    sd_tags = np.fromfile(prefix + ".sd_tags.bin", dtype=REC)
    sym_tags = np.fromfile(prefix + ".sym_tags.bin", dtype=REC)
    # detector through the wrapper against the oracle detector (syncword_detection.hpp:204-356)
    ref = orc.SyncwordDetection(rrc, sig.SYNCWORD, sig.BPSK, -4, 4, power_threshold=9.5)
    _, ref_out, ref_tags = ref.process(x)
    assert np.array_equal(sd_tags["index"], [1537 + sps * s for s in starts])
    assert np.array_equal(sd_tags["index"], ref_tags["index"]) and np.array_equal(sd_tags["freq_bin"], ref_tags["freq_bin"])
    assert np.allclose(sd_tags["amplitude"], ref_tags["amplitude"], rtol=2e-4)
    assert np.allclose(sd_tags["freq"], ref_tags["freq"], atol=2e-6)
    if host_output:
        sd = np.fromfile(prefix + ".sd.c64", dtype=np.complex64)
        assert np.array_equal(sd.view(np.uint64), ref_out[: sd.size].view(np.uint64))  # delayed pass-through, bit-exact
        assert ref_out.size - sd.size < 2048
    # the chain behind it: the oracle blocks fed with the wrapper's own detector tags
    otags = np.zeros(sd_tags.size, dtype=orc.TAG_DTYPE)
    for k in ("index", "amplitude", "phase", "freq", "freq_bin", "noise_power", "esn0_db", "time_est"):
        otags[k] = sd_tags[k]
    otags["flags"] = ref_tags["flags"]  # "has syncword_* keys"
    n_sd = int(np.fromfile(prefix + ".counts.bin", dtype=np.uint64)[0])
    z = orc.coarse_frequency_correction(ref_out[:n_sd], sd_tags["index"], sd_tags["freq"], delay=26)
    pfb = orc.rrc_taps(32.0 / float(orc.unit_norm_rrc(sps)[1]), 128.0, 1.0, 0.35, 32 * sps * 11)[:-1]
    sym, ref_sym_tags, _ = orc.symbol_filter(z, pfb, 32, sps, 44, tags=otags)
    bipolar = np.where(sig.SYNCWORD == 1, -1.0, 1.0).astype(np.float32)
    w = orc.syncword_wipeoff(sym, bipolar, ref_sym_tags["index"])
    c = orc.costas_loop(w, "QPSK", 0.01, ref_sym_tags["index"], ref_sym_tags["phase"])
    assert np.array_equal(sym_tags["index"], ref_sym_tags["index"])
    assert np.array_equal(sym_tags["phase"].view(np.uint32), ref_sym_tags["phase"].view(np.uint32))  # re-timed, adjusted
    assert got.size == c.size
    assert np.array_equal(got.view(np.uint64), c.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("host_output,max_chunk", [(1, 1 << 20), (0, 30000)])
def test_receiver_behind_the_front_end_through_processBulk(tmp_path, host_output, max_chunk):
    """The drop-in classes BEHIND the front end -- PayloadMetadataInsert<>, CostasLoop<> steered by its `constellation` /
    `loop_bandwidth` tags, SyncwordRemove<>, ConstellationLLRDecoder<>, AdditiveScrambler<float>, HeaderPayloadSplit<>,
    HeaderFecDecoder, and the symbol split of zmq_output HeaderPayloadSplit<std::complex<float>> -- created with the
    literal property maps of packet_receiver.hpp:123-139,159-162, wired like :208-240 and driven through processBulk()
    chunk by chunk (tags at chunk heads, parsed_header messages, two-output split, the Resampling<1, 64> decoder):
    their outputs are the ones of the Python composition over the same C ABI (blocks.PacketReceiver(soft_bits) and the
    header-loop blocks, which tests/test_gpu_parity.py holds against the oracle) bit for bit.  Until round 5 the
    processBulk() bodies of these six wrappers had never been instantiated by any compiler."""
    import torch
    ge.build_gr4_driver()
    pkg = ge.load_package()
    sps, payload_len = 4, 100
    x, rrc, starts = _packets(8, payload_len, seed=9)
    fin = tmp_path / "in.c64"
    x.tofile(fin)
    prefix = str(tmp_path / "out")
    env = dict(os.environ, GR4PM_DATA_DIR=os.path.join(ge.PKG_DIR, "data"))
    r = subprocess.run([DRIVER, "receiver", str(fin), prefix, str(host_output), str(max_chunk), str(payload_len)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    print(r.stdout)
    rx = pkg.PacketReceiver(max_items=x.size, soft_bits=True)
    res = rx.process_bulk(torch.from_numpy(x).cuda(), payload_len)
    f = lambda name, dt: np.fromfile(prefix + name, dtype=dt)
    costas, llr = f(".costas.c64", np.complex64), f(".llr.f32", np.float32)
    want_costas, want_llr = res["symbols"].cpu().numpy(), res["llr"].cpu().numpy()
    n_pkt = len(starts)
    assert costas.size == want_costas.size == n_pkt * (64 + 128 + (payload_len + 4) * 4)
    assert np.array_equal(costas.view(np.uint64), want_costas.view(np.uint64))      # PayloadMetadataInsert + tag-steered Costas
    assert llr.size == want_llr.size == 2 * n_pkt * (128 + (payload_len + 4) * 4)
    assert np.array_equal(llr.view(np.uint32), want_llr.view(np.uint32))            # SyncwordRemove + LLR decoder
    # the header loop over those LLRs: descrambler -> split -> LDPC decoder
    hd = pkg.HeaderDecoder().process_bulk(res["llr"], res["llr_tags"])
    hdr_llr, pay_llr, hdr_bytes = f(".hdr_llr.f32", np.float32), f(".pay_llr.f32", np.float32), f(".hdr_bytes.u8", np.uint8)
    assert hdr_llr.size == 256 * n_pkt and pay_llr.size == 8 * (payload_len + 4) * n_pkt
    assert np.array_equal(pay_llr.view(np.uint32), hd["payload_llr"].cpu().numpy().view(np.uint32))
    assert np.array_equal(f(".pay_tag_index.u64", np.uint64), hd["payload_tags"]["index"])
    assert np.array_equal(hdr_bytes, np.asarray(hd["header_bytes"]).reshape(-1))
    counts = f(".tail_counts.bin", np.uint64)
    assert counts[7] == int(np.sum(hd["invalid"]))  # (random header symbols: the decoder reports what it finds)
    # the symbol split of zmq_output: SyncwordRemove's symbols, 128 per header, payload_symbols per payload
    data, data_tags = pkg.SyncwordRemove(64).process_bulk(res["symbols"], res["packet_tags"])
    sym_tags = data_tags.copy()
    sym_tags["payload_bits"] = sym_tags["payload_symbols"]  # payload_length_key = "payload_symbols" (:161)
    hs, ps, _, _ = pkg.HeaderPayloadSplit(128).process_bulk(data, sym_tags)
    assert np.array_equal(f(".data.c64", np.complex64).view(np.uint64), data.cpu().numpy().view(np.uint64))
    assert np.array_equal(f(".hdr_sym.c64", np.complex64).view(np.uint64), hs.cpu().numpy().view(np.uint64))
    assert np.array_equal(f(".pay_sym.c64", np.complex64).view(np.uint64), ps.cpu().numpy().view(np.uint64))
    assert hs.numel() == 128 * n_pkt and ps.numel() == (payload_len + 4) * 4 * n_pkt


@pytest.mark.gpu
def test_device_arena_gives_identical_results(tmp_path):
    """host_output = 0 on the internal edges (samples stay on the device between wrapped blocks, one staging at
    the entry and one at the exit of the chain) == every block staging through the host, bit for bit"""
    ge.build_gr4_driver()
    x, _, _ = _packets(5, 60, seed=9)
    fin = tmp_path / "in.c64"
    x.tofile(fin)
    outs = []
    for ho, chunk in ((1, 1 << 20), (0, 1 << 20), (0, 30011)):
        prefix = str(tmp_path / f"o{ho}_{chunk}")
        r = subprocess.run([DRIVER, "chain", str(fin), prefix, str(ho), str(chunk), "60"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append((np.fromfile(prefix + ".symbols.c64", dtype=np.uint64), np.fromfile(prefix + ".sym_tags.bin", dtype=REC)))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])


@pytest.mark.gpu
def test_rotator_fir_resampler_wrappers(tmp_path):
    """Rotator<>, InterpolatingFirFilter<c64, c64, float>, PfbArbResampler<c64, c64, float, double> chained
    through processBulk() in odd chunk sizes against the oracle blocks"""
    ge.build_gr4_driver()
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(20000) + 1j * rng.standard_normal(20000)).astype(np.complex64)
    fin = tmp_path / "in.c64"
    x.tofile(fin)
    prefix = str(tmp_path / "b")
    taps_file = os.path.join(ge.PKG_DIR, "data", "pfb_arb_taps.f32")
    r = subprocess.run([DRIVER, "blocks", str(fin), prefix, taps_file], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rot = np.fromfile(prefix + ".rot.c64", dtype=np.complex64)
    fir = np.fromfile(prefix + ".fir.c64", dtype=np.complex64)
    arb = np.fromfile(prefix + ".arb.c64", dtype=np.complex64)
    ref_rot = orc.rotator(x, np.float32(0.1))
    assert np.array_equal(rot.view(np.uint64), ref_rot.view(np.uint64))
    ref_fir = orc.interpolating_fir(ref_rot, 4, orc.rrc_taps(1.0, 4.0, 1.0, 0.35, 44))
    assert np.array_equal(fir.view(np.uint64), ref_fir.view(np.uint64))
    ref_arb, _ = orc.pfb_arb_resampler(ref_fir, 1.1234, np.fromfile(taps_file, dtype=np.float32), 32, True)
    n = min(arb.size, ref_arb.size)
    assert abs(arb.size - ref_arb.size) <= 2 and np.array_equal(arb[:n].view(np.uint64), ref_arb[:n].view(np.uint64))


@pytest.mark.gpu
def test_float_symbol_filter_and_fir_wrappers_reference_qa(tmp_path):
    """test/qa_symbol_filter.cpp:17-63 on the wrappers (the instantiations the reference registers and tests,
    python/bindings/register_symbol_filter.cpp:9-15): +-1 float symbols -> InterpolatingFirFilter<float, float, float>
    (4 x, 44-tap RRC) -> SymbolFilter<float, float, float> (32 arms) through processBulk() in ragged chunks.  Bit for
    bit against the oracle's float forms, and the reference's own assertion: one symbol out per symbol in, amplitude
    0.24819523 +- 5e-3 behind the 11-symbol transient."""
    ge.build_gr4_driver()
    rng = np.random.default_rng(17)
    n = 100000  # (the reference runs 10^6; the oracle's serial loops set the size here)
    x = (1.0 - 2.0 * rng.integers(0, 2, n)).astype(np.float32)
    fin = tmp_path / "in.f32"
    x.tofile(fin)
    prefix = str(tmp_path / "f")
    r = subprocess.run([DRIVER, "floats", str(fin), prefix], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    fir = np.fromfile(prefix + ".fir.f32", dtype=np.float32)
    sym = np.fromfile(prefix + ".sym.f32", dtype=np.float32)
    ref_fir = orc.interpolating_fir(x, 4, orc.rrc_taps(1.0, 4.0, 1.0, 0.35, 44))
    assert np.array_equal(fir.view(np.uint32), ref_fir.view(np.uint32))
    ref_sym, _, consumed = orc.symbol_filter(ref_fir, orc.rrc_taps(32.0, 128.0, 1.0, 0.35, 32 * 4 * 11), 32, 4, 0)
    assert consumed == ref_fir.size and sym.size == n == ref_sym.size                     # qa_symbol_filter.cpp:56
    assert np.array_equal(sym.view(np.uint32), ref_sym.view(np.uint32))
    assert np.all(np.abs(np.abs(sym[11:]) - 0.24819523) < 5e-3)                           # qa_symbol_filter.cpp:57-62


@pytest.mark.gpu
def test_pdu_forms_of_scrambler_and_fir_wrappers(tmp_path):
    """AdditiveScrambler<Pdu<uint8_t>> (additive_scrambler.hpp:102-159: the LFSR restarts at the head of every PDU) and
    InterpolatingFirFilter<Pdu<c64>, Pdu<c64>, float> (interpolating_fir_filter.hpp:104-175: the history runs on across
    PDUs, tag indices times the interpolation) -- the forms the reference's transmitter uses
    (packet_transmitter_pdu.hpp:119,288) -- through processOne() on PDUs of ragged sizes, empty ones among them,
    against the oracle."""
    ge.build_gr4_driver()
    rng = np.random.default_rng(23)
    bits = rng.integers(0, 2, 60000).astype(np.uint8)
    sym = ((1 - 2 * rng.integers(0, 2, 30000)) + 1j * (1 - 2 * rng.integers(0, 2, 30000))).astype(np.complex64) * np.float32(0.70710677)
    fb, fs = tmp_path / "bits.u8", tmp_path / "sym.c64"
    bits.tofile(fb)
    sym.tofile(fs)
    prefix = str(tmp_path / "p")
    r = subprocess.run([DRIVER, "pdus", str(fb), str(fs), prefix], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    sizes = np.fromfile(prefix + ".scr_sizes.u64", dtype=np.uint64).astype(np.int64)
    fir_sizes = np.fromfile(prefix + ".fir_sizes.u64", dtype=np.uint64).astype(np.int64)
    scr = np.fromfile(prefix + ".scr.u8", dtype=np.uint8)
    fir = np.fromfile(prefix + ".fir.c64", dtype=np.complex64)
    assert np.sum(sizes) == bits.size and np.sum(fir_sizes) == sym.size and np.any(sizes == 0) and np.any(fir_sizes == 0)
    starts = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    want = orc.AdditiveScrambler(0x4001, 0x18E38, 16, 0).process(bits, reset_index=np.unique(starts[sizes > 0]))
    assert np.array_equal(scr, want)
    ref = orc.InterpolatingFir(4, orc.rrc_taps(1.0, 4.0, 1.0, 0.35, 44))  # one running filter over the concatenated PDUs
    want_fir = ref.process(sym)
    assert np.array_equal(fir.view(np.uint64), want_fir.view(np.uint64))
    fir_starts = np.concatenate(([0], np.cumsum(fir_sizes)[:-1]))
    want_tags = [4 * (s0 + k // 2) for s0, k in zip(fir_starts, fir_sizes) if k]
    assert np.array_equal(np.fromfile(prefix + ".fir_tag_index.u64", dtype=np.uint64), np.asarray(want_tags, dtype=np.uint64))


REFERENCE_ROOT = "/root/reference"
GXX, CLANGXX = "g++", "/opt/rocm/lib/llvm/bin/clang++"
# (source under /root/reference, compiler, arguments, blocks, edges, settings calls that need the device): the reference's
# own flowgraph sources.  The transmit side (packet_transmitter_pdu.hpp, through packet_transceiver.cpp and the
# transceiver benchmark) uses C++23 constexpr std::vector (packet_transmitter_rrc_taps.hpp:9), which g++ 11 rejects and
# ROCm's clang accepts against the same libstdc++ -- the compiler oracle/Makefile builds oracle/_ref with.
FLOWGRAPHS = [
    ("benchmarks/benchmark_syncword_detection.cpp", GXX, ["2", "7.5"], 4, 3, 0),
    ("benchmarks/benchmark_packet_receiver.cpp", GXX, [], 20, 22, 3),
    ("apps/packet_receiver_file.cpp", GXX, ["/dev/null", "3"], 25, 27, 3),
    # round 6: the other link targets north_star names, and the flowgraphs that use the Rotator / PfbArbResampler /
    # InterpolatingFirFilter drop-ins (apps/packet_transceiver.cpp:71-75, packet_transmitter_pdu.hpp:288,343)
    ("apps/packet_receiver_soapy.cpp", GXX, ["100e6"], 25, 27, 3),
    ("apps/packet_transceiver.cpp", CLANGXX, ["10", "0.01", "1", "0"], 52, 56, 6),
    ("apps/packet_transceiver.cpp", CLANGXX, ["10", "0.01", "1", "1"], 47, 51, 6),      # stream_mode
    ("benchmarks/benchmark_packet_transceiver.cpp", CLANGXX, ["0"], 41, 44, 4),
    ("benchmarks/benchmark_packet_transceiver.cpp", CLANGXX, ["1", "2", "9.5", "1"], 34, 37, 4),
]


_FLOWGRAPH_BINARIES = {}


def _flowgraph_binary(source, cxx):
    """one build per source (the transceivers take a minute each and are run in two modes)"""
    if source in _FLOWGRAPH_BINARIES:
        return _FLOWGRAPH_BINARIES[source]
    if not os.path.exists(os.path.join(ge.PKG_DIR, "libgr4pm_hip.so")):
        ge.build()
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="gr4pm_flowgraph_"), "flowgraph.bin")
    subprocess.check_call([cxx, "-std=c++23", "-O1", "-D__HIP_PLATFORM_AMD__",
                           "-I", os.path.join(ROOT, "tests", "gr4_stub"),       # gnuradio4 stand-in (test-only)
                           "-I", os.path.join(ge.PKG_DIR, "host"),              # the drop-in headers, first
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           "-I", REFERENCE_INCLUDE,                             # everything else: the reference's own
                           "-o", exe, os.path.join(REFERENCE_ROOT, source), "-L" + ge.PKG_DIR, "-lgr4pm_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + ge.PKG_DIR, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    _FLOWGRAPH_BINARIES[source] = exe
    return exe


@pytest.mark.parametrize("source,cxx,args,n_blocks,n_edges,deferred", FLOWGRAPHS)
def test_reference_flowgraph_sources_compile_and_link_against_the_drop_in_headers(tmp_path, source, cxx, args, n_blocks, n_edges,
                                                                                 deferred):
    """north_star: "... so the existing apps/packet_receiver_* and benchmark_syncword_detection flowgraphs link against
    them unchanged".  The reference's OWN translation units -- benchmarks/benchmark_syncword_detection.cpp,
    benchmarks/benchmark_packet_receiver.cpp, apps/packet_receiver_file.cpp, apps/packet_receiver_soapy.cpp (round 6:
    gr::blocks::soapy::SoapyBlock<c64, 1> and its four settings from the stand-in), apps/packet_transceiver.cpp and
    benchmarks/benchmark_packet_transceiver.cpp (round 6: the whole transmitter of packet_transmitter_pdu.hpp in front of
    the receiver, with the PfbArbResampler / Rotator / InterpolatingFirFilter<c64> / InterpolatingFirFilter<Pdu<c64>> /
    AdditiveScrambler<Pdu<uint8_t>> drop-ins), and through them packet_receiver.hpp:34-265 -- are compiled unchanged, by path, with gr4-packet-modem_amd/host IN FRONT OF the reference's blocks/include (exactly
    INTEGRATION.md's CMake switch) and linked against libgr4pm_hip.so.  Every fg.emplaceBlock<T>({...}) spelling,
    property-map key, fg.connect<"port">(a).to<"port">(b) port name and item type, and pointer member
    (packet_receiver.syncword_detection, .payload_crc_check) then resolves against the HIP classes or the build fails;
    the blocks that have no drop-in (sources, sinks, CRC, parser ...) come from the reference's headers.  gnuradio4 is
    absent: Graph.hpp / Scheduler.hpp are the test stand-in (tests/gr4_stub), whose connect() checks names and types at
    compile time and whose runAndWait() executes nothing.  Here (no GPU, GR4_STUB_LIFECYCLE=0) the binary builds its
    graph -- the three device-needing settingsChanged() calls of the receiver noted -- and reports it.  Build container
    only: nothing of the reference, source or binary, travels to the GPU box."""
    src = os.path.join(REFERENCE_ROOT, source)
    if not os.path.exists(src):
        pytest.skip("the reference tree is not on this machine")
    exe = _flowgraph_binary(source, cxx)
    syms = subprocess.run(["nm", "-C", exe], capture_output=True, text=True, check=True).stdout
    assert "gr::packet_modem::hip::SyncwordDetection" in syms           # the HIP class is the one instantiated
    assert " U gr4pm_syncword_detection_create" in syms                 # ... and it calls the C ABI of the library
    assert "gr::packet_modem::SyncwordDetection::" not in syms          # the reference's detector is not in the binary
    if "transceiver" in source:  # "nm shows hip::PfbArbResampler / hip::Rotator in the transceiver"
        for cls in ("InterpolatingFirFilter<std::complex<float>, std::complex<float>, float>",
                    "InterpolatingFirFilter<gr::packet_modem::Pdu<std::complex<float>", "AdditiveScrambler<gr::packet_modem::Pdu<unsigned char>"):
            assert f"gr::packet_modem::hip::{cls}" in syms, cls
        if "apps/" in source:
            assert "gr::packet_modem::hip::PfbArbResampler<std::complex<float>, std::complex<float>, float, float>" in syms
            assert "gr::packet_modem::hip::Rotator<float>" in syms
        for ref_cls in ("InterpolatingFirFilter<", "AdditiveScrambler<", "PfbArbResampler<", "Rotator<"):
            assert f"gr::packet_modem::{ref_cls}" not in syms, ref_cls  # none of the reference's own forms was instantiated
    if "packet_receiver" in source or "transceiver" in source:
        for cls in ("SyncwordDetectionFilter", "CoarseFrequencyCorrection", "SymbolFilter", "SyncwordWipeoff",
                    "PayloadMetadataInsert", "CostasLoop", "SyncwordRemove", "ConstellationLLRDecoder", "AdditiveScrambler",
                    "HeaderPayloadSplit", "HeaderFecDecoder"):
            assert f"gr::packet_modem::hip::{cls}" in syms, cls
        assert "ldpc_toolbox" not in syms                               # header_fec_decoder.hpp:276: replaced, not linked
        # packet_receiver.hpp:163-168 (zmq_output): the sink is the library's ZMTP endpoint, not cppzmq (round 6)
        assert "gr::packet_modem::hip::ZmqPduPubSink<std::complex<float>" in syms and " U gr4pm_zmq_pub_create" in syms
        assert "zmq::socket_t" not in syms
        assert "gr::packet_modem::CrcCheck<" in syms and "gr::packet_modem::HeaderParser<" in syms  # the reference's own
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, GR4_STUB_LIFECYCLE="0"))
    assert r.returncode == 1, r.stdout + r.stderr
    assert (f"{n_blocks} blocks, {n_edges} edges, lifecycle skipped, {deferred} settings calls deferred" in r.stdout + r.stderr), \
        r.stdout + r.stderr
    if "syncword_detection" not in source:  # without the switch: the first device-needing call says that there is no CPU path
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


def test_remaining_reference_receive_headers_compile_on_the_stub(tmp_path):
    """syncword_detection_filter / payload_metadata_insert / syncword_remove / constellation_llr_decoder /
    additive_scrambler / header_payload_split of the reference, with processBulk() / processOne() instantiated on the
    stand-in's span and message types: the stand-in declares the API surface these blocks use (nothing is run; their
    arithmetic is pinned by the restated qa_*.cpp in test_oracle_reference_qa.py)"""
    if not os.path.isdir(REFERENCE_INCLUDE):
        pytest.skip("the reference tree is not on this machine")
    subprocess.check_call(["g++", "-std=c++23", "-O1", "-I", os.path.join(ROOT, "tests", "gr4_stub"), "-I", REFERENCE_INCLUDE,
                           "-o", str(tmp_path / "compile_only"), os.path.join(ROOT, "tests", "ref_headers_compile_only.cpp")])
    assert subprocess.call([str(tmp_path / "compile_only")]) == 0


def test_reference_syncword_detection_header_on_the_stub_agrees_with_the_oracle(ref_check, tmp_path):
    """syncword_detection.hpp itself -- start() (templates), the overlap-save correlation, the sequential best-bin /
    median scan over its mutable history (:267-298), output_tag (:56-115) -- compiled against the stand-in with the
    ORACLE's FFT underneath (gr4_stub/gnuradio-4.0/algorithm/fourier/fftw.hpp -> orc_fft), against the oracle's
    restatement: the same items, the same tag positions and the same tag values bit for bit, at nine bins and at one,
    over two chunkings.  Earns no parity credit (the API and the FFT under the block are stand-ins); it catches slips in
    the restatement of the detector, whose float compare chain no reference test pins beyond tag positions."""
    locations = [100, 1000, 1250, 10000, 13721, 43124, 58000 - 64]
    x, rrc = sig.qa_syncword_stream(60000, locations, 0.007, seed=31)
    x = (x + sig.awgn(x.size, 0.3, 32)).astype(np.complex64)
    for case, edge in (("syncword_detection", 4), ("syncword_detection_1bin", 0)):
        sd = orc.SyncwordDetection(rrc, sig.SYNCWORD, sig.BPSK, -edge, edge)
        st, want, want_tags = sd.process(x)
        assert st == 0 and want_tags.size >= len(locations) - 1
        for chunk in (1 << 20, 9000):
            y, t, c, p = _run_ref(ref_check, tmp_path, case, x, chunk=chunk)
            # with a 9000-item chunk the block takes 4 strides + one FFT length per call, the oracle everything at once:
            # the tail the reference leaves unconsumed may be longer, what it did produce must agree
            assert c == p and p <= want.size and want.size - p < 9000 + 2048
            assert np.array_equal(_bits(y), _bits(want[:p]))
            k = int(np.sum(want_tags["index"] < p))
            assert t.size == k and np.array_equal(t["index"], want_tags["index"][:k])
            for f in ("amplitude", "phase", "freq", "freq_bin", "noise_power", "esn0_db", "time_est"):
                assert np.array_equal(t[f], want_tags[f][:k]), (case, chunk, f)


@pytest.mark.gpu
def test_device_arena_over_a_double_mapped_ring(tmp_path):
    """gnuradio4's CircularBuffer is mapped twice back to back: a producer span runs past the end of the first mapping
    and the consumer sees the items behind the wrap one ring size lower.  Two wrapped Rotators with host_output = false
    between them over such a ring (registered with Arena::add_mirrored_ring; its host memory holds NaNs throughout):
    the output equals the oracle's two rotators only if the arena compares addresses modulo the ring"""
    ge.build_gr4_driver()
    rng = np.random.default_rng(8)
    n = 40000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    xin = tmp_path / "x.c64"
    x.tofile(xin)
    r = subprocess.run([DRIVER, "mirror", str(xin), str(tmp_path / "m")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert int(r.stdout.split("ring,")[1].split()[0]) > 5, r.stdout      # the aliased case really occurred
    got = np.fromfile(str(tmp_path / "m.mirror.c64"), dtype=np.complex64)
    want = orc.rotator(orc.rotator(x, np.float32(0.1)), np.float32(-0.03))
    assert not np.isnan(got.view(np.float32)).any()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
def test_device_arena_with_the_two_ends_of_an_edge_in_two_threads(tmp_path):
    """The reference's multi-threaded schedulers (benchmarks/README.md:8-26) run producer and consumer of an edge
    concurrently.  Two wrapped Rotators in two threads over a double-mapped ring, host_output = false between them and
    NaNs in the ring's host memory: the producer runs up to a ring ahead while the consumer is still reading the spans
    before, so every consumer span must come from a device buffer that has not been reused (the arena's buffer pool
    and pins; round 3's single staging buffer was overwritten under the reader).  Ten runs, bit-identical each."""
    ge.build_gr4_driver()
    rng = np.random.default_rng(9)
    n = 300000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    xin = tmp_path / "x.c64"
    x.tofile(xin)
    want = orc.rotator(orc.rotator(x, np.float32(0.1)), np.float32(-0.03))
    ahead = 0
    for run in range(10):
        r = subprocess.run([DRIVER, "threads", str(xin), str(tmp_path / "t")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        ahead = max(ahead, int(r.stdout.split("up to")[1].split()[0]))
        hits, uploads = int(r.stdout.split("device hits")[1].split()[0]), int(r.stdout.split("uploads")[1].split()[0])
        # uploads: the producer's own input spans (plain host memory) + the consumer spans whose device copy had been
        # retired; most consumer spans must have come from the device
        assert hits > 200 and uploads < hits, r.stdout
        got = np.fromfile(str(tmp_path / "t.threads.c64"), dtype=np.complex64)
        assert not np.isnan(got.view(np.float32)).any(), run
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), run
    assert ahead > 2900  # the producer really was more than one of its spans ahead of the reader


# ------------------------------------------------------------------ the stand-in checked in the other direction
REFERENCE_INCLUDE = "/root/reference/blocks/include"


@pytest.fixture(scope="module")
def ref_check(tmp_path_factory):
    """tests/ref_headers_check.cpp: the reference's OWN block headers compiled against tests/gr4_stub (only where
    /root/reference exists: the build container).  That it compiles is the first half of the check: the stand-in
    declares the API surface the reference uses."""
    if not os.path.isdir(REFERENCE_INCLUDE):
        pytest.skip("the reference tree is not on this machine")
    exe = tmp_path_factory.mktemp("refcheck") / "ref_headers_check"
    subprocess.check_call(["g++", "-std=c++23", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "tests", "gr4_stub"),
                           "-I", REFERENCE_INCLUDE, "-I", os.path.join(ROOT, "tests"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "ref_headers_check.cpp"),
                           # the FFT under the reference's SyncwordDetection is the oracle's (gr4_stub/.../fftw.hpp)
                           "-L", os.path.join(ROOT, "oracle"), "-lgr4pm_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return str(exe)


def _run_ref(exe, tmp_path, case, x, tags=None, chunk=4000):
    xin = tmp_path / f"{case}.in.c64"
    np.ascontiguousarray(x, dtype=np.complex64).tofile(xin)
    tpath = "-"
    if tags is not None:
        tpath = str(tmp_path / f"{case}.tags.bin")
        np.ascontiguousarray(tags, dtype=orc.TAG_DTYPE).tofile(tpath)
    prefix = str(tmp_path / case)
    subprocess.check_call([exe, case, str(xin), tpath, prefix, str(chunk)], stdout=subprocess.DEVNULL)
    out = np.fromfile(prefix + ".out.c64", dtype=np.complex64)
    consumed, produced = np.fromfile(prefix + ".counts.bin", dtype=np.uint64)
    return out, np.fromfile(prefix + ".out_tags.bin", dtype=REC), int(consumed), int(produced)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _detection_tags(idx, seed):
    rng = np.random.default_rng(seed)
    tags = np.zeros(len(idx), dtype=orc.TAG_DTYPE)
    tags["index"] = idx
    tags["amplitude"] = rng.uniform(0.5, 2.0, len(idx))
    tags["phase"] = rng.uniform(-3, 3, len(idx))
    tags["freq"] = rng.uniform(-0.03, 0.03, len(idx)).astype(np.float32)
    tags["time_est"] = rng.uniform(-0.5, 0.5, len(idx))
    tags["flags"] = 1
    return tags


def test_reference_headers_on_the_stub_agree_with_the_oracle(ref_check, tmp_path):
    """The reference's rotator / coarse_frequency_correction / symbol_filter / costas_loop / interpolating_fir_filter /
    pfb_arb_resampler / syncword_wipeoff headers, driven through processBulk() on the stand-in, against the oracle's
    restatement of the same blocks, bit for bit -- on the paths the reference's own qa_*.cpp do not cover (tags into
    SymbolFilter, CFC with delay = 26, a float-rate resampler, phase tags into the Costas loop).  Earns no parity
    credit (the API under the blocks is a stand-in); it catches restatement slips."""
    rng = np.random.default_rng(77)
    n = 30000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for chunk in (4000, 997):
        # Rotator (processOne)
        y, _, c, p = _run_ref(ref_check, tmp_path, "rotator", x, chunk=chunk)
        assert c == p == n and np.array_equal(_bits(y), _bits(orc.rotator(x, np.float32(0.1))))
        # CoarseFrequencyCorrection, delay 26, several set_freq tags incl. two closer than the delay
        tags = _detection_tags([0, 5000, 5010, 12000, 20001, 29990], 1)
        y, t, c, p = _run_ref(ref_check, tmp_path, "cfc", x, tags, chunk)
        want = orc.coarse_frequency_correction(x, tags["index"], tags["freq"], delay=26)
        assert c == p == n and np.array_equal(_bits(y), _bits(want))
        assert np.array_equal(t["index"], tags["index"])                    # forwarded by the default policy
        # SymbolFilter with tags: every clock-phase case of symbol_filter.hpp:130-206
        idx = [0, 1001, 2002, 2500, 3000, 3003, 4444, 5557, 7000, 7001, 9998, 20000]
        tags = _detection_tags(idx, 2)
        tags["time_est"] = [0.1, -0.2, 0.49, -0.5, 0.0, -0.01, 0.3, -0.3, 0.5, 0.2, -0.45, 0.25]
        tags["flags"][4] = 2                                                # not a detection: only re-timed
        rrc, norm = orc.unit_norm_rrc(4)
        pfb = orc.rrc_taps(32.0 / float(norm), 128.0, 1.0, 0.35, 32 * 4 * 11)[:-1]
        y, t, c, p = _run_ref(ref_check, tmp_path, "symbol_filter", x, tags, chunk)
        want, want_tags, want_cons = orc.symbol_filter(x, pfb, 32, 4, 44, tags=tags)
        assert c == want_cons and p == want.size and np.array_equal(_bits(y), _bits(want))
        assert np.array_equal(t["index"], want_tags["index"])
        det = want_tags["flags"] == 1
        for k in ("amplitude", "phase", "time_est"):
            assert np.array_equal(t[k][det], want_tags[k][det]), k
        assert np.array_equal(t["freq"][det], want_tags["freq"][det])
        # CostasLoop, QPSK, phase tags
        tags = _detection_tags([0, 3000, 3001, 17000], 3)
        y, _, c, p = _run_ref(ref_check, tmp_path, "costas", x, tags, chunk)
        want = orc.costas_loop(x, "QPSK", 0.01, tags["index"], tags["phase"])
        assert c == p == n and np.array_equal(_bits(y), _bits(want))
        # InterpolatingFirFilter x4 with the 45-tap RRC
        y, _, c, p = _run_ref(ref_check, tmp_path, "interp_fir", x[:8000], chunk=chunk)
        want = orc.interpolating_fir(x[:8000], 4, orc.rrc_taps(1.0, 4.0, 1.0, 0.35, 44))
        assert c == 8000 and np.array_equal(_bits(y), _bits(want))
        # PfbArbResampler: float rate 1 + 1.2e-6 (TRate = float) and 1.1234 (TRate = double)
        pkg_taps = np.fromfile(os.path.join(ROOT, "gr4-packet-modem_amd", "data", "pfb_arb_taps.f32"), dtype=np.float32)
        for case, rate, dbl in (("arb_float", float(np.float32(1.0) + np.float32(1.2e-6)), False), ("arb_double", 1.1234, True)):
            y, _, c, p = _run_ref(ref_check, tmp_path, case, x, chunk=chunk)
            want, wcons = orc.pfb_arb_resampler(x, rate, pkg_taps, 32, rate_is_double=dbl, out_cap=8 * n + 4096)
            assert c == wcons and p == want.size and np.array_equal(_bits(y), _bits(want)), case
        # SyncwordWipeoff
        sw = np.array([(-1.0 if (i * 7 % 3) else 1.0) for i in range(64)], dtype=np.float32)
        tags = _detection_tags([10, 500, 6000, 29000], 4)
        y, _, c, p = _run_ref(ref_check, tmp_path, "wipeoff", x, tags, chunk)
        assert c == p == n and np.array_equal(_bits(y), _bits(orc.syncword_wipeoff(x, sw, tags["index"])))
