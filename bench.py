#!/usr/bin/env python3
"""bench.py -- RX Msamples/s of the gr4-packet-modem receiver hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1]): one channel per GPU, the full RX front end of
PacketReceiver (packet_receiver.hpp:34-127): SyncwordDetection (overlap-save FFT correlator,
9 frequency bins, detector, tags, delayed pass-through) -> SyncwordDetectionFilter (tag gate)
-> CoarseFrequencyCorrection -> SymbolFilter (32-arm RRC PFB, /4) -> SyncwordWipeoff ->
CostasLoop, on a synthetic 3.2 Msps-shaped burst stream (64-symbol BPSK syncword + 128-symbol header + QPSK payload,
45-tap unit-norm RRC at 4 samples/symbol, per-packet CFO, AWGN) already resident in HBM.
A step is one process() call over one batch of `--items` samples; the stream state carries
from step to step exactly as in the reference block.  With N GPUs every rank runs its own
channel (the path shards by channel, no data-path collective): weak scaling.

Extra objects on the JSON line:
  repeats / values / value_min / value_max
                the timed region (exactly --steps steps between barriers) is run --repeats times back to back;
                `value` and `ms_per_step` are the median region's.
  roofline      the dominant kernel (k_correlate_w64) timed alone with HIP events on the launch
                stream; achieved = 8 B/sample (HBM read, SURVEY.md 8(d) read-only variant)
                x samples per launch / mean launch time, against the 8 TB/s HBM peak; per_bins: the same for
                1 / 3 / 5 / 7 / 9 templates (the rows benchmarks/results.md:37-41 publishes) with the FP32 fraction.
  cpu_baseline  the CPU oracle (kind "port": the reference cannot be built here) on a bounded
                sample of the same stream, after every GPU leg, at N = 1 only; numpy.fft cross-check beside it.
  config2 / config3
                64 channels per GPU through gr4pm_multichannel_receiver, measured after the headline region in the
                same run: BASELINE configs[2] at N = 1, configs[3] (64 x N channels, rank 0's host sample ring
                scattered over the job's backend) at N > 1.
  job           backend, world size and every rank's device name / PCI bus id (gathered over the backend).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3
N_FFT, SPS, BINS = 2048, 4, 4
SHARED_DEVICE_TEST = os.environ.get("GR4PM_BENCH_SHARED_DEVICE_TEST") == "1"  # tests only, see main()
SYNCWORD = np.array(
    [0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 1, 1, 1,
     0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0],
    dtype=np.uint8)


def unit_norm_rrc(pkg):
    """packet_receiver.hpp:60-74"""
    t = pkg.root_raised_cosine(1.0, float(SPS), 1.0, 0.35, SPS * 11)
    norm = np.float32(0.0)
    for v in t:
        norm = np.float32(norm + np.float32(v * v))
    return (t / np.float32(np.sqrt(norm))).astype(np.float32)


def header_symbols(packet_length):
    """the 128 QPSK symbols of a valid packet header (used by --decode-headers): header bytes
    (header_formatter.hpp:104-107), (128, 32) LDPC + repetition (header_fec_encoder.hpp:60-107,
    generator = tests/golden/header_ldpc_generator.npy), CCSDS 131.0-B-5 scrambler from its seed
    (packet_transmitter_pdu.hpp wiring), bit pairs -> (I, Q), 0 -> +"""
    gen = np.load(os.path.join(ROOT, "tests", "golden", "header_ldpc_generator.npy"))
    hdr = [(packet_length >> 8) & 0xFF, packet_length & 0xFF, 0x00, 0x55]
    info = (hdr[0] << 24) | (hdr[1] << 16) | (hdr[2] << 8) | hdr[3]
    bits = [(info >> (31 - i)) & 1 for i in range(32)]
    bits += [bin(info & int(g)).count("1") & 1 for g in gen]
    bits = np.array(bits + bits, dtype=np.uint8)
    reg, seq = 0x18E38, []
    for _ in range(256):
        seq.append(reg & 1)
        reg = ((bin(reg & 0x4001).count("1") & 1) << 16) | (reg >> 1)
    bits ^= np.array(seq, dtype=np.uint8)
    a = np.float32(np.sqrt(0.5))
    return ((1 - 2.0 * bits[0::2]) * a + 1j * (1 - 2.0 * bits[1::2]) * a).astype(np.complex64)


def packet_stream(pkg, n_items, seed, device):
    """--decode-headers: real packets from the package's BurstGenerator (the reference's burst format,
    packet_transmitter_pdu.hpp): 1500 random payload bytes + CRC-32 each, 500-symbol gaps, carrier
    offset 0.01 rad/sample, Es/N0 = 20 dB (the payload is uncoded: at 10 dB no CRC would pass)"""
    rng = np.random.default_rng(seed)
    gen = pkg.BurstGenerator()
    per_packet = (64 + 128 + 1504 * 4 + gen.RAMP_DOWN + gen.FLUSH + 500) * SPS
    n_pkt = n_items // per_packet + 1
    payloads = [rng.integers(0, 256, 1500, dtype=np.uint8).tobytes() for _ in range(n_pkt)]
    x = gen.stream(payloads, np.full(n_pkt, 500 * SPS), freq_error=0.01, esn0_db=20.0, seed=seed, tail=0)
    return x[:n_items].contiguous(), n_pkt


def burst_stream(pkg, n_items, rrc, seed, device, header=None, cfo=None):
    """synthetic 3.2 Msps-shaped bursts, generated on the GPU:
    packets of 64 (BPSK syncword) + 128 (header) + 1504*4 (payload) QPSK symbols, gaps of 500
    zero symbols, CFO uniform in +-0.03 rad/sample per packet (cfo = f: the constant carrier offset f of the whole
    stream instead, config 3's channels), AWGN at Es/N0 = 10 dB."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pkt_syms, gap = 64 + 128 + 1504 * 4, 500
    period = pkt_syms + gap
    n_sym = n_items // SPS + 64
    n_pkt = n_sym // period + 1
    a = np.float32(np.sqrt(0.5))
    bits_i = torch.randint(0, 2, (n_pkt, period), generator=g, device=device)
    bits_q = torch.randint(0, 2, (n_pkt, period), generator=g, device=device)
    sym = torch.complex((1 - 2 * bits_i).float() * a, (1 - 2 * bits_q).float() * a)
    sw = torch.from_numpy(np.where(SYNCWORD == 1, -1.0, 1.0).astype(np.float32)).to(device)
    sym[:, :64] = torch.complex(sw, torch.zeros_like(sw))
    if header is not None:
        sym[:, 64:192] = torch.from_numpy(header).to(device)
    sym[:, pkt_syms:] = 0
    sym = sym.reshape(-1)[:n_sym]
    # pulse shaping with this package's own InterpolatingFirFilter kernel (the TX-side block the
    # reference uses to make its test signals, interpolating_fir_filter.hpp)
    x = pkg.InterpolatingFirFilter(SPS, rrc).process_bulk(sym.to(torch.complex64).contiguous())[:n_items]
    per_packet = (torch.rand(n_pkt, generator=g, device=device) * 0.06 - 0.03).repeat_interleave(period * SPS)[:n_items]
    if cfo is None:
        k = torch.arange(n_items, device=device) % (period * SPS)
        x = x * torch.polar(torch.ones_like(per_packet), per_packet * k)
    else:  # (the draw above keeps the generator's sequence, i.e. the noise, independent of the choice)
        ph = (torch.arange(n_items, device=device, dtype=torch.float64) * float(cfo)) % (2.0 * np.pi)
        x = x * torch.polar(torch.ones(n_items, device=device), ph.to(torch.float32))
    sigma = np.float32(np.sqrt(0.1 / 2.0))  # Es = 1 per symbol -> N0 = 0.1
    noise = torch.complex(torch.randn(n_items, generator=g, device=device) * sigma,
                          torch.randn(n_items, generator=g, device=device) * sigma)
    return (x + noise).to(torch.complex64).contiguous(), n_pkt


def _cpu_leg(piece_path, leg, seconds, workers):
    """`workers` processes (tools/cpu_baseline_worker.py: numpy + the oracle library, no torch, no GPU), each with
    its own detector / chain state over the same samples -- one independent channel per process, the way the path
    shards; rate = samples of all workers / the longest worker time"""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), piece_path, leg, str(seconds)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env) for _ in range(workers)]
    done, tmax = 0, 0.0
    for p in procs:
        out = p.communicate()[0].split()
        if p.returncode == 0 and len(out) == 2:
            done += int(out[0])
            tmax = max(tmax, float(out[1]))
    return (done / tmax / 1e6 if tmax > 0 else 0.0), done, tmax


def cpu_baseline(x_host, rrc, seconds_target=24.0):
    """the CPU oracle (kind "port": the reference cannot be built on the GPU box) on a bounded sample of the same
    workload, ~seconds_target of wall time over five legs: value = the full front end on ALL host cores (one
    channel per process, `cores` processes); beside it the same on one core, the detector alone on one core and on
    all cores, and the detector on the reference benchmark's all-zeros input (benchmark_syncword_detection.cpp:31)."""
    import tempfile
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    leg = seconds_target / 6.0
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm) as d:
        big, small, zeros = os.path.join(d, "big.npy"), os.path.join(d, "small.npy"), os.path.join(d, "zeros.npy")
        np.save(big, x_host[: min(x_host.size, 1 << 24)])     # one-core legs: passes of 2^24 samples
        np.save(small, x_host[: min(x_host.size, 1 << 21)])   # all-core legs: short passes, every worker gets several
        np.save(zeros, np.zeros(1 << 22, dtype=np.complex64))
        fe_all, fe_all_n, fe_all_dt = _cpu_leg(small, "front_end", leg, cores)
        fe_one, _, _ = _cpu_leg(big, "front_end", leg, 1)
        det_one, _, _ = _cpu_leg(big, "detector", leg, 1)
        det_all, _, _ = _cpu_leg(small, "detector", leg, cores)
        det_zero, _, _ = _cpu_leg(zeros, "detector", leg, 1)
    # the FFT cost alone on this host with numpy.fft (pocketfft, complex64 in,
    # complex128 arithmetic): (1 + B) 2048-point transforms per 1752 samples, one core
    blk = np.ascontiguousarray(x_host[: 2048 * 256].reshape(256, 2048))
    n_tr, t_fft = 0, time.perf_counter()
    while time.perf_counter() - t_fft < 1.5:
        np.fft.fft(blk, axis=1)
        n_tr += blk.shape[0]
    t_fft = time.perf_counter() - t_fft
    fft_msps = {str(b): round(n_tr / t_fft / (1 + b) * 1752 / 1e6, 3) for b in (1, 9)}
    return {"value": round(fe_all, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "numpy_fft_bound_one_core": {"transforms_per_s": round(n_tr / t_fft, 1), "msps_at_bins": fft_msps,
                                         "what": "numpy.fft.fft of 2048-point blocks alone: (1 + B) transforms per 1752 samples"},
            "sample": f"full front end (detector 9 bins + CFC + SymbolFilter + wipe-off + Costas), CPU oracle "
                      f"oracle/gr4pm_oracle.cpp, {cores} processes x passes over the first {min(x_host.size, 1 << 21)} "
                      f"samples of the same burst stream ({fe_all_n} samples in {fe_all_dt:.1f} s); four more legs of "
                      f"~{leg:.0f} s each",
            "front_end_one_core": round(fe_one, 3),
            "detector_one_core": round(det_one, 3),
            "detector_all_cores": round(det_all, 3),
            "detector_zeros_input_one_core": round(det_zero, 3),
            "reference_published": "13 Msps detector / 6-8 Msps receiver at 9 bins on a Ryzen 7 5800X "
                                   "(benchmarks/results.md:41,51)"}


def correlator_kernel():
    """name of the correlator kernel a default-size detector runs (GR4PM_CORRELATOR selects the round-1 ones)"""
    return {"wave": "k_correlate", "pair": "k_correlate_pair"}.get(os.environ.get("GR4PM_CORRELATOR", ""), "k_correlate_w64")


MEASURED_TRAFFIC = None  # measure_traffic_in_run()'s result, when it ran


def measure_traffic_in_run(items, bins=BINS, timeout=150):
    """roofline.traffic measured in THIS run (round 6; VERDICT round 5: "a builder-side number riding in a driver record").
    Two rocprofv3 passes -- `--kernel-trace --pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`, separately, as
    MI355X_MICROARCH.md's HBM section prescribes -- over tools/bench_correlate.py (the correlator alone on `items` samples
    at nine bins: the launch the roofline leg times), started as child processes `rocprofv3 ... -- python3 <program>`
    BEFORE this process makes any GPU call.  HBM bytes per launch = 2 x FETCH_SIZE (gfx950: the counter reports half of
    a wide streaming read) + WRITE_SIZE, both in KiB, summed over the XCDs' rows of a dispatch, mean over the launches.
    None when rocprofv3 is not on PATH, a pass fails or takes too long: the committed PMC file is used then."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    # (this process is itself running under a profiler: no profiler inside a profiler)
    if any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    kernel = correlator_kernel()
    kib = {}
    tmp = tempfile.mkdtemp(prefix="gr4pm_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.join(ROOT, "tools", "bench_correlate.py"), str(items), "3", str(bins)]
            env = dict(os.environ, WARM="2", TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd="/tmp")
            except (subprocess.TimeoutExpired, OSError):
                return None
            if r.returncode != 0:
                return None
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            per[(f, row["Dispatch_Id"])] = per.get((f, row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
            if not per:
                return None
            kib[counter] = sum(per.values()) / len(per)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    samples = ((items - N_FFT) // 1752 + 1) * 1752
    read_b, write_b = 2.0 * kib["FETCH_SIZE"] * 1024.0, kib["WRITE_SIZE"] * 1024.0
    return {"bytes_per_launch": read_b + write_b, "bytes_per_sample": (read_b + write_b) / samples, "samples": samples,
            "read_bytes_corrected": read_b, "write_bytes": write_b, "kernel": kernel}


def pmc_traffic(samples, kernel="k_correlate_w64"):
    """roofline.traffic: HBM bytes per launch.  Measured in this run when measure_traffic_in_run() ran (N = 1, the headline
    workload); else from the PMC passes committed under profiles/ (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note
    + WRITE_SIZE, per sample of the profiled launch, scaled to this launch; `traffic_measured: false`).  null when the file
    is missing or describes another kernel."""
    import glob
    if MEASURED_TRAFFIC is not None and MEASURED_TRAFFIC["kernel"] == kernel:
        t = MEASURED_TRAFFIC
        return {"traffic": round(t["bytes_per_sample"] * samples), "traffic_measured": True,
                "traffic_source": f"this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes over "
                                  f"tools/bench_correlate.py, {t['samples']} samples per launch, before this process touched the "
                                  f"GPU): 2 x {t['read_bytes_corrected'] / 2 / t['samples']:.3f} + {t['write_bytes'] / t['samples']:.3f} "
                                  f"= {t['bytes_per_sample']:.3f} B/sample"}
    # the newest round's file (profiles/r<N>_k_correlate_hbm_traffic.json)
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_k_correlate_hbm_traffic.json")),
                   key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    path = found[-1] if found else os.path.join(ROOT, "profiles", "r4_k_correlate_hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if not str(t.get("kernel", "")).startswith(kernel + " "):
            return {"traffic": None, "traffic_measured": False,
                    "traffic_source": f"profiles/{os.path.basename(path)} describes {t.get('kernel', '?')}, "
                                                       f"this run launched {kernel}"}
        return {"traffic": round(float(t["traffic_bytes_per_sample"]) * samples),
                "traffic_measured": False,  # in THIS run: the figure is the committed PMC pass scaled to this launch
                "traffic_source": f"profiles/{os.path.basename(path)}: {t['traffic_bytes_per_sample']:.3f} B/sample "
                                  f"(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel {t.get('kernel', '?')})"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None, "traffic_measured": False, "traffic_source": f"profiles/{os.path.basename(path)} missing"}


def config2_reference():
    """the configs[2] sub-record of the latest committed single-GPU bench line (profiles/r<N>_bench.json)"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json"))):
        try:
            d = json.loads(open(path).read().strip().splitlines()[-1])
            if d.get("n_gpus") == 1 and "config2" in d:
                best = {"value": d["config2"]["value"], "source": os.path.relpath(path, ROOT)}
        except (ValueError, OSError, IndexError, KeyError):
            continue
    return best


def channel_bank(x, n_channels):
    """configs[2]/[3]: C channels from one burst stream: channel c is the stream rotated by 997 c
    samples with a carrier offset of -0.04 + 0.08 c / (C - 1) rad/sample on top"""
    n = x.numel()
    k = torch.arange(n, device=x.device, dtype=torch.float32)
    xs = torch.empty((n_channels, n), dtype=torch.complex64, device=x.device)
    for c in range(n_channels):
        f = -0.04 + 0.08 * c / max(n_channels - 1, 1)
        xs[c] = x.roll(997 * c) * torch.polar(torch.ones_like(k), (f * k) % (2 * np.pi))
    return xs


def channel_bank_config3(pkg, n_items, rrc, n_channels, device, seed0=0):
    """configs[2] / [3] as SURVEY.md 8(d) config 3 defines the channels: channel c is its own burst stream (seed
    seed0 + c) with a carrier offset of -0.04 + 0.08 c / (C - 1) rad/sample -- inside the +-4-bin search range
    (+-0.0423), so every channel's packets can be found.  (channel_bank() puts that sweep ON TOP of a stream whose
    packets already carry +-0.03: its edge channels lie outside the range and lose detections, which the parity test of
    the multi-channel receiver wants; as a throughput workload it measured missed detections -- serial segments five
    packets long -- rather than the configuration.)"""
    xs = torch.empty((n_channels, n_items), dtype=torch.complex64, device=device)
    for c in range(n_channels):
        f = -0.04 + 0.08 * c / max(n_channels - 1, 1)
        xs[c] = burst_stream(pkg, n_items, rrc, seed=seed0 + c, device=device, cfo=f)[0]
    return xs


def host_sample_ring(world, shape):
    """configs[3]: rank 0's host sample ring [world, channels, n] (pinned when the host allows it)"""
    try:
        return torch.empty((world,) + tuple(shape), dtype=torch.complex64, pin_memory=True), "pinned"
    except RuntimeError:
        return torch.empty((world,) + tuple(shape), dtype=torch.complex64), "pageable"


def scatter_channels(dist, make_all, n_items, device, rank, world):
    """the only collective of this workload is the initial sample scatter -- rank 0
    holds every channel's samples ([world, n_items] complex64) and sends one channel to each
    rank (RCCL send/recv over xGMI on the GPUs; any torch.distributed backend works)."""
    shape = (n_items,) if isinstance(n_items, int) else tuple(n_items)  # per-rank shape: [n] or [channels, n]
    mine = torch.empty(shape, dtype=torch.complex64, device=device)
    mine_f = torch.view_as_real(mine)  # interleaved float32 pairs: every backend moves floats
    via_host = dist.get_backend() == "gloo" and device.type == "cuda"  # the shared-device test: gloo scatters host tensors
    recv = torch.empty(shape + (2,), dtype=torch.float32) if via_host else mine_f
    parts = None
    if rank == 0:
        allx = make_all()
        assert tuple(allx.shape) == (world,) + shape
        parts = [torch.view_as_real(allx[r].contiguous()) for r in range(world)]
        if via_host:
            parts = [p.cpu() for p in parts]
    # the collective alone, between two barriers (filling rank 0's ring is not the scatter's time)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    dist.scatter(recv, parts, src=0)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    if via_host:
        mine_f.copy_(recv)
    nbytes = 8 * int(np.prod(shape)) * (world - 1)  # what leaves rank 0 (its own part stays)
    LAST_SCATTER.update(seconds=round(dt, 4), bytes_from_rank0=nbytes, gbs=round(nbytes / dt / 1e9, 2),
                        backend=dist.get_backend())
    return mine


LAST_SCATTER = {}


def aggregate(dist, dt, consumed, device):
    """whole-job numbers over all ranks: time = MAX over ranks, items = SUM over ranks (each
    rank runs its own channel: weak scaling, no data-path collective).  Works with any
    torch.distributed backend (RCCL on the GPUs, gloo in the CPU test)."""
    if dist is None:
        return dt, consumed
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    csum = torch.tensor([consumed], dtype=torch.float64, device=device)
    dist.all_reduce(csum, op=dist.ReduceOp.SUM)
    return tmax.item(), csum.item()


def per_rank(dist, value, device):
    """every rank's own number (a region's time, ...) in rank order, on every rank: the line shows which GPU of a job was
    the slow one instead of only the MAX"""
    if dist is None:
        return [float(value)]
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    every = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return [float(t.item()) for t in every]


def launch_ranks(n, selfcheck_first=True):
    """one worker process per GPU, rendezvous on 127.0.0.1 (what torch.distributed.run would set up); the parent
    only waits -- it never initialises the GPU, and every rank is a FRESH child (a process that has touched the GPU
    is never re-executed).  A rank that ends with an error ends the run: the ranks still alive (blocked in a
    collective the failed rank never joins) are terminated -- the processes started here, by handle -- and the
    launcher returns non-zero; no line can come out of a job that lost a rank.
    First contact: before the real job the same ranks run `--selfcheck` (rendezvous with a 120-s timeout, identities,
    a 2^22-item scatter, two steps of the receiver, no CPU legs: seconds); when that fails, the failing rank's stderr
    is shown and the real job is not started."""
    if selfcheck_first and "--selfcheck" not in sys.argv:
        rc = _launch_once(n, sys.argv[1:] + ["--selfcheck"], capture=True)
        if rc != 0:
            print(f"bench.py: the {n}-rank self-check failed (exit {rc}); not starting the job", file=sys.stderr)
            return rc
    return _launch_once(n, sys.argv[1:], capture=False)


def _launch_once(n, argv, capture):
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        log = tempfile.TemporaryFile(mode="w+") if capture else None
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.DEVNULL if capture else None, stderr=log))
    worst = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = max(worst, abs(rc))
                print(f"bench.py: rank {procs.index(p)} exited with {rc}; stopping the other ranks", file=sys.stderr)
                failed_first = procs.index(p)
                for q in alive:
                    q.terminate()
                for q in alive:
                    try:
                        q.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        q.kill()
                alive = []
                if capture:  # the self-check's output is only shown when it fails: the failing rank's first
                    for r in [failed_first] + [k for k in range(n) if k != failed_first]:
                        logs[r].seek(0)
                        text = logs[r].read()[-3000:]
                        if text.strip():
                            print(f"---- stderr of rank {r} ----\n{text}", file=sys.stderr)
                break
    return worst


def rank_identities(dist, device, world):
    """who took part: every rank's device name and PCI bus id, gathered over the job's own backend, so that the line
    shows N distinct GPUs joined one group (and which backend carried the scatter)"""
    idx = device.index if device.type == "cuda" else 0
    if device.type == "cuda":
        pr = torch.cuda.get_device_properties(idx)
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        me = {"rank": int(os.environ.get("RANK", "0")), "device": idx, "name": pr.name, "pci_bus_id": bus,
              "uuid": str(getattr(pr, "uuid", ""))}
    else:
        me = {"rank": int(os.environ.get("RANK", "0")), "device": "cpu", "name": "cpu", "pci_bus_id": str(os.getpid()), "uuid": ""}
    if dist is None:
        return {"backend": None, "world": 1, "ranks": [me]}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    return {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks": everyone}


def check_distinct_devices(job, world):
    """every rank evaluates the gathered list (nobody is left waiting in a barrier for a rank that quit): N ranks must
    sit on N different devices"""
    ids = {(r["pci_bus_id"], r["uuid"]) for r in job["ranks"]}
    if SHARED_DEVICE_TEST and job["ranks"][0]["device"] != "cpu":
        job["shared_device_test"] = True
        return
    if len(ids) != world:
        raise SystemExit(f"bench.py: {world} ranks on {len(ids)} distinct devices: {job['ranks']}")


def init_ranks(backend, local_rank=None):
    """rendezvous with a bounded wait (a wedged rank costs two minutes, not the ten-minute default)"""
    import datetime
    import torch.distributed as dist
    kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("GR4PM_DIST_TIMEOUT_S", "120")))}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, **kw)
    return dist


def scatter_budget(dist, rank, world, per_rank_bytes, device, host_ring):
    """what rank 0 needs for a scatter, against what it has, BEFORE anything is allocated: `world` slabs on the device
    (the ring uploaded for the scatter) + its own, and with host_ring the same bytes of (pinned) host memory.  Every
    rank gets the verdict and leaves together when it does not fit."""
    verdict = [None]
    if rank == 0:
        need_dev = per_rank_bytes * (world + 1)
        free_dev = torch.cuda.mem_get_info(device)[0] if device.type == "cuda" else 1 << 62
        need_host = per_rank_bytes * world if host_ring else 0
        try:
            import psutil
            free_host = psutil.virtual_memory().available
        except ImportError:
            free_host = 1 << 62
        # (tests only: what the check sees as available host / free device memory, so that the refusal is exercised at
        # configs[3]'s real size on machines that do have the memory)
        if os.environ.get("GR4PM_BENCH_TEST_HOST_AVAILABLE"):
            free_host = int(os.environ["GR4PM_BENCH_TEST_HOST_AVAILABLE"])
        if os.environ.get("GR4PM_BENCH_TEST_DEVICE_FREE"):
            free_dev = int(os.environ["GR4PM_BENCH_TEST_DEVICE_FREE"])
        msg = (f"scatter budget on rank 0: device {need_dev / 2**30:.1f} GiB needed / {free_dev / 2**30:.1f} GiB free, "
               f"host {need_host / 2**30:.1f} GiB needed / {free_host / 2**30:.1f} GiB available")
        print("bench.py:", msg, file=sys.stderr)
        ok = need_dev < 0.9 * free_dev and need_host < 0.8 * free_host
        verdict = [None if ok else msg]
    dist.broadcast_object_list(verdict, src=0)
    if verdict[0] is not None:
        raise SystemExit(f"bench.py: rank 0 cannot hold the sample ring of {world} ranks: {verdict[0]}")


def selfcheck(pkg, dist, device, rank, world, rrc):
    """first contact of an N-rank job, seconds long: identities (N distinct devices), one scatter of 2^22-item channels
    from rank 0, two batches through the native receiver on what arrived, detections counted on every rank (SUM > 0,
    MIN > 0).  Any failure ends the rank with a message; the launcher ends the job."""
    job = rank_identities(dist, device, world)
    check_distinct_devices(job, world)
    n = 1 << 22
    def make_all():
        return torch.stack([burst_stream(pkg, n, rrc, seed=77 + r, device=device)[0] for r in range(world)])
    x = scatter_channels(dist, make_all, n, device, rank, world) if dist else burst_stream(pkg, n, rrc, seed=77, device=device)[0]
    rx = pkg.NativePacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=n, tags_cap=1024, pipelined=True)
    tags = 0
    for _ in range(2):
        r = rx.process_bulk(x, 1500)
        if r is not None:
            tags += int(r["tags"].size)
    for r in rx.flush():
        tags += int(r["tags"].size)
    del rx
    t = torch.tensor([float(tags)], dtype=torch.float64, device="cpu" if (dist and dist.get_backend() == "gloo") else device)
    if dist:
        lo = t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if lo.item() <= 0:
            raise SystemExit(f"bench.py self-check: a rank found no packet in its scattered channel (rank {rank}: {tags})")
    elif tags <= 0:
        raise SystemExit("bench.py self-check: no packet found")
    return {"ranks": world, "tags": int(t.item()), "job": job}


def dry_run(args):
    """--dry-run: the N > 1 plumbing (rendezvous, the channel scatter, MAX / SUM aggregation, rank identities,
    rank 0 prints) on gloo.  GR4PM_BENCH_TEST_FAIL_RANK=r makes rank r die just before the scatter
    (tests/test_distributed_cpu.py: the launcher must end the job non-zero instead of hanging)."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cpu")
    if world > 1:
        init_ranks("gloo")
        # first thing after the rendezvous, evaluated on EVERY rank (GR4PM_BENCH_TEST_SAME_DEVICE: all ranks report
        # one device -- the test of exactly that)
        early = rank_identities(dist, device, world)
        if os.environ.get("GR4PM_BENCH_TEST_SAME_DEVICE"):
            for r in early["ranks"]:
                r["pci_bus_id"] = "same"
        check_distinct_devices(early, world)
    if os.environ.get("GR4PM_BENCH_TEST_FAIL_RANK") == str(rank):
        # tests: die in the self-check pass (default) or, with GR4PM_BENCH_TEST_FAIL_IN_JOB, only in the job behind it
        if bool(os.environ.get("GR4PM_BENCH_TEST_FAIL_IN_JOB")) != bool(args.selfcheck):
            print(f"rank {rank}: forced failure (test)", file=sys.stderr, flush=True)
            os._exit(3)
    scattered_ok = None
    if world > 1:
        mine = scatter_channels(dist, lambda: torch.stack([torch.full((64,), complex(r, 1), dtype=torch.complex64)
                                                           for r in range(world)]), 64, device, rank, world)
        ok = torch.tensor([1.0 if bool((mine == complex(rank, 1)).all()) else 0.0], dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        scattered_ok = bool(ok.item() == 1.0)
    every = per_rank(dist if world > 1 else None, 1.0 + rank, device)
    dt, total = aggregate(dist if world > 1 else None, 1.0 + rank, 1000.0 * (rank + 1), device)
    ids = rank_identities(dist if world > 1 else None, device, world)
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": total / dt, "n_gpus": world, "gpus_requested": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "scatter_ok": scattered_ok,
                          "ms_per_step_per_rank": every, "scatter": dict(LAST_SCATTER) or None, "job": ids}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def config5(args):
    """--config 5 = BASELINE configs[4], the stress shape: 1 channel, fft_size 4096, RRC 1024 requested ->
    1025 taps -> syncword of 63 * 4 + 1025 = 1277 samples, stride 2820, 9 bins;
    SyncwordDetection over a resident noise + burst stream, and beside it the 1025-tap shaping /
    matched-filter leg (InterpolatingFirFilter x4, interpolating_fir_filter.hpp:76-102).  One rank."""
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    pkg = ge.load_package()
    n, nfft = args.items, 4096
    rrc = pkg.root_raised_cosine(1.0, float(SPS), 1.0, 0.35, 1024)
    rrc = (rrc / np.sqrt(np.sum(rrc.astype(np.float64) ** 2))).astype(np.float32)
    L = 63 * SPS + rrc.size
    S = nfft - L + 1
    g = torch.Generator(device=device)
    g.manual_seed(5)
    # bursts: syncword + random BPSK symbols shaped with the 1025-tap RRC, every 16384 symbols, in AWGN
    n_sym = n // SPS
    bits = torch.randint(0, 2, (n_sym,), generator=g, device=device)
    sw = torch.from_numpy(SYNCWORD.astype(np.int64)).to(device)
    starts = torch.arange(2000, n_sym - 64, 16384, device=device)
    for k in range(64):
        bits[starts + k] = sw[k]
    sym = torch.complex((1 - 2 * bits).float(), torch.zeros(n_sym, device=device))
    fir = pkg.InterpolatingFirFilter(SPS, rrc)
    x = fir.process_bulk(sym.contiguous())[:n]
    x = (x + torch.view_as_complex(0.05 * torch.randn((n, 2), device=device, generator=g))).contiguous()
    bpsk = np.array([1, -1], dtype=np.complex64)
    # power_threshold 30, not the receiver's 9.5: with a 1025-tap template the correlation power is smooth over
    # hundreds of lags, a 1537-item history holds few independent values, and at 9.5 the detector (reference and
    # oracle alike: 305 tags instead of 8 on 480k samples) fires on plain data
    sd = pkg.SyncwordDetection(rrc, SYNCWORD, bpsk, -BINS, BINS, fft_size=nfft, power_threshold=30.0, max_items=n)
    cap = 1 << 16

    def step():
        st, _, tags, nd = sd.process_bulk(x, want_output=False, tags_cap=cap)
        return nd, tags.size

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = n_tags = 0
    for _ in range(args.steps):
        nd, nt = step()
        done += nd
        n_tags += nt
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the 1025-tap filter leg, timed beside it (symbols in, 4 samples per symbol out)
    for _ in range(2):
        fir.process_bulk(sym)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        y = fir.process_bulk(sym)
    torch.cuda.synchronize()
    fir_rate = 5 * y.numel() / (time.perf_counter() - t1) / 1e6
    # the receiving side of the same pulse: SymbolFilter with 32 arms x 1025 taps (symbol_filter.hpp:208-238),
    # samples in, one symbol per 4 samples out
    pfb = pkg.root_raised_cosine(32.0, 32.0 * SPS, 1.0, 0.35, 32 * 1024)[: 32 * 1025]
    sf = pkg.SymbolFilter(pfb, 32, SPS, delay=1025)
    xs = x[: 1 << 24]
    for _ in range(2):
        sf.process_bulk(xs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(5):
        sf.process_bulk(xs)
    torch.cuda.synchronize()
    symf_rate = 5 * xs.numel() / (time.perf_counter() - t2) / 1e6
    # roofline of the correlator kernel for this size
    reps = 5
    sd.correlate_only(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sd.correlate_only(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    samples = ((n - nfft) // S + 1) * S
    flops = 950.0 * samples  # N = 4096, B = 9
    achieved = 8.0 * samples / (ms * 1e-3) / 1e9
    line = {"metric": "RX Msamples/s (syncword-detect + RRC chain)", "value": round(done / dt / 1e6, 2),
            "unit": "Msamples/s", "n_gpus": 1, "gpus_requested": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[4]: 1 channel, SyncwordDetection with fft_size 4096, 1025-tap RRC "
                                   f"(syncword {L} samples, stride {S}), 9 bins, power_threshold 30, resident burst + AWGN stream",
                       "items_per_step_per_gpu": n, "freq_bins": 2 * BINS + 1, "tags_per_step": n_tags // max(args.steps, 1),
                       "fir_1025_taps_x4_msps_out": round(fir_rate, 1),
                       "symbol_filter_32x1025_taps_msps_in": round(symf_rate, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "k_correlate_4096",
                         "launch_ms": round(ms, 4), "samples_per_launch": samples, "alg_bytes_per_sample": 8,
                         "fp32_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                         "fp32_frac": round(flops / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)},
            "cpu_baseline": None}
    print(json.dumps(line))


def config5_stream(pkg, n, device, seed=5, chunk=1 << 26):
    """the configs[4] input: bursts (syncword + random BPSK, shaped with the 1025-tap RRC at 4 samples per symbol) every
    16384 symbols in AWGN, n samples, generated on the GPU chunk by chunk (the noise is added in place)"""
    rrc = pkg.root_raised_cosine(1.0, float(SPS), 1.0, 0.35, 1024)
    rrc = (rrc / np.sqrt(np.sum(rrc.astype(np.float64) ** 2))).astype(np.float32)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_sym = n // SPS
    bits = torch.randint(0, 2, (n_sym,), generator=g, device=device, dtype=torch.int8)
    sw = torch.from_numpy(SYNCWORD.astype(np.int8)).to(device)
    starts = torch.arange(2000, n_sym - 64, 16384, device=device)
    for k in range(64):
        bits[starts + k] = sw[k]
    fir = pkg.InterpolatingFirFilter(SPS, rrc)
    x = torch.empty(n, dtype=torch.complex64, device=device)
    sym_chunk = chunk // SPS
    for a in range(0, n_sym, sym_chunk):  # the filter carries its history from call to call: one continuous stream
        b = min(a + sym_chunk, n_sym)
        sym = torch.complex((1 - 2 * bits[a:b]).float(), torch.zeros(b - a, device=device))
        y = fir.process_bulk(sym.contiguous())
        y += torch.view_as_complex(0.05 * torch.randn((y.numel(), 2), device=device, generator=g))
        x[a * SPS:a * SPS + y.numel()] = y
    return x, rrc, fir


def config5_windows(total, window, nfft, stride):
    """the streamed calls over the configs[4] ring: (position, items offered) -- `window` items offered, whole strides
    consumed (syncword_detection.hpp:238), the next window starts where the call stopped"""
    pos, out = 0, []
    while total - pos >= nfft:
        take = min(window, total - pos)
        out.append((pos, take))
        pos += ((take - nfft) // stride + 1) * stride
    return out


def config5_leg(pkg, device, total=1 << 30, window=1 << 28, passes=2):
    """BASELINE configs[4] as SURVEY.md 8(d) config 5 defines it, in the default run: 1 channel, fft_size 4096, 1025-tap
    RRC (syncword 63 * 4 + 1025 = 1277 samples, stride 2820), B in {1, 9} bins, 2^30 samples STREAMED through a device
    ring: SyncwordDetection takes them window by window (2^28 items offered, whole strides consumed, the next window
    starts where the call stopped and is announced a call ahead: gr4pm_syncword_detection_hint_next), then the
    receiving side of the same pulse, SymbolFilter with 32 arms x 1025 taps (symbol_filter.hpp:208-238), over the same
    ring.  `value` = 2^30 samples / (detector at nine bins + symbol filter), one after the other."""
    nfft = 4096
    x, rrc, _fir = config5_stream(pkg, total, device)
    del _fir
    L = 63 * SPS + rrc.size
    S = nfft - L + 1
    bpsk = np.array([1, -1], dtype=np.complex64)

    wins = config5_windows(total, window, nfft, S)
    per_bins = {}
    for b in (0, BINS):
        # power_threshold 30 (nine bins) / 60 (one), not the receiver's 9.5: with a 1025-tap template the correlation
        # power is smooth over hundreds of lags, a 1537-item history holds few independent values, and at 9.5 the
        # detector (reference and oracle alike) fires on plain data.  One bin has no maximum over bins lifting the
        # history's median, so the same stream needs twice the threshold for the same tags (measured on 2^26 samples
        # with 1024 bursts: 9 bins 30 -> 1039 tags, 1 bin 30 -> 8092, 1 bin 60 -> 1039; every burst found in all three)
        thr = 30.0 if b else 60.0
        sd = pkg.SyncwordDetection(rrc, SYNCWORD, bpsk, -b, b, fft_size=nfft, power_threshold=thr, max_items=window)

        def one_pass(indices=None):
            done = tags = 0
            for k, (pos, take) in enumerate(wins):
                nxt = x[wins[k + 1][0]:wins[k + 1][0] + wins[k + 1][1]] if k + 1 < len(wins) else None
                _, _, t, nd = sd.process_bulk(x[pos:pos + take], want_output=False, tags_cap=1 << 17, next_x=nxt)
                assert nd == ((take - nfft) // S + 1) * S
                if indices is not None:
                    indices.append(t["index"].astype(np.int64) + pos)
                done += nd
                tags += t.size
            return done, tags
        # the warm-up pass, outside the timed region, is also the check that the measured path does the detector's work:
        # every burst config5_stream placed (a syncword every 16384 symbols from symbol 2000) is found at 2T + 1 + 4 x
        # its symbol index, and the detections beside them (plain data, see the threshold note above) stay under 2 %
        idx = []
        done0, _ = one_pass(idx)
        idx = np.concatenate(idx)
        bursts = 1537 + 4 * np.arange(2000, total // 4 - 64, 16384)
        bursts = bursts[bursts < done0 - 1537]
        found = int(np.isin(bursts, idx).sum())
        if found != bursts.size or idx.size - found > 0.02 * bursts.size:
            raise SystemExit(f"bench.py: config5 at {2 * b + 1} bins: {found} of {bursts.size} bursts found, "
                             f"{idx.size - found} other detections: the measured path does not do the detector's work")
        sd.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = tags = 0
        for _ in range(passes):
            d, t = one_pass()
            sd.reset()  # the ring wraps: a new stream
            done += d
            tags += t
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the correlator alone on one window: HIP events on the launch stream
        xs = x[:window]
        for _ in range(ROOF_WARM // 3):
            sd.correlate_only(xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            sd.correlate_only(xs)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        samples = ((window - nfft) // S + 1) * S
        fl = correlator_flops_per_sample(2 * b + 1, nfft, S) * samples / (ms * 1e-3) / 1e12
        per_bins[str(2 * b + 1)] = {
            "value": round(done / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_2^30": round(dt / passes * 1e3, 3),
            "tags_per_2^30": tags // passes, "bursts_in_2^30": int(bursts.size), "bursts_found": found,
            "power_threshold": thr, "reference_default_power_threshold": 9.5,
            "roofline": {"bound": "hbm", "kernel": "k_correlate_4096", "launch_ms": round(ms, 4), "samples_per_launch": samples,
                         "achieved": round(8.0 * samples / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(8.0 * samples / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "alg_bytes_per_sample": 8,
                         "traffic": None, "fp32_tflops": round(fl, 2), "fp32_frac": round(fl / FP32_PEAK_TFLOPS, 4),
                         "ceiling": roofline_ceiling(2 * b + 1, nfft, S)["frac"]}}
        del sd
    pfb = pkg.root_raised_cosine(32.0, 32.0 * SPS, 1.0, 0.35, 32 * 1024)[: 32 * 1025]
    sf = pkg.SymbolFilter(pfb, 32, SPS, delay=1025)

    def symf_pass():
        n = 0
        for a in range(0, total, window):
            _, _, c = sf.process_bulk(x[a:a + window])
            n += c
        return n
    symf_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_sf = sum(symf_pass() for _ in range(passes))
    torch.cuda.synchronize()
    dt_sf = time.perf_counter() - t0
    sf_rate = n_sf / dt_sf / 1e6
    det9 = per_bins[str(2 * BINS + 1)]["value"]
    both = 1.0 / (1.0 / det9 + 1.0 / sf_rate)
    return {"workload": f"configs[4] (SURVEY.md 8(d) config 5): 1 channel, fft_size 4096, 1025-tap RRC (syncword {L} samples, "
                        f"stride {S}), 2^30 samples streamed through a device ring in {len(wins)} windows of "
                        "2^28 offered items (look-ahead one window ahead); SyncwordDetection at 1 and 9 bins, then "
                        "SymbolFilter 32 arms x 1025 taps over the same ring.  DEVIATION from the reference's defaults: "
                        "power_threshold 60 (one bin) / 30 (nine bins) instead of 9.5 -- under a 1025-tap template the "
                        "correlation power is smooth over hundreds of lags and 9.5 fires on plain data (same tags at both "
                        "settings on this stream: every burst found, other detections under 2 %)",
            "value": round(both, 2), "unit": "Msamples/s", "samples_per_pass": total, "passes": passes, "windows": len(wins),
            "per_bins": per_bins,
            "symbol_filter_32x1025": {"value": round(sf_rate, 2), "unit": "Msamples/s in",
                                      "fp32_tflops": round(1025.0 * sf_rate * 1e6 / 1e12, 2),
                                      "fp32_frac": round(1025.0 * sf_rate * 1e6 / 1e12 / FP32_PEAK_TFLOPS, 4)}}


def host_stream_leg(pkg, device, rrc, chunk=1 << 25, host_chunks=4, total=1 << 29):
    """PCIe-inclusive rate (never `value`): the headline front end fed from HOST memory, the way a GR4 port buffer or an
    SDR driver hands samples over.  A pinned host ring of `host_chunks` chunks holds a burst + AWGN stream; every chunk
    travels to one of six device slots on a copy stream (hipMemcpyAsync, two chunks ahead of the receiver) while the
    pipelined native receiver works on the chunks before it; `total` samples = the host ring several times over.
    Reported beside it: the same copies with no receiver behind them (what the link gives), so that the line says how
    much of the link the receiver keeps busy."""
    n_slots = 6
    x, n_pkt = burst_stream(pkg, chunk * host_chunks, rrc, seed=77, device=device)
    host = torch.empty(chunk * host_chunks, dtype=torch.complex64).pin_memory()
    host.copy_(x)
    del x
    slots = [torch.empty(chunk, dtype=torch.complex64, device=device) for _ in range(n_slots)]
    copy_stream = torch.cuda.Stream()
    events = [torch.cuda.Event() for _ in range(n_slots)]
    n_chunks = total // chunk

    def upload(i):
        src = host[(i % host_chunks) * chunk:(i % host_chunks + 1) * chunk]
        with torch.cuda.stream(copy_stream):
            slots[i % n_slots].copy_(src, non_blocking=True)
            events[i % n_slots].record(copy_stream)

    # the link alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_chunks):
        upload(i)
    torch.cuda.synchronize()
    dt_link = time.perf_counter() - t0
    rx = pkg.NativePacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=chunk, tags_cap=max(64, 2 * n_pkt + 64),
                                  pipelined=True, output_ring=True)

    def run(n):
        done = tags = 0
        upload(0)
        upload(1)
        for i in range(n):
            if i + 2 < n:
                upload(i + 2)  # slot (i + 2) % 6 held batch i - 4, collected by the process_bulk call of chunk i - 1
            events[i % n_slots].synchronize()
            res = rx.process_bulk(slots[i % n_slots], 1500)
            if res is not None:
                done += res["consumed"]
                tags += res["tags"].size
        for res in rx.flush():
            done += res["consumed"]
            tags += res["tags"].size
        return done, tags
    run(n_slots)  # warm-up: every slot and every output ring entry has been used
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, tags = run(n_chunks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del rx
    check_tag_count("host-stream leg", tags, n_chunks, chunk)
    return {"workload": f"configs[1] front end fed from pinned host memory: {n_chunks} chunks of 2^{chunk.bit_length() - 1} samples, "
                        f"hipMemcpyAsync two chunks ahead into {n_slots} device slots, pipelined native receiver behind them",
            "value": round(done / dt / 1e6, 2), "unit": "Msamples/s", "samples": done, "tags": tags,
            "h2d_gbs": round(8.0 * done / dt / 1e9, 2),
            "link_only": {"value": round(n_chunks * chunk / dt_link / 1e6, 2), "unit": "Msamples/s",
                          "h2d_gbs": round(8.0 * n_chunks * chunk / dt_link / 1e9, 2)},
            "note": "PCIe-inclusive; not `value` (inputs of the headline are resident in HBM when the timed region starts)"}


SPARSE_STREAMS = ("zeros", "awgn", "one_packet_per_2^20", "dense_packets")


def sparse_leg(pkg, device, rrc, n=1 << 28, passes=48, streams=SPARSE_STREAMS):
    """What packet density does to the rate (never `value`): the whole receiver in decode_headers mode -- the reference's
    PacketReceiver wiring, IQ in, CRC-checked packets out; what benchmarks/benchmark_packet_receiver.cpp runs -- over 2^28
    resident samples of (i) zeros (that benchmark's own input, benchmarks/README.md:49-53, results.md:45-51: 6-8 Msps
    at nine templates on eight CPU cores), (ii) AWGN only, (iii) one 1500-byte packet per 2^20 samples in AWGN (Es/N0 20
    dB), (iv) such packets back to back with 500-symbol gaps (the density of the headline's stream: what the whole
    receiver, header decode and payload tail included, does where the headline times the front end).  The frequency
    correction's phasor and the Costas loop are recurrences that are replayed in the reference's
    rounding: a stretch between two syncword tags is ONE serial chain.  (i) and (ii) never see a tag: the phasor sits at
    its fixed point (1, -0) (no chain at all) and PayloadMetadataInsert passes no symbol on to the Costas loop.  (iii): 256
    chains of 2^20 items side by side per pass.  `detector_alone` is SyncwordDetection on the same stream;
    `behind_the_detector_share` = 1 - detector_alone time / receiver time, i.e. the share of the serial part."""
    hist = 2 * 768 + 1
    bpsk = np.array([1, -1], dtype=np.complex64)
    rows = {}
    g = torch.Generator(device=device)
    g.manual_seed(99)

    def make(kind):
        if kind == "zeros":
            return torch.zeros(n, dtype=torch.complex64, device=device), 0
        sigma = np.float32(np.sqrt(0.1 / 2.0))  # the headline stream's noise floor (Es/N0 = 10 dB at Es = 1)
        if kind == "awgn":
            return torch.complex(torch.randn(n, generator=g, device=device) * sigma,
                                 torch.randn(n, generator=g, device=device) * sigma).contiguous(), 0
        gen = pkg.BurstGenerator()
        burst = (64 + 128 + 1504 * 4 + gen.RAMP_DOWN + gen.FLUSH) * SPS
        # (iv) back to back with the 500-symbol gaps of the headline's stream: the other end of the density scale
        period = burst + 500 * SPS if kind == "dense_packets" else 1 << 20
        n_pkt = n // period
        rng = np.random.default_rng(5)
        payloads = [rng.integers(0, 256, 1500, dtype=np.uint8).tobytes() for _ in range(n_pkt)]
        x = gen.stream(payloads, np.full(n_pkt, period - burst), freq_error=0.01, esn0_db=20.0, seed=6, tail=0,
                       carrier="closed_form")
        if x.numel() < n:  # (the stream ends with its last packet: noise-free padding up to the window)
            x = torch.cat([x, torch.zeros(n - x.numel(), dtype=x.dtype, device=x.device)])
        return x[:n].contiguous(), n_pkt

    for kind in streams:
        x, n_pkt = make(kind)
        ring = torch.empty(hist + 1 + n, dtype=torch.complex64, device=device)
        ring[1:1 + hist] = x[-hist:]
        ring[1 + hist:] = x
        w, history = ring[1 + hist:], ring[1:1 + hist]
        del x
        # round 6: packets_only -- what the reference's benchmark measures is packets out; the streams between the Costas loop
        # and the packer are not written to memory (include/gr4pm_hip.h; same packets as the full form, tests)
        rx = pkg.NativePacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=n, tags_cap=max(4096, 4 * n_pkt), pipelined=True,
                                      decode_headers=True, output_ring=True, packets_only=True, result_fields="packets")
        stats = {"consumed": 0, "tags": 0, "packets_crc_ok": 0}

        def note(r):
            if r is not None:
                stats["consumed"] += r["consumed"]
                stats["tags"] += int(r["tags"].size)
                stats["packets_crc_ok"] += int(np.sum(r["packet_lengths"] > 0))

        def run(k):
            for i in range(k):
                if i + 1 < k:
                    rx.announce(w)
                note(rx.process_bulk(w, None, history=history))
            for r in rx.flush():
                note(r)

        def timed(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(k)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        run(2)
        # The receiver is a pipeline of six stages (detector | pass A | gate + plan | symbol filter | PayloadMetadataInsert +
        # PLL | header loop + payload tail): a run of k passes costs k steady-state periods + one fill / drain of the
        # pipeline (~ 3 periods on packets back to back).  Two run lengths separate the two: `value` is the longer run
        # as a whole (drain included), `steady_state_ms_per_2^28` the difference quotient.
        short = max(4, passes // 3)
        dt_short = timed(short)
        stats = {"consumed": 0, "tags": 0, "packets_crc_ok": 0}
        dt = timed(passes)
        steady = (dt - dt_short) / (passes - short)
        del rx
        sd = pkg.SyncwordDetection(rrc, SYNCWORD, bpsk, -BINS, BINS, power_threshold=9.5, max_items=n)
        for _ in range(2):
            sd.process_bulk(w, want_output=False, tags_cap=max(4096, 4 * n_pkt))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        done = 0
        for _ in range(passes):
            done += sd.process_bulk(w, want_output=False, tags_cap=max(4096, 4 * n_pkt))[3]
        torch.cuda.synchronize()
        dt_sd = (time.perf_counter() - t1) * stats["consumed"] / max(done, 1)
        del sd, ring, w, history
        torch.cuda.empty_cache()
        if n_pkt == 0 and (stats["tags"] or stats["packets_crc_ok"]):
            raise SystemExit(f"bench.py: sparse leg {kind}: {stats['tags']} tags on a stream without a packet")
        # (Es/N0 = 20 dB, uncoded payload: a CRC fails now and then; the packet across each pass's seam is lost)
        if n_pkt and stats["packets_crc_ok"] < passes * (n_pkt - 2) * (0.99 if kind == "dense_packets" else 1.0):
            raise SystemExit(f"bench.py: sparse leg {kind}: {stats['packets_crc_ok']} of {passes * n_pkt} packets came back")
        rows[kind] = {"value": round(stats["consumed"] / dt / 1e6, 2), "unit": "Msamples/s",
                      "ms_per_2^28": round(dt / passes * 1e3, 3), "passes": passes,
                      "steady_state_ms_per_2^28": round(steady * 1e3, 3),
                      "fill_and_drain_ms": round((dt - passes * steady) * 1e3, 2), "tags_per_2^28": stats["tags"] // passes,
                      "packets_crc_ok_per_2^28": stats["packets_crc_ok"] // passes,
                      "detector_alone": round(stats["consumed"] / dt_sd / 1e6, 2),
                      "behind_the_detector_share": round(max(0.0, 1.0 - dt_sd / dt), 3)}
    return {"workload": "whole receiver (decode_headers, packets_only form: IQ in, CRC-checked packets out), 2^28 resident samples per pass, "
                        "nine templates: zeros (the reference's benchmark_packet_receiver input) / AWGN only / one "
                        "1500-byte packet per 2^20 samples / 1500-byte packets back to back (500-symbol gaps)",
            "reference_benchmark_packet_receiver_msps_ryzen_5800x": "6-8 (nine templates) ... 28-32 (one)",
            "streams": rows,
            "note": "not `value`: the headline is the packet-dense stream of configs[1]"}


def sparse_leg_in_a_fresh_process():
    """The `sparse` sub-record from a child process (`bench.py --sparse-leg-only`, started the way cpu_baseline starts its
    workers: a child, never an exec of this process).  Round 6: in THIS process -- minutes of allocating and freeing
    multi-GiB device buffers behind it -- the packet-dense stream ran at 45 - 52 Gsps depending on which legs had run
    before it (47 after the host-stream leg, 57 right behind config 5's allocate-and-free of 8 GiB), in a process of its
    own at 56 - 57 (tools/r6_leg_ab.sh).  The cause is NOT established: allocating and freeing GiB-sized blocks out of order in
    a fresh process did not reproduce it (tools/r6_dense_kstats.py, R6_CHURN: 55.9 - 56.6 Gsps with and without).  What a
    receiver in production is -- a process that allocates its buffers once -- is what the child process measures; the
    parent keeps its context and releases its cached blocks first."""
    import gc
    import subprocess
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    def child(extra_args, extra_env):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--sparse-leg-only"] + extra_args, capture_output=True,
                           text=True, env={**env, **extra_env})
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise SystemExit(f"bench.py: the sparse leg's child process failed ({r.returncode}):\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}")
        return json.loads(lines[-1])
    rec = child([], {})
    rec["measured_in"] = "a child process of its own (bench.py --sparse-leg-only)"
    # Round 6: one packet per 2^20 samples again, in a process whose HIP runtime has 32 hardware queues instead of the
    # default four (GPU_MAX_HW_QUEUES, read when the runtime starts).  There the library runs the phasor chains of
    # consecutive batches SIDE BY SIDE (each is 2^20 dependent steps, 16.7 ms; csrc/stream_blocks.hip: PlanSync) instead of
    # one batch's behind the other's -- with four queues a chain kernel holds back whatever shares its queue, and the
    # library keeps the one-kernel form.  A process-wide setting with a price elsewhere (the front end alone: -1 %, 64
    # channels: +3 %), so a receiver process chooses it by its traffic; the rows above are the runtime's default.
    key = "one_packet_per_2^20"
    if key in rec["streams"]:
        wide = child(["--sparse-streams", key], {"GPU_MAX_HW_QUEUES": "32"})["streams"][key]
        wide["process_setting"] = "GPU_MAX_HW_QUEUES=32 (the phasor chains of consecutive batches side by side)"
        rec["streams"][key + ", 32 hardware queues"] = wide
    return rec


def headline_ring(x, xb):
    """the headline's device ring [.. | window A | window B]: two different stretches of the burst stream that the steps
    present alternately, each preceded in memory by the 2T+1 items "before" it (for A: a copy of B's tail, for B: A's
    tail itself) -> (ring, [(window, its history)] x 2).  tests/test_gpu_parity.py builds the same ring."""
    hist, n_items = 2 * 768 + 1, x.numel()
    ring = torch.empty(hist + 1 + 2 * n_items, dtype=torch.complex64, device=x.device)  # +1: keep A 16-byte aligned
    ring[1:1 + hist] = xb[-hist:]
    ring[1 + hist:1 + hist + n_items] = x
    ring[1 + hist + n_items:] = xb
    return ring, [(ring[1 + hist:1 + hist + n_items], ring[1:1 + hist]),
                  (ring[1 + hist + n_items:], ring[1 + n_items:1 + hist + n_items])]


# samples per packet period of burst_stream(): 64 + 128 + 1504 * 4 symbols + a gap of 500, 4 samples per symbol
BURST_PERIOD = (64 + 128 + 1504 * 4 + 500) * SPS


def check_tag_count(what, n_tags, n_windows, items_per_window, period=BURST_PERIOD, raw=False):
    """outside every timed region: the detector found the packets the generator put in -- one tag per packet period of
    every window, give or take the packet cut by each window edge.  raw: the detector's own tags without the
    SyncwordDetectionFilter behind it (--detector-only): every packet, and up to 2 % more (a packet's second detection
    is what that filter drops, syncword_detection_filter.hpp:50-95)"""
    want = n_windows * (items_per_window / period)
    slack = 2 * n_windows + 1
    if (n_tags < want - slack or n_tags > want * 1.02 + slack) if raw else abs(n_tags - want) > slack:
        raise SystemExit(f"bench.py: {what}: {n_tags} tags over {n_windows} windows of {items_per_window} items, "
                         f"the generator's packet count is {want:.1f}: the measured path does not do the receiver's work")


def correlator_flops_per_sample(n_bins, n_fft=N_FFT, stride=1752):
    """[(1 + B) 5 N log2 N + 6 B N + 1.5 N + 4 B S] / S"""
    return ((1 + n_bins) * 5.0 * n_fft * np.log2(n_fft) + 6.0 * n_bins * n_fft + 1.5 * n_fft + 4.0 * n_bins * stride) / stride


ROOF_WARM = 30  # untimed launches in front of a kernel-alone timing (see the roofline leg in main())
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured (float4 copy)
# What the correlator's own ACCESS PATTERN reaches with no transform at all (tools/overlap_save_pattern.hip, round 6:
# every block reads 2048 samples of which the next re-reads 296, writes a 4-byte power per lag; one wave per block, the
# library's launch shape): 5.24 TB/s of the 12 B/sample that reach HBM -- 8-byte and 16-byte loads alike (5.22), 5.0 with
# round 5's launch shape (profiles/r6_overlap_save_pattern.txt).  The ceiling of the traffic side is priced with THIS.
OVERLAP_SAVE_PATTERN_GBS = 5240.0


def roofline_ceiling(n_bins, n_fft=N_FFT, stride=1752):
    """What `frac` (8 algorithmic bytes per sample over the 8 TB/s HBM peak) can be at most for an exact-FP32 FFT
    correlator with n_bins templates on this chip: the arithmetic alone at the FP32 vector peak (SURVEY.md 8(d)'s flop
    count), the real traffic alone (8 B read + 4 B zpow write per sample) at the rate the overlap-save access pattern
    reaches with no arithmetic (measured, OVERLAP_SAVE_PATTERN_GBS; round 5 assumed the 6.3 TB/s of a float4 copy);
    the smaller one is the ceiling.  north_star's >= 0.6 is above it at every bin count."""
    # (nine bins at N = 2048: SURVEY.md 8(d) quotes 710 flop/sample, the figure `fp32_frac` of the headline is computed
    # from; its own formula gives 744 -- the quoted figure is used where it exists, so that frac / ceiling = fp32_frac)
    fl = 710.0 if (n_bins, n_fft, stride) == (9, N_FFT, 1752) else float(correlator_flops_per_sample(n_bins, n_fft, stride))
    fp32_bound = float(FP32_PEAK_TFLOPS * 1e12 / fl * 8.0 / (HBM_PEAK_GBS * 1e9))
    traffic_bound = OVERLAP_SAVE_PATTERN_GBS / 12.0 * 8.0 / HBM_PEAK_GBS
    return {"fp32_bound_frac": round(fp32_bound, 4), "traffic_bound_frac": round(traffic_bound, 4),
            "frac": round(min(fp32_bound, traffic_bound), 4),
            "assumes": f"{fl:.0f} flop/sample at {FP32_PEAK_TFLOPS} TFLOP/s FP32 vector; 12 B/sample of real traffic at the "
                       f"{OVERLAP_SAVE_PATTERN_GBS / 1e3:.2f} TB/s the overlap-save access pattern reaches by itself "
                       f"(measured: profiles/r6_overlap_save_pattern.txt)"}


def per_bins_roofline(pkg, rrc, bpsk, x, n_items, stream, reps=5):
    """the correlator alone for 1 / 3 / 5 / 7 / 9 templates (the rows the reference publishes,
    benchmarks/results.md:37-41): mean launch time (HIP events on the launch stream), fraction of the HBM roofline
    at 8 B/sample read, fraction of the FP32 vector peak at SURVEY.md 8(d)'s flop count"""
    out = {}
    samples = ((n_items - N_FFT) // 1752 + 1) * 1752
    with torch.cuda.stream(stream):
        for b in range(0, BINS + 1):
            sd = pkg.SyncwordDetection(rrc, SYNCWORD, bpsk, -b, b, power_threshold=9.5, max_items=n_items)
            for _ in range(ROOF_WARM // 2):
                sd.correlate_only(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                sd.correlate_only(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = correlator_flops_per_sample(2 * b + 1) * samples / (ms * 1e-3) / 1e12
            out[str(2 * b + 1)] = {"launch_ms": round(ms, 4), "gsps": round(samples / ms / 1e6, 2),
                                   "frac": round(8.0 * samples / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                   "fp32_tflops": round(fl, 2), "fp32_frac": round(fl / FP32_PEAK_TFLOPS, 4),
                                   "ceiling": roofline_ceiling(2 * b + 1)["frac"]}
            del sd
    return out


def config2_latency(pkg, xs, channels, n_pkt, sizes=(1 << 12, 1 << 13, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22), rate_sps=3.2e6):
    """the real-time operating point of configs[2] (64 channels of 3.2 Msps each, README.md:46-49): for batches of b
    samples per channel -- one arrives every b / 3.2e6 s -- the submit -> collect latency of ONE batch with nothing else
    in flight (median of 9), and the sustained rate with four batches in flight; `sustains` = the receiver keeps up with
    64 x 3.2 Msps at that batch size (rate above 204.8 Msps and latency below the batch period)."""
    rows = {}
    for b in sizes:
        x = xs[:, :b]
        multi = pkg.NativeMultiChannelReceiver(channels, SPS, BINS, 9.5, "QPSK", max_items=b, tags_cap=max(64, 2 * n_pkt + 64),
                                               workers=12, output_ring=True)
        multi.set_input_in_place(True)
        for _ in range(3):
            multi.submit(x, 1500)
            multi.collect()
        lat = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            multi.submit(x, 1500)
            multi.collect()
            lat.append(time.perf_counter() - t0)
        steps = max(8, min(200, (1 << 24) // b))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        consumed = 0
        for _ in range(steps):
            if multi.in_flight() == 4:
                consumed += sum(r["consumed"] for r in multi.collect())
            multi.announce(x)
            multi.submit(x, 1500)
        while multi.in_flight():
            consumed += sum(r["consumed"] for r in multi.collect())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        period = b / rate_sps
        lat_ms = sorted(lat)[len(lat) // 2] * 1e3
        rows[str(b)] = {"batch_period_ms": round(period * 1e3, 3), "latency_ms": round(lat_ms, 3),
                        "latency_min_ms": round(min(lat) * 1e3, 3), "msps": round(consumed / dt / 1e6, 1),
                        "ms_per_batch_pipelined": round(dt / steps * 1e3, 3),
                        "sustains": bool(consumed / dt > channels * rate_sps and lat_ms < period * 1e3)}
        del multi
    ok = [int(k) for k, v in rows.items() if v["sustains"]]
    return {"channel_rate_sps": rate_sps, "channels": channels, "batches": rows,
            "smallest_batch_that_sustains_realtime": min(ok) if ok else None}


def channels64_leg(pkg, dist, device, rank, world, rrc, steps, warmup, repeats, channels=64, n_items=1 << 22, latency_block=True):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1: 64 channels per GPU, 64 N in all): rank 0 fills a host
    sample ring [N, 64, 2^22] and scatters one [64, 2^22] slab per rank (the job's backend: RCCL on the GPUs); every
    rank then runs its 64 channels through gr4pm_multichannel_receiver (one batched detector + every channel's own
    chain, submit / collect with four batches in flight, input read in place).  Timed like the headline region."""
    n_pkt = (n_items // SPS + 64) // (64 + 128 + 1504 * 4 + 500) + 1  # packets per channel and batch (burst_stream)
    scatter_rec = None
    if dist:
        scatter_budget(dist, rank, world, 8 * n_items * channels, device, host_ring=True)

        def make_all():
            host, _ = host_sample_ring(world, (channels, n_items))
            for r in range(world):
                host[r].copy_(channel_bank_config3(pkg, n_items, rrc, channels, device, seed0=1000 * (r + 1)))
            return host.to(device, non_blocking=False)
        xs = scatter_channels(dist, make_all, (channels, n_items), device, rank, world)
        scatter_rec = dict(LAST_SCATTER)
        input_mode = f"rank 0 host sample ring [{world}, {channels}, {n_items}] -> all ranks, scatter ({dist.get_backend()})"
    else:
        xs = channel_bank_config3(pkg, n_items, rrc, channels, device, seed0=1000 * (rank + 1))
        input_mode = "generated on the GPU"
    multi = pkg.NativeMultiChannelReceiver(channels, SPS, BINS, 9.5, "QPSK", max_items=n_items,
                                           tags_cap=max(64, 2 * n_pkt + 64), workers=12, output_ring=True)
    multi.set_input_in_place(True)
    state = {"step": 0, "announced": 0, "tags": 0}

    def step(left):
        target = state["step"] + min(left, 2)  # look-ahead two batches ahead, as the headline
        state["announced"] = max(state["announced"], state["step"])
        while state["announced"] < target:
            state["announced"] += 1
            multi.announce(xs)
        state["step"] += 1
        res = multi.collect() if multi.in_flight() == 4 else None
        multi.submit(xs, 1500)
        if res is not None:
            state["tags"] += sum(r["tags"].size for r in res)
        return 0 if res is None else sum(r["consumed"] for r in res)

    def drain():
        n = 0
        while multi.in_flight():
            res = multi.collect()
            n += sum(r["consumed"] for r in res)
            state["tags"] += sum(r["tags"].size for r in res)
        return n

    for i in range(warmup):
        step(warmup - 1 - i)
    drain()

    def region():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        consumed = 0
        state["tags"] = 0
        for i in range(steps):
            consumed += step(steps - 1 - i)
        consumed += drain()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, consumed

    # Round 6: `warmup` steps do not reach steady clocks for this leg (round 5's first timed region was 12 % slower than
    # the other two, VERDICT): untimed regions until two consecutive ones agree within 2 % (at most six; every rank
    # sees the same gathered times, so every rank stops after the same region)
    warm_regions, prev = 0, None
    while warm_regions < 6:
        own, _ = region()
        cur = max(per_rank(dist, own, device))
        warm_regions += 1
        if prev is not None and abs(cur - prev) <= 0.02 * prev:
            break
        prev = cur
    rates, times, rank_ms = [], [], []
    for _ in range(max(1, repeats)):
        own, consumed = region()
        rank_ms.append([round(v / steps * 1e3, 4) for v in per_rank(dist, own, device)])
        dt, total = aggregate(dist, own, float(consumed), device)
        rates.append(total / dt / 1e6)
        times.append(dt)
        tags_per_step = state["tags"] // steps
        check_tag_count(f"{channels}-channel leg", state["tags"], steps * channels, n_items)
    med = sorted(range(len(rates)), key=lambda i: rates[i])[len(rates) // 2]
    del multi
    latency = None
    if dist is None and latency_block:
        latency = config2_latency(pkg, xs, channels, n_pkt)
    return {"workload": f"{channels} channels per GPU x {n_items} samples per batch, {channels * world} channels in all, "
                        "full RX front end per channel (gr4pm_multichannel_receiver), per-channel CFO sweep",
            "value": round(rates[med], 2), "unit": "Msamples/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(times[med] / steps * 1e3, 4), "repeats": len(rates), "warm_regions": warm_regions,
            "tags_per_step": tags_per_step,
            "value_min": round(min(rates), 2), "value_max": round(max(rates), 2), "input": input_mode,
            "ms_per_step_per_rank": rank_ms[med],
            **({"scatter": scatter_rec} if scatter_rec else {}),
            **({"latency": latency} if latency else {})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=8)  # covers the first use of every buffer ring
    ap.add_argument("--items", type=int, default=None,
                    help="samples per step per GPU (per channel with --channels).  Default: 2^28 (SURVEY.md 8(d) config 2: "
                         "2^28 samples resident in HBM per pass), 2^22 per channel with --channels (config 3), 2^26 with "
                         "--config 5")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps between two barriers) is run this many times back to "
                         "back; `value` is the median region, min / max beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-channels-leg", action="store_true",
                    help="skip the 64-channels-per-GPU sub-record (configs[2] at N = 1, configs[3] at N > 1) that follows "
                         "the headline region")
    ap.add_argument("--detector-only", action="store_true", help="time SyncwordDetection alone")
    ap.add_argument("--channels", type=int, default=1,
                    help="config 3: C independent channels on one GPU: one batched SyncwordDetection handle and "
                         "every channel's own chain behind it (with --detector-only: the detector alone); "
                         "--items is per channel")
    ap.add_argument("--channel-workers", type=int, default=12,
                    help="--channels: host threads (each with its own stream) that drive the per-channel chains")
    ap.add_argument("--copy-delay", action="store_true",
                    help="materialise SyncwordDetection's delayed output instead of reading the ring in place")
    ap.add_argument("--no-scatter", action="store_true",
                    help="N > 1, --channels: let every rank generate its own channels instead of the scatter from rank 0")
    ap.add_argument("--scatter-headline", action="store_true",
                    help="N > 1: rank 0 generates every rank's headline channel and scatters them (rounds 1-4); by "
                         "default every rank generates its own and only the configs[3] host sample ring is scattered")
    ap.add_argument("--lookahead-depth", type=int, default=None,
                    help="how many calls ahead the detector's front part (correlator, candidates, tables) is launched "
                         "(max 2; default 2, and 1 with --channels: the multi-channel receiver is synchronous, a second "
                         "front in flight only slows the phases of the current call)")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="do not announce the next window to SyncwordDetection (no correlator look-ahead)")
    ap.add_argument("--soft-bits", action="store_true",
                    help="continue the chain to LLRs: PayloadMetadataInsert -> tag-driven CostasLoop -> "
                         "SyncwordRemove -> ConstellationLLRDecoder (SURVEY.md 8(f) rank 1; not the headline workload)")
    ap.add_argument("--decode-headers", action="store_true",
                    help="--soft-bits plus the header decode loop on the device (descrambler, header/payload split, "
                         "LDPC header decoder, parser) instead of a given packet length (SURVEY.md 8(f) rank 2)")
    ap.add_argument("--python-pipeline", action="store_true",
                    help="drive the three stages from Python threads (blocks.py PacketReceiver) instead of the "
                         "native composition gr4pm_packet_receiver (identical results)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the two halves of the chain back to back on one stream")
    ap.add_argument("--config", type=int, default=2, choices=[2, 5],
                    help="2 (default): BASELINE configs[1], the headline workload; 5: BASELINE configs[4], the stress "
                         "shape (N = 4096 overlap-save blocks, 1025-tap RRC, SyncwordDetection + the 1025-tap filter leg)")
    ap.add_argument("--no-config5-leg", action="store_true",
                    help="leave out the configs[4] sub-record (2^30 samples, fft_size 4096, 1025-tap RRC) of the default line")
    ap.add_argument("--no-sparse-leg", action="store_true",
                    help="skip the packet-density sub-record (whole receiver on zeros / AWGN only / one packet per 2^20 samples)")
    ap.add_argument("--no-host-stream-leg", action="store_true",
                    help="skip the PCIe-inclusive sub-record (front end fed from pinned host memory)")
    ap.add_argument("--no-per-bins", action="store_true",
                    help="leave out the roofline.per_bins legs (profiling runs: the correlator's rocprof average is then "
                         "the nine-bin launch alone)")
    ap.add_argument("--config3-items", type=int, default=1 << 22,
                    help="samples per channel and batch of the 64-channels-per-GPU leg (configs[2] / configs[3]: 2^22; tests "
                         "cut the ring down in items, never in shape)")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes over the correlator alone, "
                         "about 20 s, before anything else): the committed PMC file is quoted instead")
    ap.add_argument("--sparse-leg-only", action="store_true",
                    help="(what the default run starts as a child process) the packet-density sub-record alone, printed as JSON")
    ap.add_argument("--sparse-streams", default=",".join(SPARSE_STREAMS),
                    help="with --sparse-leg-only: which of its streams (comma-separated)")
    ap.add_argument("--selfcheck", action="store_true",
                    help="N-rank first-contact check only (rendezvous, identities, a small scatter, two batches): what "
                         "`bench.py --gpus N` runs by itself before the job")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / aggregation check without a GPU: every rank joins a gloo group, aggregates "
                         "fixed numbers and rank 0 prints the line (tests/test_distributed_cpu.py)")
    args = ap.parse_args()
    if args.items is None:
        args.items = 1 << 26 if args.config == 5 else (1 << 22 if args.channels > 1 else 1 << 28)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes,
        # BEFORE this process makes any GPU call (a process that has touched the GPU is never re-executed)
        sys.exit(launch_ranks(args.gpus))
    if int(world_env or 1) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env or 1}: refusing to print a line "
                         f"for a different GPU count")
    if args.dry_run:
        return dry_run(args)
    if args.sparse_leg_only:
        pkg = ge.load_package()
        torch.cuda.set_device(0)
        sys.setswitchinterval(float(os.environ.get("GR4PM_SWITCH_INTERVAL", "5e-5")))
        print(json.dumps(sparse_leg(pkg, torch.device("cuda", 0), unit_norm_rrc(pkg),
                                    streams=tuple(k for k in args.sparse_streams.split(",") if k in SPARSE_STREAMS))))
        return None
    if args.config == 5:
        return config5(args)
    if args.lookahead_depth is None:
        args.lookahead_depth = 2
    # roofline.traffic, measured in this run: child processes under rocprofv3, BEFORE this process makes its first GPU call
    # (N = 1, the default headline workload only; profiling runs of bench.py itself pass --no-pmc-traffic or --no-per-bins)
    global MEASURED_TRAFFIC
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.channels == 1 and not args.no_pmc_traffic and not args.no_per_bins
            and not args.no_cpu_baseline and not (args.soft_bits or args.decode_headers or args.detector_only or args.no_pipeline)):
        MEASURED_TRAFFIC = measure_traffic_in_run(args.items)
    # three host threads drive the three pipeline stages and spend most of their time inside the
    # C library (GIL released); when one comes back it should not wait 5 ms (the default switch
    # interval) for whichever thread is running Python glue at that moment
    sys.setswitchinterval(float(os.environ.get("GR4PM_SWITCH_INTERVAL", "5e-5")))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # GR4PM_BENCH_SHARED_DEVICE_TEST=1 (tests only; the line says so and is not a measurement): every rank takes
    # cuda:0 and the group runs on gloo, so that the N > 1 code of this file -- self-check, scatter, barriers and
    # aggregation around the timed regions, the configs[3] leg -- can run on a box with ONE GPU.
    shared = SHARED_DEVICE_TEST and world > 1
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist = init_ranks("gloo" if shared else "nccl", local_rank)
        # N distinct devices, checked right after the rendezvous and on every rank -- not after the timed regions
        check_distinct_devices(rank_identities(dist, device, world), world)
    pkg = ge.load_package()

    rrc = unit_norm_rrc(pkg)
    if args.selfcheck or world > 1:
        sc = selfcheck(pkg, dist, device, rank, world, rrc)
        if args.selfcheck:
            if rank == 0:
                print(json.dumps({"selfcheck": "ok", **sc}))
            if dist:
                dist.barrier()
                dist.destroy_process_group()
            return
    bpsk = np.array([1, -1], dtype=np.complex64)
    n_items = args.items
    input_mode = "generated on each GPU"
    hdr_syms = None

    def make_stream(seed):
        if args.decode_headers:
            return packet_stream(pkg, n_items, seed, device)
        return burst_stream(pkg, n_items, rrc, seed=seed, device=device, header=hdr_syms)

    x, n_pkt = make_stream(1 + rank)
    xs_bank = None
    if dist and not args.no_scatter:
        # multi-channel receive: rank 0 owns the sample ring of all channels and scatters it.  No fallback: a rank
        # that fails here ends with an error, and the launcher (torch.distributed.run or launch_ranks) ends the job.
        scatter_budget(dist, rank, world, 8 * n_items * max(args.channels, 1), device, host_ring=args.channels > 1)
        if args.channels > 1:
            # configs[3]: `channels` per GPU; rank 0 fills its host sample ring [world, C, n] (D2H), uploads
            # it and scatters one [C, n] slab per rank: the workload's only collective
            def make_all():
                host, _ = host_sample_ring(world, (args.channels, n_items))
                for r in range(world):
                    host[r].copy_(channel_bank_config3(pkg, n_items, rrc, args.channels, device, seed0=1000 * (r + 1)))
                return host.to(device, non_blocking=False)
            xs_bank = scatter_channels(dist, make_all, (args.channels, n_items), device, rank, world)
            input_mode = (f"rank 0 host sample ring [{world}, {args.channels}, {n_items}] -> all ranks, "
                          f"torch.distributed scatter ({dist.get_backend()})")
        elif args.scatter_headline:
            def make_all():
                chans = [x] + [make_stream(1 + r)[0] for r in range(1, world)]
                return torch.stack(chans)
            x = scatter_channels(dist, make_all, n_items, device, rank, world)
            input_mode = f"rank 0 -> all ranks, torch.distributed scatter ({dist.get_backend()})"
        else:
            # the headline's one channel per GPU is generated where it is used, seeded by the rank (rank 0 building
            # N x 2 GiB one after the other while N - 1 ranks wait was serial start-up for nothing); the scatter of
            # SURVEY 8(e) -- rank 0's HOST sample ring to every GPU -- is the configs[3] leg's (channels64_leg)
            input_mode = "generated on each GPU (seed = 1 + rank); the host sample ring's scatter is the config3 leg's"
    # the stream lives in a device ring [.. | window A | window B]: two different stretches of the
    # burst stream that the steps present alternately, each preceded in memory by the 2T+1 items
    # "before" it (for A: a copy of B's tail, for B: A's tail itself).  While one window is being
    # processed the detector is told which one comes next (look-ahead of the correlator).
    xb, n_pkt_b = make_stream(1001 + rank)
    ring, windows = headline_ring(x, xb)
    del xb
    x = windows[0][0]
    n_pkt = max(n_pkt, n_pkt_b)
    native = not (args.python_pipeline or args.detector_only or args.channels > 1)
    if native:
        rx = pkg.NativePacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=n_items, tags_cap=max(64, 2 * n_pkt + 64),
                                      pipelined=not args.no_pipeline, soft_bits=args.soft_bits,
                                      decode_headers=args.decode_headers, output_ring=True)
        sd = None
    else:
        rx = pkg.PacketReceiver(SPS, BINS, 9.5, "QPSK", max_items=n_items, pipelined=not args.no_pipeline,
                                soft_bits=args.soft_bits, decode_headers=args.decode_headers)
        sd = rx.syncword_detection
    out_keep = None
    multi = None
    if args.channels > 1:
        # config 3: every channel its own burst stream (seed) with a carrier offset of -0.04 .. +0.04 rad/sample
        full_chain = not args.detector_only
        args.detector_only = True
        C = args.channels
        xs = xs_bank if xs_bank is not None else channel_bank_config3(pkg, n_items, rrc, C, device, seed0=1000 * (rank + 1))
        x = xs
        windows = [(xs, None), (xs, None)]
        if full_chain:
            # every channel's own chain behind one batched detector (blocks.py MultiChannelPacketReceiver)
            if args.python_pipeline:  # the same composition from Python threads (host-bound)
                multi = pkg.MultiChannelPacketReceiver(C, SPS, BINS, 9.5, "QPSK", max_items=n_items,
                                                       workers=args.channel_workers)
                sd = multi.syncword_detection
            else:
                multi = pkg.NativeMultiChannelReceiver(C, SPS, BINS, 9.5, "QPSK", max_items=n_items,
                                                       tags_cap=max(64, 2 * n_pkt + 64), workers=args.channel_workers,
                                                       output_ring=True)
                if not args.copy_delay:  # the ring's windows stay valid and unchanged: no delayed copy per batch
                    multi.set_input_in_place(True)
                sd = multi  # announce() goes to the library's detector
        else:
            with torch.cuda.stream(rx._streams[0]):
                sd = pkg.SyncwordDetection(rrc, SYNCWORD, bpsk, -BINS, BINS, power_threshold=9.5, n_channels=C,
                                           max_items=n_items)

    step_no = 0
    announced_upto = 0
    hdr_stats = {"decoded": 0, "valid_1500": 0, "mismatches": 0, "packets_crc_ok": 0, "packets_crc_failed": 0}

    def note_headers(res):
        if args.decode_headers and "header_messages" in res:
            m = res["header_messages"]
            hdr_stats["decoded"] += int(m.size)
            hdr_stats["valid_1500"] += int(np.sum((m["invalid_header"] == 0) & (m["packet_length"] == 1500)))
            hdr_stats["mismatches"] += int(res["header_mismatches"])
            hdr_stats["packets_crc_ok"] += int(np.sum(res["packet_lengths"] > 0))
            hdr_stats["packets_crc_failed"] += int(np.sum(res["packet_lengths"] == 0))
            if os.environ.get("GR4PM_BENCH_DEBUG"):
                bad = np.nonzero(~((m["invalid_header"] == 0) & (m["packet_length"] == 1500)))[0]
                print("batch headers", m.size, "bad", bad.size, bad[:6], bad[-3:], m[bad[:3]], file=sys.stderr)

    def step(left=0):
        """one pass over one window; `left`: how many further steps follow in this region (the
        look-ahead never reaches beyond it, so that exactly `steps` correlator launches fall
        inside the timed region)"""
        nonlocal out_keep, step_no, announced_upto
        w, history = windows[step_no % 2]
        w_next = None
        if not args.no_lookahead:
            # announce the inputs of the next calls, at most `lookahead_depth` ahead
            target = step_no + min(left, args.lookahead_depth)
            announced_upto = max(announced_upto, step_no)
            while announced_upto < target:
                announced_upto += 1
                (sd if args.detector_only else rx).announce(windows[announced_upto % 2][0])
        step_no += 1
        if multi is not None:
            if args.python_pipeline:
                res = multi.process_bulk(w, 1500, tags_cap=max(64, 2 * n_pkt + 64))
            elif args.no_pipeline:
                res = multi.process_bulk(w, 1500)
            else:  # native pipeline: up to four batches in flight, results in submission order
                res = multi.collect() if multi.in_flight() == 4 else None
                multi.submit(w, 1500)
                if res is None:
                    return 0, 0
            out_keep = res[-1]["symbols"]
            return sum(r["consumed"] for r in res), sum(r["tags"].size for r in res)
        if args.detector_only:
            with torch.cuda.stream(rx._streams[0]):
                st, out, tags, n = sd.process_bulk(w, want_output=True, tags_cap=max(64, 2 * n_pkt + 64),
                                                   next_x=w_next)
            out_keep = out
            if args.channels > 1:
                return n * args.channels, sum(t.size for t in tags)
            return n, tags.size
        res = rx.process_bulk(w, 1500, tags_cap=max(64, 2 * n_pkt + 64),  # payload length of the generator
                              history=None if args.copy_delay else history, next_x=w_next)
        if res is None:  # pipelined: first call has no finished batch yet
            return 0, 0
        out_keep = res["symbols"]
        note_headers(res)
        return res["consumed"], res["tags"].size

    def drain():
        nonlocal out_keep
        if multi is not None and not args.python_pipeline and not args.no_pipeline:
            n = nt = 0
            while multi.in_flight():
                res = multi.collect()
                out_keep = res[-1]["symbols"]
                n += sum(r["consumed"] for r in res)
                nt += sum(r["tags"].size for r in res)
            return n, nt
        if args.detector_only:
            return 0, 0
        n = nt = 0
        for res in rx.flush():
            out_keep = res["symbols"]
            note_headers(res)
            n += res["consumed"]
            nt += res["tags"].size
        return n, nt

    for i in range(args.warmup):
        step(left=args.warmup - 1 - i)
    drain()

    def timed_region():
        """exactly --steps steps between two (barrier + synchronize) pairs; time = MAX over ranks, items = SUM"""
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        consumed = n_tags = 0
        for i in range(args.steps):
            n, nt = step(left=args.steps - 1 - i)
            consumed += n
            n_tags += nt
        n, nt = drain()  # pipelined: the last batch finishes inside the timed region
        consumed += n
        n_tags += nt
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ranks_ms.append([round(v / args.steps * 1e3, 4) for v in per_rank(dist, dt, device)])
        dt, total = aggregate(dist, dt, float(consumed), device)
        return dt, total, n_tags

    ranks_ms = []  # per region: every rank's own ms per step
    regions = [timed_region() for _ in range(max(1, args.repeats))]
    if not args.decode_headers:  # (packet_stream has its own period; its leg counts CRC-checked packets instead)
        for _, _, nt in regions:
            check_tag_count("headline region", nt, args.steps * max(args.channels, 1), n_items,
                            raw=args.detector_only and multi is None)
    by_rate = sorted(regions, key=lambda r: r[1] / r[0])
    dt, total, n_tags = by_rate[len(by_rate) // 2]  # the median region is the number of record
    median_region = regions.index(by_rate[len(by_rate) // 2])
    region_rates = [round(r[1] / r[0] / 1e6, 2) for r in regions]

    # ---- 64 channels per GPU (configs[2] at N = 1, configs[3] = 64 x N channels at N > 1), after the headline
    # region, in the same run; only for the default headline workload
    channels_leg = None
    headline = native and not (args.soft_bits or args.decode_headers or args.no_pipeline or args.no_lookahead or args.copy_delay)
    if headline and not args.no_channels_leg:
        rx = None  # the headline receiver is done: its stage threads and streams go before the next leg starts
        channels_leg = channels64_leg(pkg, dist, device, rank, world, rrc, steps=max(args.steps, 50), warmup=6,
                                      repeats=min(args.repeats, 3), n_items=args.config3_items)
    # ---- BASELINE configs[4] (the 2-Gsps stress shape) as a sub-record of the default line, N = 1 only
    config5_rec = None
    if headline and not args.no_config5_leg and world == 1:
        multi = None
        torch.cuda.empty_cache()
        config5_rec = config5_leg(pkg, device)
        torch.cuda.empty_cache()
    # ---- PCIe-inclusive: the same front end fed from pinned host memory (N = 1 only; never `value`)
    host_stream_rec = None
    if headline and not args.no_host_stream_leg and world == 1:
        host_stream_rec = host_stream_leg(pkg, device, rrc)
        torch.cuda.empty_cache()
    # ---- packet density: the whole receiver on zeros / AWGN / one packet per 2^20 samples (N = 1 only; never `value`)
    sparse_rec = None
    if headline and not args.no_sparse_leg and world == 1:
        sparse_rec = sparse_leg_in_a_fresh_process()
        torch.cuda.empty_cache()
    job = rank_identities(dist, device, world)
    check_distinct_devices(job, world)

    # ---- roofline of the dominant kernel: HIP events on the stream the kernel runs on
    roofline = None
    cpu = None
    if rank == 0:
        reps = 10
        # HIP events must sit on the stream the kernel is launched on: a detector handle created
        # under that stream (the native receiver keeps its own detector inside the library, so the
        # roofline leg uses a second, identically configured one)
        own_detector = sd is None or not hasattr(sd, "correlate_only")
        roof_stream = torch.cuda.Stream() if own_detector else rx._streams[0]
        with torch.cuda.stream(roof_stream):
            if own_detector:
                sd = pkg.SyncwordDetection(rrc, SYNCWORD, bpsk, -BINS, BINS, power_threshold=9.5, max_items=n_items,
                                           n_channels=args.channels)
            # untimed launches first: the legs before this one end in host work, and the first launches after a few idle
            # milliseconds run at the clocks the chip idles at (tools/archive/r5_sustained.py: 2.99 ms for the first ten launches
            # after 3 s of idling against 2.70 - 2.74 ms for every launch of 12 s back to back)
            for _ in range(ROOF_WARM):
                sd.correlate_only(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                sd.correlate_only(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        n_blocks = (n_items - N_FFT) // 1752 + 1
        samples = n_blocks * 1752 * args.channels
        alg_bytes = 8.0 * samples
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        flops = 710.0 * samples  # (1+B) 5N log2 N + 6BN + 1.5N + 4BS per stride, B = 9
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    **pmc_traffic(samples, correlator_kernel()),
                    "kernel": correlator_kernel(),
                    "launch_ms": round(ms, 4), "samples_per_launch": samples,
                    "alg_bytes_per_sample": 8,
                    "fp32_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                    "fp32_frac": round(flops / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                    "ceiling": roofline_ceiling(2 * BINS + 1)}
        roofline["frac_of_ceiling"] = round(roofline["frac"] / roofline["ceiling"]["frac"], 4)
        if args.channels == 1 and not args.no_per_bins:
            roofline["per_bins"] = per_bins_roofline(pkg, rrc, bpsk, x, n_items, roof_stream)
        # the CPU legs come after every GPU leg (256 busy host processes slow the GPU legs' launches), at N = 1 only
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline((x[0] if args.channels > 1 else x)[: min(n_items, 1 << 27)].cpu().numpy(), rrc)
    if rank == 0:
        line = {
            "metric": "RX Msamples/s (syncword-detect + RRC chain)",
            "value": round(total / dt / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "gpus_requested": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "repeats": len(regions), "value_min": min(region_rates), "value_max": max(region_rates),
            "values": region_rates,
            "ms_per_step_per_rank": ranks_ms[median_region],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"configs[2]: {args.channels} channels on one GPU, one batched SyncwordDetection + every "
                                    "channel's own tag gate + CFC + SymbolFilter + wipe-off + Costas (gr4pm_multichannel_receiver)"
                                    if multi is not None else "SyncwordDetection only" if args.detector_only else
                                    "configs[1]: 1 channel/GPU, full RX front end (SyncwordDetection 9 bins FFT 2048 + tag "
                                    "gate + CFC + 32-arm RRC SymbolFilter + wipe-off + Costas)") +
                                   (" + PayloadMetadataInsert + SyncwordRemove + LLR decoder" if (args.soft_bits or args.decode_headers) else "") +
                                   (" + header decode loop" if args.decode_headers else "") +
                                   " on resident burst+AWGN stream",
                       "items_per_step_per_gpu": n_items * args.channels, "channels_per_gpu": args.channels, "freq_bins": 2 * BINS + 1, "tags_per_step": n_tags // max(args.steps, 1),
                       "parallelism": f"channel-per-gpu x{world}", "input": input_mode,
                       "pipelined_streams": 1 if (args.no_pipeline or args.detector_only) else (3 if args.python_pipeline else 4),
                       "pipeline_driver": ("native (gr4pm_multichannel_receiver: submit / collect, 4 batches in flight, "
                                           + ("delayed copy per batch" if args.copy_delay else "input read in place") + ")"
                                           if (multi is not None and not args.python_pipeline and not args.no_pipeline) else
                                           "native (gr4pm_multichannel_receiver, synchronous)" if (multi is not None and not args.python_pipeline) else
                                           "native (gr4pm_packet_receiver)" if native else "python threads"),
                       "windows": 2, "correlator_lookahead": 0 if args.no_lookahead else args.lookahead_depth,
                       **({"headers": hdr_stats} if args.decode_headers else {})},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "job": job,
        }
        if shared:
            line["shared_device_test"] = f"{world} ranks on ONE GPU over gloo: a check of the N > 1 code path, not a measurement"
        if config5_rec is not None:
            line["config5"] = config5_rec
        if host_stream_rec is not None:
            line["host_stream"] = host_stream_rec
        if sparse_rec is not None:
            line["sparse"] = sparse_rec
        if channels_leg is not None:
            if world > 1:
                # 64 channels per GPU are the same work on every rank: the per-GPU rate of configs[3] should be the
                # configs[2] rate of one GPU (the latest committed single-GPU line); flagged when it is not
                ref = config2_reference()
                channels_leg["per_gpu"] = round(channels_leg["value"] / world, 2)
                channels_leg["config2_reference"] = ref
                if ref:
                    channels_leg["within_10pct_of_config2"] = bool(abs(channels_leg["per_gpu"] / ref["value"] - 1.0) <= 0.10)
                    if not channels_leg["within_10pct_of_config2"]:
                        print(f"bench.py: WARNING: config3 per GPU {channels_leg['per_gpu']} Msps vs config2 {ref['value']} "
                              f"({ref['source']}): more than 10 % apart", file=sys.stderr)
            line["config2" if world == 1 else "config3"] = channels_leg
        print(json.dumps(line))
    if dist:
        dist.barrier()  # rank 0 has printed: nobody leaves the group before
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
