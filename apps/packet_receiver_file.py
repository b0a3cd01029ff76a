#!/usr/bin/env python3
"""MI355X counterpart of the reference's apps/packet_receiver_file.cpp (apps/README.md:5-24):

    packet_receiver_file.py input_file [syncword_freq_bins=4] [syncword_threshold=9.5] [--out packets.bin] [--zmq]

reads IQ samples from `input_file` in raw little-endian complex64 (std::complex<float>, what
FileSource<c64> freads, file_source.hpp:32,53) at 4 samples/symbol, runs the whole receiver on
the GPU (gr4pm_packet_receiver, decode_headers: detection ... header decode ... CRC check) and
hands over the packets whose CRC-32 matches.  The reference writes them to a TUN device (needs
root and a network namespace); here they go to `--out` as records of a big-endian uint16 length
followed by the bytes, or are just counted.

`--zmq` is the reference app's `zmq_output = true` (apps/packet_receiver_file.cpp, packet_receiver.hpp:159-189): the
header symbols of every packet are published as one ZeroMQ message of raw complex64 on tcp port 5000 and the payload
symbols on 5001 (`--zmq-ports H P` for others), where scripts/plot_symbols.py of the reference connects its SUB sockets;
the library speaks the ZeroMQ wire protocol itself (gr4pm_zmq_pub_*), libzmq is not needed.

The file is streamed: host chunks are staged in pinned memory and copied to the device on a
copy stream while the previous chunk is being processed; the samples the detector leaves
unconsumed (less than one FFT block) are presented again in front of the next chunk."""
import argparse
import os
import sys
import time

# A recording is mostly silence between packets: the stretch between two detections is one serial phasor chain, and the
# library runs the chains of consecutive batches side by side where the HIP runtime has hardware queues to spare
# ("Process settings"; read when the runtime starts, i.e. before torch is imported)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def receive_file(path, syncword_freq_bins=4, syncword_threshold=9.5, chunk_items=1 << 24, out=None, pkg=None, zmq_ports=None):
    """returns dict(packets: list of bytes, items, seconds, headers, invalid_headers, crc_failures)"""
    pkg = pkg or ge.load_package()
    n_file = os.path.getsize(path) // 8
    dev = torch.device("cuda", torch.cuda.current_device())
    # packets_only: nothing between the Costas loop and the packer is written to memory (the app delivers packets); the
    # symbol tap of --zmq needs SyncwordRemove's output stream, i.e. the full form
    rx = pkg.NativePacketReceiver(4, syncword_freq_bins, syncword_threshold, max_items=chunk_items + 4096,
                                  tags_cap=chunk_items // 768 + 64, decode_headers=True, packets_only=zmq_ports is None)
    if zmq_ports is not None:  # packet_receiver.hpp:163-168
        rx.publish_symbol_pdus(f"tcp://*:{zmq_ports[0]}", f"tcp://*:{zmq_ports[1]}")
    fft = 3072  # smallest batch the receiver takes in this mode (one header window + one FFT block)
    pinned = [torch.empty(chunk_items, dtype=torch.complex64).pin_memory() for _ in range(2)]
    staged = [torch.empty(chunk_items, dtype=torch.complex64, device=dev) for _ in range(2)]
    work = torch.empty(chunk_items + 4096, dtype=torch.complex64, device=dev)
    copy_stream = torch.cuda.Stream()
    events = [torch.cuda.Event(), torch.cuda.Event()]
    packets, stats = [], {"headers": 0, "invalid_headers": 0, "crc_failures": 0}
    fh = open(path, "rb")

    def load(slot):
        """file -> pinned -> device (asynchronously on the copy stream); returns items read"""
        view = pinned[slot].numpy().view(np.uint8)
        got = fh.readinto(memoryview(view)) // 8
        if got:
            with torch.cuda.stream(copy_stream):
                staged[slot][:got].copy_(pinned[slot][:got], non_blocking=True)
                events[slot].record(copy_stream)
        return got

    def deliver(res):
        m = res["header_messages"]
        stats["headers"] += int(m.size)
        stats["invalid_headers"] += int(np.sum(m["invalid_header"] != 0))
        lens = res["packet_lengths"]
        stats["crc_failures"] += int(np.sum(lens == 0))
        data = res["packets"].cpu().numpy()
        pos = 0
        for n in lens[lens > 0]:
            n = int(n)
            packets.append(data[pos:pos + n].tobytes())
            pos += n

    t0 = time.perf_counter()
    slot, left, done = 0, 0, 0
    got = load(slot)
    while got or left >= fft:
        nxt = 0
        if got:
            torch.cuda.current_stream().wait_event(events[slot])
            work[left:left + got].copy_(staged[slot][:got])
            nxt = load(slot ^ 1)          # the next chunk travels while this one is processed
        n = left + got
        if n < fft:
            break
        res = rx.process_bulk(work[:n])
        deliver(res)
        c = res["consumed"]
        done += c
        left = n - c
        if left:
            work[:left].copy_(work[c:n].clone())
        slot ^= 1
        got = nxt
        if not got and left < fft:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fh.close()
    if out:
        with open(out, "wb") as f:
            for p in packets:
                f.write(len(p).to_bytes(2, "big"))
                f.write(p)
    return {"packets": packets, "items": done, "file_items": n_file, "seconds": dt, **stats}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input_file")
    ap.add_argument("syncword_freq_bins", nargs="?", type=int, default=4)   # packet_receiver_file.cpp:25
    ap.add_argument("syncword_threshold", nargs="?", type=float, default=9.5)  # :26
    ap.add_argument("--out", help="write the received packets here (uint16 big-endian length + bytes each)")
    ap.add_argument("--chunk-items", type=int, default=1 << 24)
    ap.add_argument("--zmq", action="store_true", help="publish header / payload symbol PDUs on ZeroMQ PUB sockets (tcp 5000 / 5001)")
    ap.add_argument("--zmq-ports", type=int, nargs=2, metavar=("HEADER", "PAYLOAD"), help="... on these ports instead")
    a = ap.parse_args()
    zmq_ports = tuple(a.zmq_ports) if a.zmq_ports else ((5000, 5001) if a.zmq else None)
    r = receive_file(a.input_file, a.syncword_freq_bins, a.syncword_threshold, a.chunk_items, a.out, zmq_ports=zmq_ports)
    print(f"{r['items']} of {r['file_items']} samples in {r['seconds']:.3f} s = {r['items'] / r['seconds'] / 1e6:.1f} Msps "
          f"(file and PCIe included); headers {r['headers']} ({r['invalid_headers']} invalid), packets "
          f"{len(r['packets'])} ({r['crc_failures']} CRC failures), {sum(len(p) for p in r['packets'])} bytes")


if __name__ == "__main__":
    main()
