"""SURVEY.md 8(f) rank 4, the ZMQ PUB side of the symbol tap (packet_receiver.hpp:159-189, zmq_pdu_pub_sink.hpp:11-44;
consumer scripts/plot_symbols.py:10-17): the library's own ZMTP 3.0 PUB endpoint (gr4pm_zmq_pub_*,
csrc/hostlogic/zmtp_pub.hpp) against
  * a SUB peer written here on a plain TCP socket from the protocol's specification (greeting, NULL handshake,
    subscription messages, short and long frames),
  * and -- where a libzmq happens to be loadable on the machine (this image carries one inside a conda tree; it is not a
    system library and the product does not use it) -- a REAL zmq SUB socket driven through ctypes: the peer the
    reference's plot script is.
Host only: none of this needs a GPU."""
import ctypes as C
import os
import socket
import struct
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

GREETING = bytes([0xFF, 0, 0, 0, 0, 0, 0, 0, 1, 0x7F, 3, 0]) + b"NULL" + bytes(16) + bytes(1) + bytes(31)  # RFC 23, 64 bytes


def ready(socket_type):
    body = b"\x05READY" + b"\x0bSocket-Type" + struct.pack(">I", len(socket_type)) + socket_type
    return bytes([0x04, len(body)]) + body


class RawSub:
    """a ZMTP 3.0 SUB peer on a plain socket"""

    def __init__(self, port, topic=b"", handshake=True):
        self.s = socket.create_connection(("127.0.0.1", port))
        self.s.settimeout(10)
        if handshake:
            self.s.sendall(GREETING)
            self.peer_greeting = self.read(64)
            self.s.sendall(ready(b"SUB"))
            flags, n = self.read(2)
            self.peer_ready = bytes([flags, n]) + self.read(n)
            self.subscribe(topic)

    def subscribe(self, topic):
        self.s.sendall(bytes([0x00, 1 + len(topic), 0x01]) + topic)

    def cancel(self, topic):
        self.s.sendall(bytes([0x00, 1 + len(topic), 0x00]) + topic)

    def read(self, n):
        b = b""
        while len(b) < n:
            c = self.s.recv(n - len(b))
            if not c:
                raise EOFError
            b += c
        return b

    def message(self):
        flags = self.read(1)[0]
        assert flags & 0x05 == 0, "a PUB socket sends single-frame messages, no commands, after the handshake"
        n = struct.unpack(">Q", self.read(8))[0] if flags & 0x02 else self.read(1)[0]
        return self.read(n)

    def close(self):
        self.s.close()


def wait_for(cond, seconds=5.0):
    t0 = time.time()
    while not cond() and time.time() - t0 < seconds:
        time.sleep(0.005)
    return cond()


@pytest.fixture(scope="module")
def pkg():
    if not os.path.exists(os.path.join(ge.PKG_DIR, "libgr4pm_hip.so")):
        ge.build()
    return ge.load_package()


def test_handshake_bytes_and_pdu_frames(pkg):
    """what the endpoint puts on the wire, byte for byte against ZMTP 3.0: the 64-byte greeting (version 3.0, NULL), the
    READY command with Socket-Type PUB, then one frame per PDU -- short (<= 255 bytes: flags 00, one size byte) and long
    (flags 02, eight size bytes, big-endian) -- holding the raw complex64 items (zmq_pdu_pub_sink.hpp:37-40)"""
    pub = pkg.ZmqPduPubSink("tcp://127.0.0.1:*")
    assert pub.port > 0 and pub.subscribers == 0
    pub.process_one(np.zeros(4, np.complex64))  # nobody there: dropped, as a PUB socket does
    sub = RawSub(pub.port)
    assert sub.peer_greeting == GREETING
    assert sub.peer_ready == ready(b"PUB")
    assert wait_for(lambda: pub.subscribers == 1)
    pdus = [np.arange(n, dtype=np.float32).view(np.complex64) for n in (2, 62, 64, 256, 0, 12032)]  # 8 .. 48128 bytes
    for p in pdus:
        pub.process_one(p)
    for p in pdus:
        assert sub.message() == p.tobytes()
    assert pub.dropped == 0
    sub.close()
    assert wait_for(lambda: pub.subscribers == 0)
    pub.close()


def test_subscriptions_are_prefix_filters_per_peer(pkg):
    pub = pkg.ZmqPduPubSink("tcp://127.0.0.1:0")
    everything, only_a, later = RawSub(pub.port), RawSub(pub.port, b"A"), RawSub(pub.port, b"zz")
    assert wait_for(lambda: pub.subscribers == 3)
    for m in (b"A1", b"B1", b"", b"A"):
        pub.process_one(m)
    later.cancel(b"zz")
    later.subscribe(b"B")
    # (a marker through the same connection orders the subscription change against the sends below)
    assert wait_for(lambda: pub.subscribers == 3)
    time.sleep(0.05)
    for m in (b"B2", b"A2"):
        pub.process_one(m)
    assert [everything.message() for _ in range(6)] == [b"A1", b"B1", b"", b"A", b"B2", b"A2"]
    assert [only_a.message() for _ in range(3)] == [b"A1", b"A", b"A2"]
    assert later.message() == b"B2"
    for s in (everything, only_a, later):
        s.close()
    pub.close()


def test_a_subscriber_that_does_not_read_loses_messages_and_nobody_waits(pkg):
    """PUB semantics: send never blocks; a peer whose queue holds 1000 messages (ZMQ_SNDHWM's default) loses the rest,
    a peer that reads gets everything"""
    pub = pkg.ZmqPduPubSink("tcp://127.0.0.1:*")
    mute, reader = RawSub(pub.port), RawSub(pub.port)
    assert wait_for(lambda: pub.subscribers == 2)
    payload = np.zeros(8192, np.complex64)  # 64 KiB per message: socket buffers fill after a few dozen
    got = 0
    t0 = time.time()
    for k in range(3000):
        payload[0] = k
        pub.process_one(payload)
        if k % 50 == 49:  # the reading peer keeps up
            while got <= k:
                assert np.frombuffer(reader.message(), np.complex64)[0].real == got
                got += 1
    assert time.time() - t0 < 60
    assert pub.dropped > 500, pub.dropped   # 3000 sent, ~1000 queued, some dozens in the socket: the rest dropped
    mute.close()
    reader.close()
    pub.close()


def test_refused_endpoints_and_peers(pkg):
    for bad in ("ipc:///tmp/x", "tcp://", "tcp://localhost:5000", "tcp://127.0.0.1:70000", "5000"):
        with pytest.raises(Exception, match="endpoint"):
            pkg.ZmqPduPubSink(bad)
    pub = pkg.ZmqPduPubSink("tcp://127.0.0.1:*")
    with pytest.raises(Exception):  # the port is taken
        pkg.ZmqPduPubSink(f"tcp://127.0.0.1:{pub.port}")
    junk = RawSub(pub.port, handshake=False)
    junk.s.sendall(b"GET / HTTP/1.0\r\n\r\n")
    with pytest.raises((EOFError, ConnectionError, socket.timeout)):  # not a ZMTP peer: the connection is closed
        junk.read(65)  # (at most our greeting arrives)
    old = RawSub(pub.port, handshake=False)
    old.s.sendall(bytes([0xFF, 0, 0, 0, 0, 0, 0, 0, 1, 0x7F, 1]))  # ZMTP 2.0 peer: not served
    with pytest.raises((EOFError, ConnectionError, socket.timeout)):
        old.read(65)
    good = RawSub(pub.port)  # ... and the endpoint lives on
    assert wait_for(lambda: pub.subscribers == 1)
    pub.process_one(b"ok")
    assert good.message() == b"ok"
    pub.close()


LIBZMQ_CANDIDATES = ["libzmq.so.5", "/opt/conda/lib/libzmq.so.5", "/usr/lib/x86_64-linux-gnu/libzmq.so.5"]


def load_libzmq():
    for name in LIBZMQ_CANDIDATES:
        try:
            z = C.CDLL(name)
        except OSError:
            continue
        z.zmq_ctx_new.restype = C.c_void_p
        z.zmq_socket.restype = C.c_void_p
        z.zmq_socket.argtypes = [C.c_void_p, C.c_int]
        z.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
        z.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        z.zmq_close.argtypes = [C.c_void_p]
        z.zmq_ctx_term.argtypes = [C.c_void_p]
        return z
    return None


class LibzmqSub:
    """socket = context.socket(zmq.SUB); socket.connect(...); socket.setsockopt(zmq.SUBSCRIBE, b'') -- plot_symbols.py:10-14"""
    ZMQ_SUB, ZMQ_SUBSCRIBE, ZMQ_RCVTIMEO = 2, 6, 27

    def __init__(self, z, port, topic=b""):
        self.z = z
        self.ctx = z.zmq_ctx_new()
        self.sock = z.zmq_socket(self.ctx, self.ZMQ_SUB)
        assert z.zmq_connect(self.sock, f"tcp://127.0.0.1:{port}".encode()) == 0
        assert z.zmq_setsockopt(self.sock, self.ZMQ_SUBSCRIBE, topic, len(topic)) == 0
        timeout = C.c_int(10000)
        z.zmq_setsockopt(self.sock, self.ZMQ_RCVTIMEO, C.byref(timeout), 4)
        self.buf = C.create_string_buffer(1 << 20)

    def recv(self):
        n = self.z.zmq_recv(self.sock, self.buf, len(self.buf), 0)
        assert n >= 0, "zmq_recv timed out"
        return self.buf.raw[:n]

    def close(self):
        self.z.zmq_close(self.sock)
        self.z.zmq_ctx_term(self.ctx)


def test_a_real_libzmq_sub_socket_receives_the_pdus(pkg):
    """the reference's consumer is a libzmq SUB socket (scripts/plot_symbols.py:10-17).  Where a libzmq can be loaded, one
    connects to the library's endpoint, subscribes to b'' and receives every PDU as `np.frombuffer(message, 'complex64')`
    -- the wire format is pinned against the real implementation, not only against this file's reading of the RFC."""
    z = load_libzmq()
    if z is None:
        pytest.skip("no libzmq on this machine")
    major, minor, patch = C.c_int(), C.c_int(), C.c_int()
    z.zmq_version(C.byref(major), C.byref(minor), C.byref(patch))
    assert major.value >= 4
    pub = pkg.ZmqPduPubSink("tcp://127.0.0.1:*")
    sub, sub_a = LibzmqSub(z, pub.port), LibzmqSub(z, pub.port, b"\x00\x00\x80\x3f")  # topic: items starting with 1.0f
    assert wait_for(lambda: pub.subscribers == 2)
    rng = np.random.default_rng(5)
    pdus = [(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for n in (128, 6016, 1, 0, 31, 32, 100000)]
    pdus[2][0] = 1.0 + 2.0j
    pdus[4][0] = 1.0 - 1.0j
    for p in pdus:
        pub.process_one(p)
    for p in pdus:
        assert np.array_equal(np.frombuffer(sub.recv(), "complex64").view(np.uint64), p.view(np.uint64))
    for p in (pdus[2], pdus[4]):
        assert np.array_equal(np.frombuffer(sub_a.recv(), "complex64").view(np.uint64), p.view(np.uint64))
    sub.close()
    sub_a.close()
    pub.close()


def test_gr4_zmq_pdu_pub_sink_wrapper_through_processOne():
    """the GR4 drop-in ZmqPduPubSink<c64> (host/gnuradio-4.0/packet-modem/zmq_pdu_pub_sink.hpp) as PacketReceiver wires
    it: `endpoint` setting, start() binds, processOne(Pdu) publishes; a hand-written SUB receives the PDUs of
    tests/gr4_blocks_driver.cpp's `zmq` mode"""
    ge.build_gr4_driver()
    n = 9
    p = subprocess.Popen([ge.GR4_DRIVER, "zmq", str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        line = p.stdout.readline()
        assert line.startswith("port "), line + p.stderr.read()
        sub = RawSub(int(line.split()[1]))
        for k in range(n):
            z = np.frombuffer(sub.message(), np.complex64)
            assert z.size == (128 if k % 3 == 0 else 100 + 37 * k)
            assert np.all(z.real == k) and np.array_equal(z.imag, np.arange(z.size, dtype=np.float32))
        out, err = p.communicate(timeout=30)
        assert p.returncode == 0 and f"published {n}" in out, out + err
        sub.close()
    finally:
        if p.poll() is None:
            p.kill()
