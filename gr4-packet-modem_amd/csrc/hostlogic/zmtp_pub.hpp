// hostlogic/zmtp_pub.hpp -- a ZeroMQ PUB endpoint without libzmq: what ZmqPduPubSink (zmq_pdu_pub_sink.hpp:11-44) puts
// on the wire for the symbol tap of packet_receiver.hpp:159-189 (tcp://*:5000 header PDUs, :5001 payload PDUs; consumer
// scripts/plot_symbols.py:10-17: a zmq.SUB socket subscribed to b'').
//
// The reference's socket is cppzmq over libzmq, a dependency that is not part of /root/reference and not installed as a
// system library here.  What travels between a PUB and a SUB socket is ZMTP (ZeroMQ Message Transport Protocol,
// https://rfc.zeromq.org/spec/23/ "ZMTP 3.0"), restated here for the one direction a PUB socket needs:
//
//   greeting, 64 bytes each way:  FF 00*7 01 7F | 03 00 (version 3.0) | "NULL" padded to 20 | 00 (as-server) | 00*31
//   NULL handshake, each way:     command frame  04 <len> | 05 "READY" | 0B "Socket-Type" 00 00 00 03 "PUB" (peer: "SUB")
//   frames:                       flags (01 MORE, 02 LONG size, 04 COMMAND) | size: 1 byte, or 8 bytes big-endian | body
//   SUB -> PUB (ZMTP 3.0):        message frame with body 01 <topic> = subscribe, 00 <topic> = cancel
//   PUB -> SUB:                   one single-frame message per PDU = the raw items (zmq_pdu_pub_sink.hpp:37-40), sent to
//                                 every peer that holds a subscription whose topic is a prefix of the message
//
// PUB semantics kept: no subscriber -> the message is dropped; a slow subscriber -> its queue fills to the high-water
// mark (libzmq's default, 1000 messages) and further messages to it are dropped; send() never blocks on the network.
// One I/O thread per endpoint (poll()): accepts, runs the handshake state machine of every connection, reads the
// subscriptions, drains the queues.  No HIP; compiled as it stands by tests/hostlogic under the sanitizers.
#pragma once
#include <arpa/inet.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "base.hpp"

namespace gr4pm {
namespace hostlogic {

class ZmtpPub
{
public:
    static constexpr size_t kHighWaterMark = 1000; // ZMQ_SNDHWM default

    ZmtpPub() = default;
    ZmtpPub(const ZmtpPub&) = delete;
    ZmtpPub& operator=(const ZmtpPub&) = delete;
    ~ZmtpPub() { close(); }

    // endpoint: "tcp://*:5000", "tcp://0.0.0.0:5000", "tcp://127.0.0.1:0" ("*" / "0" as the port: an ephemeral one).
    // = socket.bind(endpoint), zmq_pdu_pub_sink.hpp:29
    gr4pm_status bind(const char* endpoint)
    {
        if (_listen >= 0) {
            set_error("zmq pub: already bound");
            return GR4PM_ERR_INVALID;
        }
        sockaddr_in addr{};
        if (!parse(endpoint, addr)) {
            set_error("zmq pub: cannot parse endpoint '%s' (tcp://<ipv4 | *>:<port | *>)", endpoint ? endpoint : "(null)");
            return GR4PM_ERR_INVALID;
        }
        _listen = ::socket(AF_INET, SOCK_STREAM | SOCK_NONBLOCK | SOCK_CLOEXEC, 0);
        if (_listen < 0) return fail("socket");
        const int one = 1;
        (void)::setsockopt(_listen, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (::bind(_listen, reinterpret_cast<const sockaddr*>(&addr), sizeof addr) != 0 || ::listen(_listen, 64) != 0) {
            const gr4pm_status st = fail("bind / listen");
            ::close(_listen);
            _listen = -1;
            return st;
        }
        socklen_t len = sizeof addr;
        (void)::getsockname(_listen, reinterpret_cast<sockaddr*>(&addr), &len);
        _port = ntohs(addr.sin_port);
        if (::pipe2(_wake, O_NONBLOCK | O_CLOEXEC) != 0) {
            const gr4pm_status st = fail("pipe2");
            ::close(_listen);
            _listen = -1;
            return st;
        }
        _stop = false;
        try {
            _io = std::thread([this] { run(); });
        } catch (...) { // no thread: nothing is bound
            ::close(_listen);
            ::close(_wake[0]);
            ::close(_wake[1]);
            _listen = -1;
            throw;
        }
        return GR4PM_OK;
    }

    // one message = one PDU's raw items (zmq_pdu_pub_sink.hpp:31-41).  Never blocks on a peer.
    gr4pm_status send(const void* data, size_t bytes)
    {
        if (_listen < 0) {
            set_error("zmq pub: not bound");
            return GR4PM_ERR_INVALID;
        }
        auto frame = std::make_shared<std::vector<uint8_t>>();
        frame->reserve(bytes + 9);
        if (bytes <= 255) {
            frame->push_back(0x00);
            frame->push_back(static_cast<uint8_t>(bytes));
        } else {
            frame->push_back(0x02); // LONG
            for (int s = 56; s >= 0; s -= 8) frame->push_back(static_cast<uint8_t>(static_cast<uint64_t>(bytes) >> s));
        }
        const uint8_t* p = static_cast<const uint8_t*>(data);
        frame->insert(frame->end(), p, p + bytes);
        bool queued = false;
        {
            std::lock_guard<std::mutex> g(_m);
            ++_sent;
            for (auto& c : _conns) {
                if (c->state != Conn::ACTIVE || !c->matches(p, bytes)) continue;
                if (c->tx.size() >= kHighWaterMark) {
                    ++_dropped;
                    continue;
                }
                c->tx.push_back({ frame, 0 });
                queued = true;
            }
        }
        if (queued) wake();
        return GR4PM_OK;
    }

    int port() const { return _port; }
    size_t subscribers() const // peers that hold at least one subscription
    {
        std::lock_guard<std::mutex> g(_m);
        size_t n = 0;
        for (const auto& c : _conns) n += c->state == Conn::ACTIVE && !c->topics.empty();
        return n;
    }
    uint64_t sent() const { return _sent.load(); }
    uint64_t dropped() const { return _dropped.load(); } // messages not queued to a subscribed peer (its queue was full)

    // stops the endpoint; what is queued gets `linger_ms` to leave
    void close(int linger_ms = 200)
    {
        if (_listen < 0) return;
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(linger_ms);
        for (;;) {
            {
                std::lock_guard<std::mutex> g(_m);
                bool pending = false;
                for (const auto& c : _conns) pending |= c->state == Conn::ACTIVE && !c->tx.empty();
                if (!pending) break;
            }
            if (std::chrono::steady_clock::now() >= until) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        _stop = true;
        wake();
        if (_io.joinable()) _io.join();
        for (auto& c : _conns) ::close(c->fd);
        _conns.clear();
        ::close(_listen);
        ::close(_wake[0]);
        ::close(_wake[1]);
        _listen = -1;
    }

    // the 64-byte greeting and the READY command of this side (tests compare them with the specification's bytes)
    static std::vector<uint8_t> greeting()
    {
        std::vector<uint8_t> g(64, 0);
        g[0] = 0xFF;
        g[8] = 0x01;
        g[9] = 0x7F;
        g[10] = 3; // version 3.0: subscriptions arrive as messages (3.1 turned them into commands)
        g[11] = 0;
        std::memcpy(&g[12], "NULL", 4);
        return g;
    }
    static std::vector<uint8_t> ready()
    {
        static const char body[] = "\x05READY\x0BSocket-Type\x00\x00\x00\x03PUB";
        std::vector<uint8_t> r{ 0x04, static_cast<uint8_t>(sizeof body - 1) };
        r.insert(r.end(), body, body + sizeof body - 1);
        return r;
    }

private:
    struct Out {
        std::shared_ptr<std::vector<uint8_t>> frame;
        size_t off;
    };
    struct Conn {
        enum State { GREETING, HANDSHAKE, ACTIVE, DEAD } state = GREETING;
        int fd = -1;
        std::vector<uint8_t> rx;
        std::deque<Out> tx;
        std::vector<std::vector<uint8_t>> topics; // a multiset: one entry per subscribe, one removed per cancel
        bool matches(const uint8_t* p, size_t n) const
        {
            for (const auto& t : topics)
                if (t.size() <= n && (t.empty() || std::memcmp(t.data(), p, t.size()) == 0)) return true;
            return false;
        }
    };

    static bool parse(const char* ep, sockaddr_in& a)
    {
        if (!ep || std::strncmp(ep, "tcp://", 6) != 0) return false;
        const std::string rest(ep + 6);
        const size_t colon = rest.rfind(':');
        if (colon == std::string::npos || colon + 1 >= rest.size()) return false;
        const std::string host = rest.substr(0, colon), port = rest.substr(colon + 1);
        a.sin_family = AF_INET;
        if (host == "*") a.sin_addr.s_addr = htonl(INADDR_ANY);
        else if (::inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) return false;
        if (port == "*") {
            a.sin_port = 0;
        } else {
            char* end = nullptr;
            const long v = std::strtol(port.c_str(), &end, 10);
            if (*end != 0 || v < 0 || v > 65535) return false;
            a.sin_port = htons(static_cast<uint16_t>(v));
        }
        return true;
    }
    gr4pm_status fail(const char* what)
    {
        set_error("zmq pub: %s failed: %s", what, std::strerror(errno));
        return GR4PM_ERR_INTERNAL;
    }
    void wake()
    {
        const char b = 1;
        (void)!::write(_wake[1], &b, 1);
    }

    void queue_bytes(Conn& c, std::vector<uint8_t> bytes) // caller holds _m
    {
        c.tx.push_back({ std::make_shared<std::vector<uint8_t>>(std::move(bytes)), 0 });
    }

    // what has arrived on c: greeting, then frames.  Returns false when the peer is not a ZMTP 3 peer we can serve.
    bool parse_rx(Conn& c) // caller holds _m
    {
        if (c.state == Conn::GREETING) {
            if (c.rx.size() >= 10 && (c.rx[0] != 0xFF || !(c.rx[9] & 1))) return false; // not a ZMTP signature
            if (c.rx.size() >= 11 && c.rx[10] < 3) return false;                            // ZMTP 1.0 / 2.0 peers: not served
            if (c.rx.size() < 64) return true;
            if (std::memcmp(&c.rx[12], "NULL\0", 5) != 0) return false; // PLAIN / CURVE: the sink uses neither
            c.rx.erase(c.rx.begin(), c.rx.begin() + 64);
            c.state = Conn::HANDSHAKE;
            queue_bytes(c, ready());
        }
        for (;;) { // frames
            if (c.rx.size() < 2) return true;
            const uint8_t flags = c.rx[0];
            size_t hdr = 2;
            uint64_t len = c.rx[1];
            if (flags & 0x02) {
                if (c.rx.size() < 9) return true;
                len = 0;
                for (int i = 0; i < 8; ++i) len = (len << 8) | c.rx[1 + static_cast<size_t>(i)];
                hdr = 9;
            }
            if (len > (1u << 20)) return false; // a subscriber has nothing that long to say
            if (c.rx.size() < hdr + len) return true;
            const uint8_t* body = c.rx.data() + hdr;
            if (flags & 0x04) { // command
                if (c.state == Conn::HANDSHAKE) {
                    if (len < 6 || body[0] != 5 || std::memcmp(body + 1, "READY", 5) != 0) return false; // ERROR, or junk
                    c.state = Conn::ACTIVE;
                } // (commands of an active peer -- none are defined for ZMTP 3.0 beyond the handshake -- are ignored)
            } else if (c.state == Conn::ACTIVE && len >= 1 && !(flags & 0x01)) {
                std::vector<uint8_t> topic(body + 1, body + len);
                if (body[0] == 1) {
                    c.topics.push_back(std::move(topic));
                } else if (body[0] == 0) {
                    const auto it = std::find(c.topics.begin(), c.topics.end(), topic);
                    if (it != c.topics.end()) c.topics.erase(it);
                }
            } else if (c.state != Conn::ACTIVE) {
                return false; // a message before READY
            }
            c.rx.erase(c.rx.begin(), c.rx.begin() + static_cast<ptrdiff_t>(hdr + len));
        }
    }

    void run()
    {
        std::vector<pollfd> fds;
        std::vector<Conn*> who;
        while (!_stop) {
            fds.clear();
            who.clear();
            fds.push_back({ _listen, POLLIN, 0 });
            fds.push_back({ _wake[0], POLLIN, 0 });
            {
                std::lock_guard<std::mutex> g(_m);
                for (auto& c : _conns) {
                    fds.push_back({ c->fd, static_cast<short>(POLLIN | (c->tx.empty() ? 0 : POLLOUT)), 0 });
                    who.push_back(c.get());
                }
            }
            if (::poll(fds.data(), fds.size(), 250) < 0 && errno != EINTR) break;
            if (fds[1].revents & POLLIN) {
                char buf[64];
                while (::read(_wake[0], buf, sizeof buf) > 0) {}
            }
            if (fds[0].revents & POLLIN) {
                for (;;) {
                    const int fd = ::accept4(_listen, nullptr, nullptr, SOCK_NONBLOCK | SOCK_CLOEXEC);
                    if (fd < 0) break;
                    const int one = 1;
                    (void)::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                    auto c = std::make_unique<Conn>();
                    c->fd = fd;
                    std::lock_guard<std::mutex> g(_m);
                    queue_bytes(*c, greeting()); // sent at once: a libzmq peer waits for our signature before it goes on
                    _conns.push_back(std::move(c));
                }
            }
            for (size_t i = 0; i < who.size(); ++i) {
                Conn& c = *who[i];
                const short ev = fds[i + 2].revents;
                if (ev & (POLLERR | POLLHUP | POLLNVAL)) c.state = Conn::DEAD;
                if (c.state != Conn::DEAD && (ev & POLLIN)) {
                    uint8_t buf[4096];
                    for (;;) {
                        const ssize_t n = ::recv(c.fd, buf, sizeof buf, 0);
                        if (n > 0) {
                            std::lock_guard<std::mutex> g(_m);
                            c.rx.insert(c.rx.end(), buf, buf + n);
                            if (!parse_rx(c)) c.state = Conn::DEAD;
                        } else {
                            if (n == 0 || (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) c.state = Conn::DEAD;
                            break;
                        }
                        if (c.state == Conn::DEAD) break;
                    }
                }
                if (c.state != Conn::DEAD) { // drain what the socket takes (also right after parse_rx queued READY)
                    for (;;) {
                        Out o;
                        {
                            std::lock_guard<std::mutex> g(_m);
                            if (c.tx.empty()) break;
                            o = c.tx.front();
                        }
                        const ssize_t n = ::send(c.fd, o.frame->data() + o.off, o.frame->size() - o.off, MSG_NOSIGNAL);
                        if (n < 0) {
                            if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) c.state = Conn::DEAD;
                            break;
                        }
                        std::lock_guard<std::mutex> g(_m);
                        c.tx.front().off += static_cast<size_t>(n);
                        if (c.tx.front().off == c.tx.front().frame->size()) c.tx.pop_front();
                    }
                }
            }
            std::lock_guard<std::mutex> g(_m);
            for (size_t i = _conns.size(); i-- > 0;)
                if (_conns[i]->state == Conn::DEAD) {
                    ::close(_conns[i]->fd);
                    _conns.erase(_conns.begin() + static_cast<ptrdiff_t>(i));
                }
        }
    }

    int _listen = -1, _port = 0;
    int _wake[2] = { -1, -1 };
    std::atomic<bool> _stop{ false };
    std::atomic<uint64_t> _sent{ 0 }, _dropped{ 0 };
    mutable std::mutex _m; // connections: their state, subscriptions and queues
    std::vector<std::unique_ptr<Conn>> _conns;
    std::thread _io;
};

} // namespace hostlogic
} // namespace gr4pm
