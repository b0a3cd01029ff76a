// TEST-ONLY stand-in for the slice of cppzmq that zmq_pdu_pub_sink.hpp:22-41 touches (libzmq is not in this image): it
// lets the reference's packet_receiver.hpp -- which includes the sink unconditionally -- go through a compiler.  bind()
// reports that there is no socket: a flowgraph built with zmq_output = true fails at start(), it does not pretend.
#pragma once
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

namespace zmq {
enum class socket_type { pub = 1 };
enum class send_flags { none = 0 };
struct context_t {};
class message_t
{
    std::vector<unsigned char> _d;

public:
    explicit message_t(size_t n) : _d(n) {}
    void* data() { return _d.data(); }
    size_t size() const { return _d.size(); }
};
struct socket_t {
    socket_t(context_t&, socket_type) {}
    void bind(const std::string& endpoint) { throw std::runtime_error("zmq stand-in: no libzmq in this image (" + endpoint + ")"); }
    bool send(message_t&, send_flags) { return false; }
};
} // namespace zmq
