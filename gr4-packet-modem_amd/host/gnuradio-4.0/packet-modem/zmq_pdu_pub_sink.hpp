// Drop-in for the reference header of the same name (blocks/include/gnuradio-4.0/packet-modem/zmq_pdu_pub_sink.hpp):
// with gr4-packet-modem_amd/host in front of the reference's blocks/include on the include path, a flowgraph that
// includes <gnuradio-4.0/packet-modem/zmq_pdu_pub_sink.hpp> (packet_receiver.hpp does, for its zmq_output tap) gets
// the library's ZMTP 3.0 PUB endpoint under the reference's own name -- and no longer needs cppzmq / libzmq.
#pragma once
#if __has_include(<gnuradio-4.0/packet-modem/pdu.hpp>)
#include <gnuradio-4.0/packet-modem/pdu.hpp> // the reference's own Pdu<T> (not replaced: it is a data format, pdu.hpp:15-21)
#endif
#include "../../gr4pm_gr4_blocks.hpp"

namespace gr::packet_modem {
using hip::ZmqPduPubSink;
} // namespace gr::packet_modem
