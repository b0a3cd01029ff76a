// pdu_pub.hip -- C ABI of the ZeroMQ PUB endpoint (hostlogic/zmtp_pub.hpp): replaces ZmqPduPubSink<T>
// (zmq_pdu_pub_sink.hpp:11-44), the sink of the receiver's symbol tap (packet_receiver.hpp:159-189).  Host code only
// (no kernel, no device needed): a .hip file so that it is built and guarded like the other ABI units.
#include <memory>

#include "common.hpp"
#include "hostlogic/zmtp_pub.hpp"

struct gr4pm_zmq_pub {
    gr4pm::hostlogic::ZmtpPub pub;
};

using namespace gr4pm;

extern "C" {

gr4pm_status gr4pm_zmq_pub_create(const char* endpoint, gr4pm_zmq_pub** out)
try {
    if (!out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<gr4pm_zmq_pub> h(new (std::nothrow) gr4pm_zmq_pub); // (bind() starts a thread: that may throw)
    if (!h) return GR4PM_ERR_NOMEM;
    const gr4pm_status st = h->pub.bind(endpoint); // start(), zmq_pdu_pub_sink.hpp:29
    if (st != GR4PM_OK) return st;
    *out = h.release();
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

void gr4pm_zmq_pub_destroy(gr4pm_zmq_pub* h)
try {
    delete h;
}
GR4PM_ABI_CATCH_VOID

gr4pm_status gr4pm_zmq_pub_send(gr4pm_zmq_pub* h, const void* data, size_t bytes)
try {
    if (!h || (!data && bytes)) return GR4PM_ERR_INVALID;
    return h->pub.send(data, bytes); // processOne(), zmq_pdu_pub_sink.hpp:31-41
}
GR4PM_ABI_CATCH

int gr4pm_zmq_pub_port(const gr4pm_zmq_pub* h)
try {
    return h ? h->pub.port() : -1;
}
GR4PM_ABI_CATCH_RET(-1)

size_t gr4pm_zmq_pub_subscribers(const gr4pm_zmq_pub* h)
try {
    return h ? h->pub.subscribers() : 0;
}
GR4PM_ABI_CATCH_RET(0)

uint64_t gr4pm_zmq_pub_dropped(const gr4pm_zmq_pub* h)
try {
    return h ? h->pub.dropped() : 0;
}
GR4PM_ABI_CATCH_RET(0)

} // extern "C"
