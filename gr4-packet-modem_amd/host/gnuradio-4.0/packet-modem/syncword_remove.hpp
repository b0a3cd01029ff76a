// Drop-in for the reference header of the same name (blocks/include/gnuradio-4.0/packet-modem/syncword_remove.hpp):
// with gr4-packet-modem_amd/host in front of the reference's blocks/include on the include path, a flowgraph
// that includes <gnuradio-4.0/packet-modem/syncword_remove.hpp> gets the MI355X block under the reference's own name.
#pragma once
#include "../../gr4pm_gr4_blocks.hpp"

namespace gr::packet_modem {
using hip::SyncwordRemove;
} // namespace gr::packet_modem
