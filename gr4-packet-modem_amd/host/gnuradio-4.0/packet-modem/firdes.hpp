// Drop-in for blocks/include/gnuradio-4.0/packet-modem/firdes.hpp: firdes::root_raised_cosine<float>
// (firdes.hpp:29-76) from the library's own design routine (bit-exact against the reference's, tests/golden).
#pragma once
#include "../../gr4pm_gr4_blocks.hpp"

namespace gr::packet_modem::firdes {
using hip::firdes::root_raised_cosine;
} // namespace gr::packet_modem::firdes
