/*
 * ref_taps_dump.cpp -- builds oracle/_ref/ref_taps_dump from the reference's OWN standalone
 * headers where they lie under /root/reference (never copied into this repo):
 *   firdes.hpp, packet_transmitter_rrc_taps.hpp, pfb_arb_taps.hpp
 * These three headers need nothing but the C++ standard library, so this is a genuine
 * build of reference code (no stand-ins).  The block headers all include gnuradio4's
 * Block.hpp, which is absent: they are unbuildable here.
 *
 * Usage: ref_taps_dump rrc <gain> <fs> <symrate> <alpha> <ntaps> | txrrc <sps> | pfbarb
 * Output: raw little-endian float32 on stdout.
 */
#include <sys/types.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <gnuradio-4.0/packet-modem/firdes.hpp>
#include <gnuradio-4.0/packet-modem/packet_transmitter_rrc_taps.hpp>
#include <gnuradio-4.0/packet-modem/pfb_arb_taps.hpp>

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    std::vector<float> taps;
    if (!std::strcmp(argv[1], "rrc") && argc == 7) {
        taps = gr::packet_modem::firdes::root_raised_cosine<float>(
            std::atof(argv[2]), std::atof(argv[3]), std::atof(argv[4]), std::atof(argv[5]),
            static_cast<size_t>(std::atol(argv[6])));
    } else if (!std::strcmp(argv[1], "txrrc") && argc == 3) {
        taps = gr::packet_modem::packet_transmitter_rrc_taps(
            static_cast<size_t>(std::atol(argv[2])));
    } else if (!std::strcmp(argv[1], "pfbarb")) {
        taps = gr::packet_modem::pfb_arb_taps;
    } else {
        return 2;
    }
    std::fwrite(taps.data(), sizeof(float), taps.size(), stdout);
    return 0;
}
