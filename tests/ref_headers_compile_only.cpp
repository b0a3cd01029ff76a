// Build-container-only: the reference's remaining receive-side block headers are COMPILED against the API stand-in
// (tests/gr4_stub/) with their processBulk() / processOne() instantiated on the stand-in's span types -- the other
// half of "the stand-in declares the API surface the reference uses" for the blocks tests/ref_headers_check.cpp does not
// drive.  Nothing is run.
#include <gnuradio-4.0/packet-modem/additive_scrambler.hpp>
#include <gnuradio-4.0/packet-modem/constellation_llr_decoder.hpp>
#include <gnuradio-4.0/packet-modem/header_payload_split.hpp>
#include <gnuradio-4.0/packet-modem/payload_metadata_insert.hpp>
#include <gnuradio-4.0/packet-modem/syncword_detection_filter.hpp>
#include <gnuradio-4.0/packet-modem/syncword_remove.hpp>

using namespace gr::packet_modem;
using c64 = std::complex<float>;

int main(int argc, char**)
{
    if (argc > 1000) { // instantiate, never execute
        std::vector<c64> x(16), y(16);
        std::vector<float> f(32), g(32);
        std::vector<gr::Message> m(1);
        gr::InSpan<c64> is(x.data(), x.size());
        gr::OutSpan<c64> os(y.data(), y.size());
        gr::InSpan<gr::Message> ms(m.data(), 0), ms2(m.data(), 0);
        gr::OutSpan<gr::Message> mo(m.data(), 1);
        SyncwordDetectionFilter<> sdf;
        sdf.start();
        (void)sdf.processBulk(ms, ms2, is, os);
        PayloadMetadataInsert<> pmi;
        (void)pmi.processBulk(ms, is, os, mo);
        SyncwordRemove<> sr;
        (void)sr.processBulk(is, os);
        ConstellationLLRDecoder<> llr;
        gr::OutSpan<float> fo(g.data(), g.size());
        (void)llr.processBulk(is, fo);
        AdditiveScrambler<float> scr;
        (void)scr.processOne(1.0f);
        HeaderPayloadSplit<> hps;
        gr::InSpan<float> fi(f.data(), f.size());
        gr::OutSpan<float> h(g.data(), 16), p(g.data() + 16, 16);
        (void)hps.processBulk(fi, h, p);
    }
    return 0;
}
