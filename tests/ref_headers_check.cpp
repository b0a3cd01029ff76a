// Build-container-only check of the test stand-in tests/gr4_stub/ in the OTHER direction: the reference's own block
// headers (read where they lie, /root/reference/blocks/include -- nothing is copied) are compiled against the
// stand-in and driven by the same miniature scheduler as the HIP wrappers.  If they compile, the stand-in declares the
// API surface the reference uses; their outputs are then compared with the CPU oracle (tests/test_gr4_blocks.py) on
// the paths no qa_*.cpp of the reference covers: SymbolFilter fed with tags, CoarseFrequencyCorrection with
// delay = 26, PfbArbResampler at a float rate, CostasLoop with phase tags.
// This is NOT a reference build (the API under the blocks is ours) and pins nothing: it catches restatement slips.
//
// usage: ref_headers_check <case> <in.c64> <tags.bin|-> <out_prefix>
#include <gnuradio-4.0/packet-modem/coarse_frequency_correction.hpp>
#include <gnuradio-4.0/packet-modem/costas_loop.hpp>
#include <gnuradio-4.0/packet-modem/interpolating_fir_filter.hpp>
#include <gnuradio-4.0/packet-modem/pfb_arb_resampler.hpp>
#include <gnuradio-4.0/packet-modem/pfb_arb_taps.hpp>
#include <gnuradio-4.0/packet-modem/rotator.hpp>
#include <gnuradio-4.0/packet-modem/symbol_filter.hpp>
#include <gnuradio-4.0/packet-modem/syncword_detection.hpp>
#include <gnuradio-4.0/packet-modem/syncword_wipeoff.hpp>
#include <gnuradio-4.0/packet-modem/firdes.hpp>

#include "gr4_mini_scheduler.hpp"

using namespace gr::packet_modem;

struct InTag { // tests/_oracle.py TAG_DTYPE
    uint64_t index;
    float amplitude, phase;
    double freq;
    int32_t freq_bin;
    float noise_power, esn0_db, time_est;
    int32_t flags;
};
static_assert(sizeof(InTag) == 48, "record layout");

static std::vector<gr::Tag> read_tags(const char* path)
{
    std::vector<gr::Tag> tags;
    if (std::strcmp(path, "-") == 0) return tags;
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error("cannot read tags");
    InTag t;
    while (std::fread(&t, sizeof t, 1, f) == 1) {
        gr::property_map m;
        if (t.flags & 1) { // a detection tag as SyncwordDetection publishes it (syncword_detection.hpp:92-113)
            m["syncword_amplitude"] = t.amplitude;
            m["syncword_phase"] = t.phase;
            m["syncword_freq"] = t.freq;
            m["syncword_freq_bin"] = t.freq_bin;
            m["syncword_noise_power"] = t.noise_power;
            m["syncword_esn0_db"] = t.esn0_db;
            m["syncword_time_est"] = t.time_est;
        } else {
            m["other"] = int32_t{ 1 };
        }
        tags.push_back({ static_cast<ssize_t>(t.index), m });
    }
    std::fclose(f);
    return tags;
}

// blocks with processOne() only (Rotator): what the runtime's default processBulk does
template <typename Blk, typename IS, typename OS>
static gr::work::Status bulk_of_one(Blk& b, IS& is, OS& os)
{
    const size_t n = std::min(is.size(), os.size());
    for (size_t i = 0; i < n; ++i) os[i] = b.processOne(is[i]);
    (void)is.consume(n);
    os.publish(n);
    return gr::work::Status::OK;
}

int main(int argc, char** argv)
{
    try {
        if (argc < 5) return 2;
        const std::string what = argv[1], prefix = argv[4];
        const auto x = read_c64(argv[2]);
        Edge<c64> in(x.size()), out(8 * x.size() + 4096);
        std::copy(x.begin(), x.end(), in.data.begin());
        in.size = x.size();
        in.tags = read_tags(argv[3]);
        gr::stub::Graph fg;
        const size_t chunk = argc > 5 ? static_cast<size_t>(std::atoll(argv[5])) : 4000;
        if (what == "rotator") {
            auto& b = fg.emplaceBlock<Rotator<>>({ { "phase_incr", 0.1f } });
            b.start();
            run(b, in, out, chunk, [&](auto& is, auto& os) { return bulk_of_one(b, is, os); });
        } else if (what == "cfc") {
            auto& b = fg.emplaceBlock<CoarseFrequencyCorrection<>>({ { "delay", size_t{ 26 } } });
            run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
        } else if (what == "symbol_filter") {
            float norm = 0.0f;
            auto rrc = firdes::root_raised_cosine(1.0, 4.0, 1.0, 0.35, 44);
            for (float t : rrc) norm += t * t;
            norm = std::sqrt(norm);
            auto pfb = firdes::root_raised_cosine(32.0 / static_cast<double>(norm), 128.0, 1.0, 0.35, 32 * 4 * 11U);
            pfb.pop_back();
            auto& b = fg.emplaceBlock<SymbolFilter<c64, c64, float>>(
                { { "taps", pfb }, { "num_arms", size_t{ 32 } }, { "samples_per_symbol", size_t{ 4 } }, { "delay", size_t{ 44 } } });
            b.start();
            run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
        } else if (what == "costas") {
            auto& b = fg.emplaceBlock<CostasLoop<>>({ { "constellation", "QPSK" }, { "loop_bandwidth", 0.01 } });
            run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
        } else if (what == "interp_fir") {
            auto& b = fg.emplaceBlock<InterpolatingFirFilter<c64, c64, float>>(
                { { "interpolation", size_t{ 4 } }, { "taps", firdes::root_raised_cosine(1.0, 4.0, 1.0, 0.35, 44) } });
            run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
        } else if (what == "arb_float" || what == "arb_double") {
            if (what == "arb_float") {
                auto& b = fg.emplaceBlock<PfbArbResampler<c64, c64, float, float>>(
                    { { "rate", 1.0f + 1.2e-6f }, { "taps", pfb_arb_taps } });
                run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
            } else {
                auto& b = fg.emplaceBlock<PfbArbResampler<c64, c64, float, double>>(
                    { { "rate", 1.1234 }, { "taps", pfb_arb_taps } });
                run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
            }
        } else if (what == "syncword_detection" || what == "syncword_detection_1bin") {
            // the reference's detector (start(): templates; processBulk(): overlap-save correlation, the best-bin / median
            // scan over its mutable history, output_tag) on the oracle's FFT (gnuradio-4.0/algorithm/fourier/fftw.hpp)
            const std::vector<uint8_t> sw = { 0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1,
                                              1, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0,
                                              1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0 }; // tests/_signals.py SYNCWORD
            auto rrc = firdes::root_raised_cosine(1.0, 4.0, 1.0, 0.35, 44);
            float norm = 0.0f;
            for (float t : rrc) norm += t * t;
            norm = std::sqrt(norm);
            for (float& t : rrc) t /= norm;
            const int edge = what == "syncword_detection" ? 4 : 0;
            auto& b = fg.emplaceBlock<SyncwordDetection>(
                { { "rrc_taps", rrc }, { "syncword", sw },
                  { "constellation", std::vector<std::complex<float>>{ { 1.0f, 0.0f }, { -1.0f, 0.0f } } },
                  { "min_freq_bin", -edge }, { "max_freq_bin", edge } });
            b.start();
            run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
        } else if (what == "wipeoff") {
            std::vector<float> sw(64);
            for (size_t i = 0; i < 64; ++i) sw[i] = (i * 7 % 3) ? -1.0f : 1.0f;
            auto& b = fg.emplaceBlock<SyncwordWipeoff<>>({ { "syncword", sw } });
            run(b, in, out, chunk, [&](auto& is, auto& os) { return b.processBulk(is, os); });
        } else {
            return 2;
        }
        dump(prefix + ".out.c64", out.data.data(), out.size);
        dump_tags(prefix + ".out_tags.bin", out.tags);
        const uint64_t counts[2] = { in.rd, out.size };
        dump(prefix + ".counts.bin", counts, 2);
        std::printf("%s: consumed %zu produced %zu tags %zu\n", what.c_str(), in.rd, out.size, out.tags.size());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ref_headers_check: %s\n", e.what());
        return 1;
    }
}
