// gr4pm_gr4_blocks.hpp -- GNU Radio 4.0 block wrappers over the C ABI (include/gr4pm_hip.h).
//
// Same namespace-level names, ports, settings and tag keys as the reference blocks, so a
// flowgraph written against
//   <gnuradio-4.0/packet-modem/syncword_detection.hpp> etc.
// links against these instead by switching the include.  The classes live
// in gr::packet_modem::hip to be able to coexist with the CPU blocks in one binary; add
// `namespace gr::packet_modem { using hip::SyncwordDetection; }` for a pure drop-in.
//
// gnuradio4 is NOT part of this repository's image (the reference's submodule is empty), so
// this header is compile-checked only where gnuradio4 is installed.  It touches exactly the
// GR4 surface the reference blocks touch: gr::Block<D>, PortIn/PortOut,
// ConsumableSpan/PublishableSpan (size, begin, consume, publish), input_tags_present(),
// mergedInputTag(), publishTag(), gr::exception, ENABLE_REFLECTION.
//
// Staging: GR4 port buffers are host memory; each wrapper owns a device input and output
// buffer and copies through the handle's stream.  Chains that should stay in HBM use the
// C ABI directly with device rings (bench.py does).
#pragma once
#include <gnuradio-4.0/Block.hpp>
#include <gnuradio-4.0/reflection.hpp>
#include <hip/hip_runtime.h>

#include <complex>
#include <string>
#include <vector>

#include "gr4pm_hip.h"

namespace gr::packet_modem::hip {

namespace detail {
inline void check(gr4pm_status s, const char* what)
{
    if (s < 0) throw gr::exception(std::string(what) + ": " + gr4pm_last_error());
}
// device staging buffer that grows on demand
template <typename T>
struct DeviceStage {
    T* p = nullptr;
    size_t n = 0;
    ~DeviceStage() { if (p) (void)hipFree(p); }
    T* get(size_t count)
    {
        if (count > n) {
            if (p) (void)hipFree(p);
            if (hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)) != hipSuccess)
                throw gr::exception("hipMalloc failed");
            n = count;
        }
        return p;
    }
};
inline gr::property_map to_map(const gr4pm_tag& t)
{
    // syncword_detection.hpp:106-114
    return { { "syncword_amplitude", t.amplitude }, { "syncword_phase", t.phase },
             { "syncword_freq", t.freq },           { "syncword_freq_bin", t.freq_bin },
             { "syncword_noise_power", t.noise_power }, { "syncword_esn0_db", t.esn0_db },
             { "syncword_time_est", t.time_est } };
}
inline gr4pm_tag from_map(const gr::property_map& m, uint64_t index)
{
    gr4pm_tag t{};
    t.index = index;
    if (m.contains("syncword_amplitude")) {
        t.flags |= GR4PM_TAG_SYNCWORD;
        t.amplitude = pmtv::cast<float>(m.at("syncword_amplitude"));
        if (m.contains("syncword_phase")) t.phase = pmtv::cast<float>(m.at("syncword_phase"));
        if (m.contains("syncword_freq")) t.freq = pmtv::cast<double>(m.at("syncword_freq"));
        if (m.contains("syncword_time_est")) t.time_est = pmtv::cast<float>(m.at("syncword_time_est"));
    }
    for (const auto& [k, v] : m)
        if (!k.starts_with("syncword_")) t.flags |= GR4PM_TAG_OTHER;
    return t;
}
} // namespace detail

// ---------------------------------------------------------------- SyncwordDetection
// replaces gr::packet_modem::SyncwordDetection (syncword_detection.hpp:32-357)
class SyncwordDetection : public gr::Block<SyncwordDetection>
{
    using c64 = std::complex<float>;
    gr4pm_syncword_detection* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;
    std::vector<gr4pm_tag> _tags;

public:
    size_t _syncword_samples_size = 0; // read by tests/apps (qa_syncword_detection.cpp:133)
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t fft_size = 2048;
    size_t samples_per_symbol = 4;
    std::vector<float> rrc_taps;
    std::vector<uint8_t> syncword;
    std::vector<c64> constellation;
    int min_freq_bin = 0;
    int max_freq_bin = 0;
    uint64_t time_threshold = 768;
    float power_threshold = 9.5;

    ~SyncwordDetection() { gr4pm_syncword_detection_destroy(_h); }

    void start()
    {
        gr4pm_syncword_detection_destroy(_h);
        _h = nullptr;
        gr4pm_syncword_detection_params p{};
        p.fft_size = fft_size;
        p.samples_per_symbol = samples_per_symbol;
        p.rrc_taps = rrc_taps.data();
        p.n_rrc_taps = rrc_taps.size();
        p.syncword = syncword.data();
        p.n_syncword = syncword.size();
        p.constellation = reinterpret_cast<const gr4pm_c64*>(constellation.data());
        p.n_constellation = constellation.size();
        p.min_freq_bin = min_freq_bin;
        p.max_freq_bin = max_freq_bin;
        p.time_threshold = time_threshold;
        p.power_threshold = power_threshold;
        p.n_channels = 1;
        p.max_items = size_t{ 1 } << 22;
        detail::check(gr4pm_syncword_detection_create(&p, &_h), "SyncwordDetection::start");
        _syncword_samples_size = gr4pm_syncword_detection_syncword_samples_size(_h);
        in.min_samples = fft_size; // syncword_detection.hpp:200-201
        out.min_samples = fft_size;
        _tags.resize(4096);
    }

    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        if (inSpan.size() < fft_size) { // :215-227
            if (!inSpan.consume(0)) throw gr::exception("consume failed");
            outSpan.publish(0);
            return gr::work::Status::INSUFFICIENT_INPUT_ITEMS;
        }
        const size_t n = std::min<size_t>(inSpan.size(), size_t{ 1 } << 22);
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        if (hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice) != hipSuccess)
            throw gr::exception("hipMemcpy H2D failed");
        size_t n_done = 0, n_tags = 0;
        detail::check(gr4pm_syncword_detection_process(_h, din, n, n, dout, n, &n_done, _tags.data(),
                                                       _tags.size(), &n_tags),
                      "SyncwordDetection::processBulk");
        if (hipMemcpy(&*outSpan.begin(), dout, n_done * sizeof(c64), hipMemcpyDeviceToHost) != hipSuccess)
            throw gr::exception("hipMemcpy D2H failed");
        for (size_t i = 0; i < n_tags; ++i)
            out.publishTag(detail::to_map(_tags[i]), static_cast<ssize_t>(_tags[i].index)); // :321-324
        if (!inSpan.consume(n_done)) throw gr::exception("consume failed"); // :346-348
        outSpan.publish(n_done);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- rotators
// replaces gr::packet_modem::Rotator<float> (rotator.hpp:20-65)
class Rotator : public gr::Block<Rotator>
{
    using c64 = std::complex<float>;
    gr4pm_rotator* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    float phase_incr = 0;
    ~Rotator() { gr4pm_rotator_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&) { start(); }
    void start()
    {
        gr4pm_rotator_destroy(_h);
        gr4pm_rotator_params p{ 0, phase_incr, 0, 1, nullptr };
        detail::check(gr4pm_rotator_create(&p, &_h), "Rotator::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        const size_t n = std::min(inSpan.size(), outSpan.size());
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        detail::check(gr4pm_rotator_process(_h, din, n, n, dout, nullptr, nullptr, 0), "Rotator");
        (void)hipMemcpy(&*outSpan.begin(), dout, n * sizeof(c64), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        return gr::work::Status::OK;
    }
};

// replaces gr::packet_modem::CoarseFrequencyCorrection<float> (coarse_frequency_correction.hpp:20-99)
class CoarseFrequencyCorrection : public gr::Block<CoarseFrequencyCorrection>
{
    using c64 = std::complex<float>;
    gr4pm_rotator* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t delay = 0;
    ~CoarseFrequencyCorrection() { gr4pm_rotator_destroy(_h); }
    void start()
    {
        gr4pm_rotator_destroy(_h);
        gr4pm_rotator_params p{ 1, 0.0f, delay, 1, nullptr };
        detail::check(gr4pm_rotator_create(&p, &_h), "CoarseFrequencyCorrection::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) { // :76-82: the tag refers to inSpan[0]
            tag = detail::from_map(this->mergedInputTag().map, 0);
            if (this->mergedInputTag().map.contains("syncword_freq")) {
                tag.flags |= GR4PM_TAG_SYNCWORD;
                n_tags = 1;
            }
        }
        const size_t n = std::min(inSpan.size(), outSpan.size());
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        detail::check(gr4pm_rotator_process(_h, din, n, n, dout, &tag, nullptr, n_tags), "CoarseFrequencyCorrection");
        (void)hipMemcpy(&*outSpan.begin(), dout, n * sizeof(c64), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        return gr::work::Status::OK;
    }
};

// The remaining wrappers (SyncwordDetectionFilter, SymbolFilter, CostasLoop, SyncwordWipeoff,
// InterpolatingFirFilter, PfbArbResampler) follow the same three steps -- head tag -> gr4pm_tag
// at index 0, stage, call gr4pm_<block>_process, publish re-timed tags -- and are listed with
// their exact settings in INTEGRATION.md.

} // namespace gr::packet_modem::hip

ENABLE_REFLECTION(gr::packet_modem::hip::SyncwordDetection, in, out, fft_size, samples_per_symbol, rrc_taps,
                  syncword, constellation, min_freq_bin, max_freq_bin, time_threshold, power_threshold);
ENABLE_REFLECTION(gr::packet_modem::hip::Rotator, in, out, phase_incr);
ENABLE_REFLECTION(gr::packet_modem::hip::CoarseFrequencyCorrection, in, out, delay);
