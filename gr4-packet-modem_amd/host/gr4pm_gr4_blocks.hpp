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

#include <cctype>
#include <complex>
#include <string>
#include <type_traits>
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

// ---------------------------------------------------------------- SyncwordDetectionFilter
// replaces gr::packet_modem::SyncwordDetectionFilter<c64> (syncword_detection_filter.hpp:10-211)
class SyncwordDetectionFilter : public gr::Block<SyncwordDetectionFilter>
{
    using c64 = std::complex<float>;
    gr4pm_syncword_detection_filter* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<gr::Message, gr::Async> parsed_header;
    gr::PortIn<gr::Message, gr::Async> ignored_syncword;
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t samples_per_symbol = 4;
    size_t syncword_size = 64;
    size_t header_size = 128;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ~SyncwordDetectionFilter() { gr4pm_syncword_detection_filter_destroy(_h); }
    void start()
    {
        gr4pm_syncword_detection_filter_destroy(_h);
        gr4pm_syncword_detection_filter_params p{ samples_per_symbol, syncword_size, header_size, nullptr };
        detail::check(gr4pm_syncword_detection_filter_create(&p, &_h), "SyncwordDetectionFilter::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& headerSpan,
                                 const gr::ConsumableSpan auto& ignoredSpan,
                                 const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        int head_flags = 0;
        gr::property_map syncword_keys, other_keys;
        if (this->input_tags_present()) { // :75-93
            for (const auto& [key, val] : this->mergedInputTag().map) {
                if (key.starts_with("syncword_")) {
                    head_flags |= GR4PM_TAG_SYNCWORD;
                    syncword_keys[key] = val;
                } else {
                    head_flags |= GR4PM_TAG_OTHER;
                    other_keys[key] = val;
                }
            }
        }
        std::vector<gr4pm_header_msg> msgs;
        for (size_t i = 0; i < headerSpan.size(); ++i) { // :134-152
            const auto& meta = headerSpan[i].data.value();
            gr4pm_header_msg m{};
            m.invalid_header = meta.contains("invalid_header") ? 1 : 0;
            if (!m.invalid_header) m.packet_length = pmtv::cast<uint64_t>(meta.at("packet_length"));
            msgs.push_back(m);
        }
        const size_t n = std::min(inSpan.size(), outSpan.size());
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        size_t consumed = 0, hc = 0, ic = 0;
        int out_flags = 0;
        detail::check(gr4pm_syncword_detection_filter_process(_h, din, n, dout, n, head_flags, msgs.data(),
                                                              msgs.size(), ignoredSpan.size(), &consumed, &hc,
                                                              &ic, &out_flags),
                      "SyncwordDetectionFilter::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, consumed * sizeof(c64), hipMemcpyDeviceToHost);
        gr::property_map output_tags; // :82-104
        if (out_flags & GR4PM_TAG_SYNCWORD) output_tags.insert(syncword_keys.begin(), syncword_keys.end());
        if (out_flags & GR4PM_TAG_OTHER) output_tags.insert(other_keys.begin(), other_keys.end());
        if (!output_tags.empty()) out.publishTag(output_tags, 0);
        if (!inSpan.consume(consumed)) throw gr::exception("inSpan.consume failed");
        if (!headerSpan.consume(hc)) throw gr::exception("headerSpan.consume failed");
        if (!ignoredSpan.consume(ic)) throw gr::exception("ignoredSpan.consume failed");
        outSpan.publish(consumed);
        this->_mergedInputTag.map.clear(); // :125,202
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SymbolFilter
// replaces gr::packet_modem::SymbolFilter<c64, c64, float> (symbol_filter.hpp:13-253)
class SymbolFilter : public gr::Block<SymbolFilter, gr::Resampling<>>
{
    using c64 = std::complex<float>;
    gr4pm_symbol_filter* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;
    std::vector<gr::property_map> _held; // full maps of queued tags (opaque keys travel with them)
    std::vector<gr4pm_tag> _tags_out;

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t samples_per_symbol = 4;
    std::vector<float> taps;
    size_t num_arms = 32;
    size_t delay = 0;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ~SymbolFilter() { gr4pm_symbol_filter_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        gr4pm_symbol_filter_destroy(_h);
        _h = nullptr;
        gr4pm_symbol_filter_params p{ samples_per_symbol, taps.data(), taps.size(), num_arms, delay, 0, nullptr };
        detail::check(gr4pm_symbol_filter_create(&p, &_h), "SymbolFilter::settingsChanged"); // :67-73 throw
        _tags_out.resize(64);
    }
    void start() { detail::check(gr4pm_symbol_filter_reset(_h), "SymbolFilter::start"); }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) { // :127-206: the tag refers to inSpan[0]
            tag = detail::from_map(this->mergedInputTag().map, 0);
            tag.freq_bin = static_cast<int32_t>(_held.size()); // handle of the full map
            _held.push_back(this->mergedInputTag().map);
            n_tags = 1;
        }
        const size_t n = inSpan.size();
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(outSpan.size());
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        size_t n_out_tags = 0, consumed = 0, produced = 0;
        detail::check(gr4pm_symbol_filter_process(_h, din, n, dout, outSpan.size(), &tag, n_tags, _tags_out.data(),
                                                  _tags_out.size(), &n_out_tags, &consumed, &produced),
                      "SymbolFilter::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, produced * sizeof(c64), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n_out_tags; ++i) { // :218-228 re-timed tags, :152-155 adjusted phase
            auto map = _held.at(static_cast<size_t>(_tags_out[i].freq_bin));
            if (_tags_out[i].flags & GR4PM_TAG_SYNCWORD) map["syncword_phase"] = _tags_out[i].phase;
            out.publishTag(map, static_cast<ssize_t>(_tags_out[i].index));
        }
        if (!inSpan.consume(consumed)) throw gr::exception("consume failed"); // :240-243
        outSpan.publish(produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- CostasLoop
// replaces gr::packet_modem::CostasLoop<float, float> (costas_loop.hpp:15-149)
class CostasLoop : public gr::Block<CostasLoop>
{
    using c64 = std::complex<float>;
    gr4pm_costas_loop* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;
    static int constellation_id(const std::string& s)
    {
        std::string u;
        for (char c : s) u.push_back(static_cast<char>(std::toupper(c)));
        if (u == "PILOT") return 0;
        if (u == "BPSK") return 1;
        if (u == "QPSK") return 2;
        throw gr::exception("unknown constellation " + s); // enum_cast(...).value() throws, :59-61
    }

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    double loop_bandwidth = 0.01;
    std::string constellation = "BPSK";

    ~CostasLoop() { gr4pm_costas_loop_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&) // :52-88 (also driven by tags)
    {
        if (!_h) {
            gr4pm_costas_loop_params p{ loop_bandwidth, constellation_id(constellation), 1, nullptr };
            detail::check(gr4pm_costas_loop_create(&p, &_h), "CostasLoop::settingsChanged");
        } else {
            detail::check(gr4pm_costas_loop_set(_h, loop_bandwidth, constellation_id(constellation)), "CostasLoop::set");
        }
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present() && this->mergedInputTag().map.contains("syncword_phase")) { // :101-106
            tag.index = 0;
            tag.flags = GR4PM_TAG_SYNCWORD;
            tag.phase = pmtv::cast<float>(this->mergedInputTag().map.at("syncword_phase"));
            n_tags = 1;
        }
        const size_t n = std::min(inSpan.size(), outSpan.size());
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        detail::check(gr4pm_costas_loop_process(_h, din, n, n, dout, &tag, nullptr, n_tags), "CostasLoop::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, n * sizeof(c64), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SyncwordWipeoff
// replaces gr::packet_modem::SyncwordWipeoff<c64, float> (syncword_wipeoff.hpp:12-91)
class SyncwordWipeoff : public gr::Block<SyncwordWipeoff>
{
    using c64 = std::complex<float>;
    gr4pm_syncword_wipeoff* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    std::vector<float> syncword;

    ~SyncwordWipeoff() { gr4pm_syncword_wipeoff_destroy(_h); }
    void start()
    {
        gr4pm_syncword_wipeoff_destroy(_h);
        gr4pm_syncword_wipeoff_params p{ syncword.data(), syncword.size(), nullptr };
        detail::check(gr4pm_syncword_wipeoff_create(&p, &_h), "SyncwordWipeoff::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present() && this->mergedInputTag().map.contains("syncword_amplitude")) { // :53-62
            tag.index = 0;
            tag.flags = GR4PM_TAG_SYNCWORD;
            n_tags = 1;
        }
        const size_t n = std::min(inSpan.size(), outSpan.size());
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        detail::check(gr4pm_syncword_wipeoff_process(_h, din, n, dout, &tag, n_tags), "SyncwordWipeoff::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, n * sizeof(c64), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- InterpolatingFirFilter
// replaces gr::packet_modem::InterpolatingFirFilter<c64, c64, float> (interpolating_fir_filter.hpp:14-103)
class InterpolatingFirFilter : public gr::Block<InterpolatingFirFilter, gr::Resampling<>>
{
    using c64 = std::complex<float>;
    gr4pm_interp_fir* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t interpolation = 1;
    std::vector<float> taps;

    ~InterpolatingFirFilter() { gr4pm_interp_fir_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        gr4pm_interp_fir_destroy(_h);
        _h = nullptr;
        gr4pm_interp_fir_params p{ interpolation, taps.data(), taps.size(), 0, nullptr };
        detail::check(gr4pm_interp_fir_create(&p, &_h), "InterpolatingFirFilter::settingsChanged"); // :45-47
        this->input_chunk_size = 1; // :50-51
        this->output_chunk_size = interpolation;
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        const size_t n = std::min(inSpan.size(), outSpan.size() / interpolation); // :91
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n * interpolation);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        detail::check(gr4pm_interp_fir_process(_h, din, n, dout), "InterpolatingFirFilter::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, n * interpolation * sizeof(c64), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n * interpolation);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- PfbArbResampler
// replaces gr::packet_modem::PfbArbResampler<c64, c64, float, TRate> (pfb_arb_resampler.hpp:23-183)
template <typename TRate = float>
class PfbArbResampler : public gr::Block<PfbArbResampler<TRate>>
{
    using c64 = std::complex<float>;
    gr4pm_pfb_arb_resampler* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<c64, gr::Async> in;   // :59-62: no rational resampling ratio
    gr::PortOut<c64, gr::Async> out;
    TRate rate{ 1.0 };
    std::vector<float> taps; // the reference's default (pfb_arb_taps.hpp) ships as data/pfb_arb_taps.f32
    size_t filter_size = 32;

    ~PfbArbResampler() { gr4pm_pfb_arb_resampler_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        gr4pm_pfb_arb_resampler_destroy(_h);
        _h = nullptr;
        gr4pm_pfb_arb_resampler_params p{ static_cast<double>(rate), std::is_same_v<TRate, double> ? 1 : 0,
                                          taps.data(), taps.size(), filter_size, nullptr };
        detail::check(gr4pm_pfb_arb_resampler_create(&p, &_h), "PfbArbResampler::settingsChanged"); // :70-72
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        const size_t n = inSpan.size();
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(outSpan.size());
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        size_t consumed = 0, produced = 0;
        detail::check(gr4pm_pfb_arb_resampler_process(_h, din, n, dout, outSpan.size(), &consumed, &produced),
                      "PfbArbResampler::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, produced * sizeof(c64), hipMemcpyDeviceToHost);
        if (!inSpan.consume(consumed)) throw gr::exception("consume failed"); // :169-172
        outSpan.publish(produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- PayloadMetadataInsert
// replaces gr::packet_modem::PayloadMetadataInsert<c64> (payload_metadata_insert.hpp:12-324)
class PayloadMetadataInsert : public gr::Block<PayloadMetadataInsert>
{
    using c64 = std::complex<float>;
    gr4pm_payload_metadata_insert* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;
    std::vector<gr4pm_packet_tag> _tags;
    gr::property_map _syncword_map; // every key of the syncword tag travels on (:104-112)
    static const char* constellation_name(int c) { return c == 0 ? "PILOT" : c == 1 ? "BPSK" : "QPSK"; }

public:
    gr::PortIn<gr::Message, gr::Async> parsed_header;
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    gr::PortOut<gr::Message, gr::Async> ignored_syncword;
    size_t syncword_size = 64;
    size_t header_size = 128;
    double syncword_costas_loop_bandwidth = 0.02;
    double header_costas_loop_bandwidth = 0.01;
    double payload_costas_loop_bandwidth = 0.005;
    bool log = false;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ~PayloadMetadataInsert() { gr4pm_payload_metadata_insert_destroy(_h); }
    void start() // :71-75
    {
        gr4pm_payload_metadata_insert_destroy(_h);
        gr4pm_payload_metadata_insert_params p{ syncword_size, header_size, syncword_costas_loop_bandwidth,
                                                header_costas_loop_bandwidth, payload_costas_loop_bandwidth, nullptr };
        detail::check(gr4pm_payload_metadata_insert_create(&p, &_h), "PayloadMetadataInsert::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& headerSpan, const gr::ConsumableSpan auto& inSpan,
                                 gr::PublishableSpan auto& outSpan, gr::PublishableSpan auto& ignoredSpan)
    {
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) {
            tag = detail::from_map(this->mergedInputTag().map, 0);
            if (tag.flags & GR4PM_TAG_SYNCWORD) _syncword_map = this->mergedInputTag().map;
            n_tags = 1;
        }
        std::vector<gr4pm_header_msg> msgs;
        std::vector<gr::property_map> metas;
        for (const auto& m : headerSpan) { // :207-242
            const auto& meta = m.data.value();
            gr4pm_header_msg hm{};
            hm.invalid_header = meta.contains("invalid_header") ? 1 : 0;
            if (!hm.invalid_header) hm.packet_length = pmtv::cast<uint64_t>(meta.at("packet_length"));
            msgs.push_back(hm);
            metas.push_back(meta);
        }
        const size_t n = inSpan.size(), cap = outSpan.size();
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(cap);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        _tags.resize(8);
        size_t n_out_tags = 0, consumed = 0, produced = 0, used = 0, ignored = 0;
        detail::check(gr4pm_payload_metadata_insert_process(_h, din, n, dout, cap, &tag, n_tags, msgs.data(),
                                                            msgs.size(), 0, _tags.data(), _tags.size(), &n_out_tags,
                                                            &consumed, &produced, &used, &ignored),
                      "PayloadMetadataInsert::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, produced * sizeof(c64), hipMemcpyDeviceToHost);
        size_t hdr = 0;
        for (size_t i = 0; i < n_out_tags; ++i) {
            const auto& t = _tags[i];
            gr::property_map m;
            if (t.kind == GR4PM_PKT_SYNCWORD) m = _syncword_map;                    // :104-112
            if (t.kind == GR4PM_PKT_HEADER_START) m["header_start"] = pmtv::pmt_null(); // :186-194
            if (t.kind == GR4PM_PKT_PAYLOAD) {                                       // :222-234
                while (hdr < used && metas[hdr].contains("invalid_header")) ++hdr;
                m = metas[hdr++];
                m["payload_symbols"] = pmtv::pmt(t.payload_symbols);
                m["payload_bits"] = pmtv::pmt(t.payload_bits);
            }
            if (t.constellation >= 0) m["constellation"] = std::string(constellation_name(t.constellation));
            if (t.loop_bandwidth >= 0) m["loop_bandwidth"] = t.loop_bandwidth;
            out.publishTag(m, static_cast<ssize_t>(t.index));
        }
        size_t ignored_published = 0;
        if (log && ignored > 0 && ignoredSpan.size() > 0) { // :126-147
            ignoredSpan[0] = {};
            ignored_published = 1;
        }
        if (!headerSpan.consume(used)) throw gr::exception("consume failed");
        if (!inSpan.consume(consumed)) throw gr::exception("consume failed");
        ignoredSpan.publish(ignored_published);
        outSpan.publish(produced);
        if (consumed != 0) this->_mergedInputTag.map.clear(); // :288-295
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SyncwordRemove
// replaces gr::packet_modem::SyncwordRemove<c64> (syncword_remove.hpp:11-112)
class SyncwordRemove : public gr::Block<SyncwordRemove>
{
    using c64 = std::complex<float>;
    gr4pm_syncword_remove* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t syncword_size = 64;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ~SyncwordRemove() { gr4pm_syncword_remove_destroy(_h); }
    void start()
    {
        gr4pm_syncword_remove_destroy(_h);
        gr4pm_syncword_remove_params p{ syncword_size, nullptr };
        detail::check(gr4pm_syncword_remove_create(&p, &_h), "SyncwordRemove::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        gr4pm_packet_tag tag{}, tout[2];
        size_t n_tags = 0;
        if (this->input_tags_present()) { // :51-64
            tag.index = 0;
            tag.kind = this->mergedInputTag().map.contains("syncword_amplitude") ? GR4PM_PKT_SYNCWORD
                                                                                 : GR4PM_PKT_HEADER_START;
            tag.constellation = -1;
            tag.loop_bandwidth = -1.0;
            n_tags = 1;
        }
        const size_t n = std::min(inSpan.size(), outSpan.size());
        gr4pm_c64* din = _din.get(n);
        gr4pm_c64* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        size_t n_out_tags = 0, produced = 0;
        detail::check(gr4pm_syncword_remove_process(_h, din, n, dout, &tag, n_tags, tout, 2, &n_out_tags, &produced),
                      "SyncwordRemove::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, produced * sizeof(c64), hipMemcpyDeviceToHost);
        if (n_out_tags) out.publishTag(this->mergedInputTag().map, static_cast<ssize_t>(tout[0].index)); // :59-62
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(produced);
        if (n != 0) this->_mergedInputTag.map.clear(); // :95-102
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- ConstellationLLRDecoder
// replaces gr::packet_modem::ConstellationLLRDecoder<float> (constellation_llr_decoder.hpp:13-142)
class ConstellationLLRDecoder : public gr::Block<ConstellationLLRDecoder, gr::Resampling<>>
{
    using c64 = std::complex<float>;
    gr4pm_constellation_llr_decoder* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din;
    detail::DeviceStage<float> _dout;

public:
    gr::PortIn<c64> in;
    gr::PortOut<float> out;
    float noise_sigma = 1.0f;
    std::string constellation = "BPSK";
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ~ConstellationLLRDecoder() { gr4pm_constellation_llr_decoder_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&) // :55-78 (also driven by tags)
    {
        std::string u;
        for (char c : constellation) u.push_back(static_cast<char>(std::toupper(c)));
        const int id = u == "BPSK" ? 1 : u == "QPSK" ? 2 : 0;
        if (id == 0) throw gr::exception("constellation " + constellation + " not supported"); // :72-74
        this->input_chunk_size = 1;
        this->output_chunk_size = static_cast<size_t>(id); // :64-71
        gr4pm_constellation_llr_decoder_destroy(_h);
        _h = nullptr;
        gr4pm_constellation_llr_decoder_params p{ noise_sigma, id, nullptr };
        detail::check(gr4pm_constellation_llr_decoder_create(&p, &_h), "ConstellationLLRDecoder::settingsChanged");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        if (this->input_tags_present()) out.publishTag(this->mergedInputTag().map, 0); // :93-99
        const size_t n = std::min(inSpan.size(), outSpan.size() / this->output_chunk_size);
        gr4pm_c64* din = _din.get(n);
        float* dout = _dout.get(2 * n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(c64), hipMemcpyHostToDevice);
        size_t produced = 0;
        detail::check(gr4pm_constellation_llr_decoder_process(_h, din, n, dout, 2 * n, nullptr, 0, nullptr, 0, nullptr,
                                                              &produced),
                      "ConstellationLLRDecoder::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, produced * sizeof(float), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- AdditiveScrambler
// replaces gr::packet_modem::AdditiveScrambler<float> / <uint8_t> (additive_scrambler.hpp:24-100)
template <typename T>
class AdditiveScrambler : public gr::Block<AdditiveScrambler<T>>
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, uint8_t>);
    gr4pm_additive_scrambler* _h = nullptr;
    detail::DeviceStage<T> _din, _dout;

public:
    gr::PortIn<T> in;
    gr::PortOut<T> out;
    uint64_t mask = 0x8a, seed = 0x7f, length = 7, count = 0; // :61-64
    std::string reset_tag_key = "";

    ~AdditiveScrambler() { gr4pm_additive_scrambler_destroy(_h); }
    void start() // :68
    {
        gr4pm_additive_scrambler_destroy(_h);
        gr4pm_additive_scrambler_params p{ mask, seed, length, count, std::is_same_v<T, float> ? 1 : 2, nullptr };
        detail::check(gr4pm_additive_scrambler_create(&p, &_h), "AdditiveScrambler::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        const uint64_t zero = 0;
        const bool reset = !reset_tag_key.empty() && this->input_tags_present() &&
                           this->mergedInputTag().map.contains(reset_tag_key); // :78-80
        const size_t n = std::min(inSpan.size(), outSpan.size());
        T* din = _din.get(n);
        T* dout = _dout.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(T), hipMemcpyHostToDevice);
        detail::check(gr4pm_additive_scrambler_process(_h, din, n, dout, &zero, reset ? 1 : 0),
                      "AdditiveScrambler::processBulk");
        (void)hipMemcpy(&*outSpan.begin(), dout, n * sizeof(T), hipMemcpyDeviceToHost);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- HeaderPayloadSplit
// replaces gr::packet_modem::HeaderPayloadSplit<float> (header_payload_split.hpp:9-147)
class HeaderPayloadSplit : public gr::Block<HeaderPayloadSplit>
{
    gr4pm_header_payload_split* _h = nullptr;
    detail::DeviceStage<float> _din, _dhdr, _dpay;

public:
    gr::PortIn<float> in;
    gr::PortOut<float> header;
    gr::PortOut<float> payload;
    size_t header_size = 256;
    std::string packet_len_tag_key = "packet_len";
    std::string payload_length_key = "payload_bits";
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ~HeaderPayloadSplit() { gr4pm_header_payload_split_destroy(_h); }
    void start() // :41-45
    {
        gr4pm_header_payload_split_destroy(_h);
        gr4pm_header_payload_split_params p{ header_size, nullptr };
        detail::check(gr4pm_header_payload_split_create(&p, &_h), "HeaderPayloadSplit::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& headerSpan,
                                 gr::PublishableSpan auto& payloadSpan)
    {
        gr4pm_packet_tag tag{}, ht[2], pt[2];
        size_t n_tags = 0;
        gr::property_map map;
        if (this->input_tags_present()) { // :68-88
            map = this->mergedInputTag().map;
            tag.kind = GR4PM_PKT_HEADER_START;
            tag.constellation = -1;
            tag.loop_bandwidth = -1.0;
            if (map.contains(payload_length_key)) {
                tag.kind = GR4PM_PKT_PAYLOAD;
                tag.payload_bits = pmtv::cast<uint64_t>(map.at(payload_length_key));
                map[packet_len_tag_key] = pmtv::pmt(tag.payload_bits); // :81
            }
            n_tags = 1;
        }
        // one output per call, like the reference (:97-123)
        const size_t n = std::min({ inSpan.size(), headerSpan.size(), payloadSpan.size() });
        float* din = _din.get(n);
        float* dh = _dhdr.get(n);
        float* dp = _dpay.get(n);
        (void)hipMemcpy(din, &*inSpan.begin(), n * sizeof(float), hipMemcpyHostToDevice);
        size_t nh = 0, np = 0, nht = 0, npt = 0;
        detail::check(gr4pm_header_payload_split_process(_h, din, n, dh, &nh, dp, &np, &tag, n_tags, ht, &nht, pt, &npt, 2),
                      "HeaderPayloadSplit::processBulk"); // the unexpected-tag exception of :75-78 included
        (void)hipMemcpy(&*headerSpan.begin(), dh, nh * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&*payloadSpan.begin(), dp, np * sizeof(float), hipMemcpyDeviceToHost);
        if (nht) header.publishTag(map, 0);
        if (npt) payload.publishTag(map, 0);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        headerSpan.publish(nh);
        payloadSpan.publish(np);
        this->_mergedInputTag.map.clear(); // :125-131
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- HeaderFecDecoder
// replaces gr::packet_modem::HeaderFecDecoder (header_fec_decoder.hpp:13-359) and with it the
// reference's calls into ldpc-toolbox (:276,285,315-321)
class HeaderFecDecoder : public gr::Block<HeaderFecDecoder, gr::Resampling<1U, 64U, true>>
{
    gr4pm_header_fec_decoder* _h = nullptr;
    detail::DeviceStage<float> _din;
    std::vector<uint8_t> _bytes, _invalid;

public:
    gr::PortIn<float> in;
    gr::PortOut<uint8_t> out;
    std::string alist; // the text of header_fec_decoder.hpp:31-258 (data/header_ldpc_128_32.alist)

    ~HeaderFecDecoder() { gr4pm_header_fec_decoder_destroy(_h); }
    void start() // :268-280
    {
        if (_h) throw gr::exception("an LDPC decoder already exists");
        gr4pm_header_fec_decoder_params p{ alist.c_str(), 25, nullptr };
        detail::check(gr4pm_header_fec_decoder_create(&p, &_h), "HeaderFecDecoder::start");
    }
    void stop() // :282-288
    {
        gr4pm_header_fec_decoder_destroy(_h);
        _h = nullptr;
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        const size_t codewords = std::min(inSpan.size() / 256, outSpan.size() / 4); // :293-294
        if (codewords == 0) { // :296-304
            std::ignore = inSpan.consume(0);
            outSpan.publish(0);
            return inSpan.size() < 256 ? gr::work::Status::INSUFFICIENT_INPUT_ITEMS
                                       : gr::work::Status::INSUFFICIENT_OUTPUT_ITEMS;
        }
        float* din = _din.get(codewords * 256);
        (void)hipMemcpy(din, &*inSpan.begin(), codewords * 256 * sizeof(float), hipMemcpyHostToDevice);
        _bytes.resize(codewords * 4);
        _invalid.resize(codewords);
        detail::check(gr4pm_header_fec_decoder_process(_h, din, codewords, _bytes.data(), _invalid.data()),
                      "HeaderFecDecoder::processBulk");
        std::copy(_bytes.begin(), _bytes.end(), outSpan.begin());
        for (size_t c = 0; c < codewords; ++c)
            if (_invalid[c]) out.publishTag({ { "invalid_header", pmtv::pmt_null() } }, static_cast<ssize_t>(4 * c)); // :322-326
        if (!inSpan.consume(codewords * 256)) throw gr::exception("consume failed");
        outSpan.publish(codewords * 4);
        return gr::work::Status::OK;
    }
};

} // namespace gr::packet_modem::hip

ENABLE_REFLECTION(gr::packet_modem::hip::SyncwordDetection, in, out, fft_size, samples_per_symbol, rrc_taps,
                  syncword, constellation, min_freq_bin, max_freq_bin, time_threshold, power_threshold);
ENABLE_REFLECTION(gr::packet_modem::hip::Rotator, in, out, phase_incr);
ENABLE_REFLECTION(gr::packet_modem::hip::CoarseFrequencyCorrection, in, out, delay);
ENABLE_REFLECTION(gr::packet_modem::hip::SyncwordDetectionFilter, parsed_header, ignored_syncword, in, out,
                  samples_per_symbol, syncword_size, header_size);
ENABLE_REFLECTION(gr::packet_modem::hip::SymbolFilter, in, out, samples_per_symbol, taps, num_arms, delay);
ENABLE_REFLECTION(gr::packet_modem::hip::CostasLoop, in, out, loop_bandwidth, constellation);
ENABLE_REFLECTION(gr::packet_modem::hip::SyncwordWipeoff, in, out, syncword);
ENABLE_REFLECTION(gr::packet_modem::hip::InterpolatingFirFilter, in, out, interpolation, taps);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::PfbArbResampler, in, out, rate, taps, filter_size);
ENABLE_REFLECTION(gr::packet_modem::hip::PayloadMetadataInsert, parsed_header, in, out, ignored_syncword,
                  syncword_size, header_size, syncword_costas_loop_bandwidth, header_costas_loop_bandwidth,
                  payload_costas_loop_bandwidth, log);
ENABLE_REFLECTION(gr::packet_modem::hip::SyncwordRemove, in, out, syncword_size);
ENABLE_REFLECTION(gr::packet_modem::hip::ConstellationLLRDecoder, in, out, noise_sigma, constellation);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::AdditiveScrambler, in, out, mask, seed, length, count,
                               reset_tag_key);
ENABLE_REFLECTION(gr::packet_modem::hip::HeaderPayloadSplit, in, header, payload, header_size, packet_len_tag_key,
                  payload_length_key);
ENABLE_REFLECTION(gr::packet_modem::hip::HeaderFecDecoder, in, out, alist);
