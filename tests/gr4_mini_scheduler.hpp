// Miniature single-threaded scheduler shared by tests/gr4_blocks_driver.cpp (the HIP wrappers) and
// tests/ref_headers_check.cpp (the reference's own headers): stream edges with tags, one processBulk() per step with
// the chunk cut at the next tag (a tag is only ever seen at the head of a chunk), default tag forwarding, binary dumps.
// Include AFTER the block headers (it needs gr::Tag, gr::InSpan / gr::OutSpan of the API stand-in).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <numeric>

using c64 = std::complex<float>;

// one stream edge: the "ring" (never reallocated: downstream spans alias it, like GR4's buffers) + its tags
template <typename T>
struct Edge {
    std::vector<T> data;
    size_t size = 0, rd = 0;
    std::vector<gr::Tag> tags; // absolute item index
    explicit Edge(size_t cap) : data(cap) {}
    const gr::Tag* tag_at(size_t idx) const
    {
        for (const auto& t : tags)
            if (static_cast<size_t>(t.index) == idx) return &t;
        return nullptr;
    }
    size_t next_tag_after(size_t idx) const
    {
        size_t best = static_cast<size_t>(-1);
        for (const auto& t : tags)
            if (static_cast<size_t>(t.index) > idx) best = std::min(best, static_cast<size_t>(t.index));
        return best;
    }
};

struct TagRecord { // what the Python side reads back
    uint64_t index;
    float amplitude, phase;
    double freq;
    int32_t freq_bin;
    float noise_power, esn0_db, time_est;
    int32_t has_syncword;
};
static TagRecord record(const gr::Tag& t)
{
    TagRecord r{};
    r.index = static_cast<uint64_t>(t.index);
    const auto& m = t.map;
    r.has_syncword = m.contains("syncword_amplitude") ? 1 : 0;
    if (r.has_syncword) {
        r.amplitude = pmtv::cast<float>(m.at("syncword_amplitude"));
        r.phase = pmtv::cast<float>(m.at("syncword_phase"));
        r.freq = pmtv::cast<double>(m.at("syncword_freq"));
        r.freq_bin = pmtv::cast<int32_t>(m.at("syncword_freq_bin"));
        r.noise_power = pmtv::cast<float>(m.at("syncword_noise_power"));
        r.esn0_db = pmtv::cast<float>(m.at("syncword_esn0_db"));
        r.time_est = pmtv::cast<float>(m.at("syncword_time_est"));
    }
    return r;
}
template <typename T>
static void dump(const std::string& path, const T* p, size_t n)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("cannot write " + path);
    std::fclose(f);
}
static void dump_tags(const std::string& path, const std::vector<gr::Tag>& tags)
{
    std::vector<TagRecord> r;
    for (const auto& t : tags) r.push_back(record(t));
    dump(path, r.data(), r.size());
}

// runs `blk` over everything `in` holds; `call(inSpan, outSpan)` forwards to processBulk (extra message
// spans are bound by the caller).  Returns when the block makes no more progress.
// one processBulk() call; returns whether the block made progress
template <typename Blk, typename TI, typename TO, typename Call>
static bool step(Blk& blk, Edge<TI>& in, Edge<TO>& out, size_t max_chunk, Call call)
{
    {
        const size_t start = in.rd;
        if (start >= in.size) return false;
        const size_t end = std::min({ in.size, start + max_chunk, in.next_tag_after(start) });
        blk._mergedInputTag = {};
        if (const gr::Tag* t = in.tag_at(start)) blk._mergedInputTag = { 0, t->map };
        // input_chunk_size : output_chunk_size is the ratio of the two spans (interpolating_fir_filter.hpp:50-51,91
        // asserts it)
        const size_t ics = std::max<size_t>(blk.input_chunk_size, 1), ocs = std::max<size_t>(blk.output_chunk_size, 1);
        size_t n_in = end - start, n_out = out.data.size() - out.size;
        // synchronous ports: the spans are sized to each other (syncword_wipeoff.hpp:50 asserts equal sizes); only a
        // block whose ports are gr::Async (pfb_arb_resampler.hpp:61-62) sees them independently
        constexpr bool async = std::decay_t<decltype(blk.in)>::is_async;
        if (!async) {
            n_in = std::min(n_in / ics, n_out / ocs) * ics;
            n_out = n_in / ics * ocs;
            if (n_in == 0) return false;
        }
        gr::InSpan<TI> is(in.data.data() + start, n_in);
        gr::OutSpan<TO> os(out.data.data() + out.size, n_out);
        blk.out.published_tags.clear();
        const auto st = call(is, os);
        // a block that calls neither consume() nor publish() is a 1:1 block that has used its whole input span
        // (the runtime then consumes / publishes the spans as given: coarse_frequency_correction.hpp:66-98,
        // costas_loop.hpp:92-147); calling only one of the two is an error
        if (is.consume_called != os.publish_called) throw std::runtime_error("processBulk called only one of consume / publish");
        if (!is.consume_called) {
            if (os.size() < is.size() / ics * ocs) throw std::runtime_error("block without room for its output");
            is.consumed = is.size();
            os.published = is.size() / ics * ocs;
        }
        // blocks without a custom policy get their input tag forwarded by the runtime (the default
        // TagPropagationPolicy: CoarseFrequencyCorrection, SyncwordWipeoff, CostasLoop rely on it)
        constexpr bool custom = requires { Blk::tag_policy; };
        if constexpr (!custom)
            if (blk.input_tags_present() && (is.consumed > 0 || os.published > 0))
                out.tags.push_back({ static_cast<ssize_t>(out.size), blk._mergedInputTag.map });
        for (const auto& t : blk.out.published_tags)
            out.tags.push_back({ static_cast<ssize_t>(out.size) + t.index, t.map });
        in.rd += is.consumed;
        out.size += os.published;
        return st == gr::work::Status::OK && (is.consumed != 0 || os.published != 0);
    }
}
template <typename Blk, typename TI, typename TO, typename Call>
static void run(Blk& blk, Edge<TI>& in, Edge<TO>& out, size_t max_chunk, Call call)
{
    for (int guard = 0; guard < 1000000; ++guard)
        if (!step(blk, in, out, max_chunk, call)) break;
}

static std::vector<c64> read_c64(const char* path)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot read ") + path);
    std::fseek(f, 0, SEEK_END);
    const size_t n = static_cast<size_t>(std::ftell(f)) / sizeof(c64);
    std::fseek(f, 0, SEEK_SET);
    std::vector<c64> x(n);
    if (std::fread(x.data(), sizeof(c64), n, f) != n) throw std::runtime_error("short read");
    std::fclose(f);
    return x;
}

