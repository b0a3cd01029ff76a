// tests/hostlogic/hostlogic_san.cpp -- the library's HIP-free host logic (csrc/hostlogic/*.hpp: the very code the .hip
// files include) under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer, checked against the CPU oracle
// on randomised streams in the style of tools/fuzz_*.py (tags at ragged distances, random messages, random call
// boundaries).  What a run table or a span table means is applied here with plain loops -- the kernels' job on the GPU
// -- so that items and tags can be compared with the oracle's bit for bit.
//   hostlogic_san.<san>.bin [cases] [seed]
#include <atomic>
#include <cassert>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gr4pm_oracle.h"
#include "hostlogic/packet_control.hpp"
#include "hostlogic/sdf_gate.hpp"
#include "hostlogic/slot_queue.hpp"
#include "hostlogic/symbol_filter_replay.hpp"
#include "hostlogic/tail_plan.hpp"
#include "hostlogic/zmtp_pub.hpp"

namespace gr4pm {
static thread_local char g_error[512];
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
} // namespace gr4pm
using namespace gr4pm::hostlogic;

static_assert(sizeof(orc_tag) == sizeof(gr4pm_tag) && sizeof(orc_ptag) == sizeof(gr4pm_packet_tag) &&
                  sizeof(orc_c64) == sizeof(gr4pm_c64),
              "the oracle's records and the ABI's have one layout");
using c64 = orc_c64;
static int g_failures = 0;
#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) {                                                     \
            ++g_failures;                                                  \
            fprintf(stderr, "FAIL %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                                  \
            fprintf(stderr, "\n");                                         \
        }                                                                  \
    } while (0)

static bool same_bits(const c64* a, const c64* b, size_t n) { return n == 0 || std::memcmp(a, b, n * sizeof(c64)) == 0; }
// field by field (the records have tail padding, which nobody defines)
template <typename T>
static bool bits_eq(const T& a, const T& b) { return std::memcmp(&a, &b, sizeof(T)) == 0; }
static bool same_tag(const gr4pm_tag& a, const gr4pm_tag& b)
{
    return a.index == b.index && bits_eq(a.amplitude, b.amplitude) && bits_eq(a.phase, b.phase) && bits_eq(a.freq, b.freq) &&
           a.freq_bin == b.freq_bin && bits_eq(a.noise_power, b.noise_power) && bits_eq(a.esn0_db, b.esn0_db) &&
           bits_eq(a.time_est, b.time_est) && a.flags == b.flags;
}
static bool same_tags(const gr4pm_tag* a, const gr4pm_tag* b, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!same_tag(a[i], b[i])) return false;
    return true;
}
static bool same_ptags(const gr4pm_packet_tag* a, const gr4pm_packet_tag* b, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(a[i].index == b[i].index && a[i].kind == b[i].kind && a[i].constellation == b[i].constellation &&
              bits_eq(a[i].loop_bandwidth, b[i].loop_bandwidth) && a[i].packet_length == b[i].packet_length &&
              a[i].payload_symbols == b[i].payload_symbols && a[i].payload_bits == b[i].payload_bits &&
              (a[i].kind != GR4PM_PKT_SYNCWORD || same_tag(a[i].syncword, b[i].syncword))))
            return false;
    return true;
}
static std::vector<c64> noise(std::mt19937_64& rng, size_t n)
{
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<c64> x(n);
    for (auto& v : x) v = { g(rng), g(rng) };
    return x;
}
static void apply(const std::vector<CopySpan>& spans, const c64* in, c64* out)
{
    for (const auto& s : spans) std::memcpy(out + s.dst, in + s.src, s.len * sizeof(c64));
}

// ---------------------------------------------------------------- SyncwordDetectionFilter, call by call
static void sdf_calls(std::mt19937_64& rng)
{
    SdfState h;
    orc_sdf* ref = orc_sdf_create(h.sps, h.syncword_size, h.header_size);
    const auto x = noise(rng, 40000);
    size_t pos = 0;
    std::vector<gr4pm_header_msg> hdrs;
    size_t n_ignored = 0;
    while (pos < x.size()) {
        const size_t n = std::min<size_t>(x.size() - pos, 1 + rng() % (rng() % 2 ? 3000 : 90)), cap = 1 + rng() % 4000;
        const int flags = (rng() % 3 == 0) ? static_cast<int>(1 + rng() % 3) : 0;
        if (rng() % 4 == 0) hdrs.push_back({ rng() % 5 == 0 ? 0u : 1 + rng() % 300, rng() % 4 == 0 ? 1 : 0 });
        if (rng() % 9 == 0) ++n_ignored;
        if (!hdrs.empty() && hdrs[0].packet_length == 0 && !hdrs[0].invalid_header) hdrs[0].packet_length = 7; // (the error path is tested below)
        std::vector<c64> got(cap, c64{ -1, -1 }), want(cap, c64{ -1, -1 });
        CopySpan runs[2];
        int n_runs = 0, of = 0, rof = 0;
        size_t c = 0, hc = 0, ic = 0, rc = 0, rhc = 0, ric = 0;
        const gr4pm_status st = sdf_process_plan(h, n, cap, flags, hdrs.data(), hdrs.size(), n_ignored, &c, &hc, &ic, &of, runs, &n_runs);
        for (int r = 0; r < n_runs; ++r) std::memcpy(got.data() + runs[r].dst, x.data() + pos + runs[r].src, runs[r].len * sizeof(c64));
        std::vector<uint64_t> pl;
        std::vector<uint8_t> inv;
        for (auto& m : hdrs) pl.push_back(m.packet_length), inv.push_back(static_cast<uint8_t>(m.invalid_header));
        const int rst = orc_sdf_process(ref, x.data() + pos, n, want.data(), cap, flags, hdrs.size(), pl.data(), inv.data(), n_ignored,
                                        &rc, &rhc, &ric, &rof);
        CHECK(st == rst && c == rc && hc == rhc && ic == ric && of == rof, "sdf call at %zu: status %d/%d consumed %zu/%zu", pos, st, rst, c, rc);
        CHECK(same_bits(got.data(), want.data(), cap), "sdf call at %zu: items differ", pos);
        hdrs.erase(hdrs.begin(), hdrs.begin() + static_cast<long>(hc));
        n_ignored -= ic;
        pos += c;
        if (c == 0 && hdrs.empty()) hdrs.push_back({ 1 + rng() % 100, 0 }); // blocked on a header: supply one
    }
    // packet_length = 0 is the reference's exception (:143-145)
    SdfState e;
    gr4pm_header_msg zero{ 0, 0 };
    CopySpan runs[2];
    int n_runs, of;
    size_t c, hc, ic;
    sdf_process_plan(e, 10, 10, GR4PM_TAG_SYNCWORD, nullptr, 0, 0, &c, &hc, &ic, &of, runs, &n_runs);
    CHECK(sdf_process_plan(e, 10, 10, 0, &zero, 1, 0, &c, &hc, &ic, &of, runs, &n_runs) == GR4PM_ERR_INVALID, "packet_length 0 accepted");
    orc_sdf_destroy(ref);
}

// ---------------------------------------------------------------- the gate: same decisions as the block run item by item
static void sdf_gate_vs_block(std::mt19937_64& rng)
{
    SdfState g;
    orc_sdf* ref = orc_sdf_create(g.sps, g.syncword_size, g.header_size);
    const size_t n = 200000;
    const uint64_t packet_length = 1 + rng() % 400;
    std::vector<uint64_t> tag_at;
    for (uint64_t p = rng() % 2000; p < n; p += 1 + rng() % 9000) tag_at.push_back(p);
    // reference: the stream chunk by chunk (chunks cut at the tags), every accepted syncword answered at once
    std::vector<uint8_t> want(tag_at.size(), 0);
    const auto x = noise(rng, n);
    std::vector<c64> out(n);
    size_t pos = 0, ti = 0, pending = 0;
    while (pos < n) {
        while (ti < tag_at.size() && tag_at[ti] < pos) ++ti;
        const bool head = ti < tag_at.size() && tag_at[ti] == pos;
        const size_t nxt = head ? ti + 1 : ti;
        const size_t end = nxt < tag_at.size() ? static_cast<size_t>(tag_at[nxt]) : n;
        size_t c = 0, hc = 0, ic = 0;
        int of = 0;
        const uint64_t pl[1] = { packet_length };
        const uint8_t inv[1] = { 0 };
        orc_sdf_process(ref, x.data() + pos, end - pos, out.data(), n, head ? GR4PM_TAG_SYNCWORD : 0, pending, pl, inv, 0, &c, &hc, &ic, &of);
        if (of & GR4PM_TAG_SYNCWORD) {
            want[ti] = 1;
            ++pending;
        }
        pending -= hc;
        if (c == 0 && !of) break; // (cannot happen: every packet has its header)
        pos += c;
        if (head && c > 0) ++ti;
    }
    // gate: the same tags in a few calls
    std::vector<uint8_t> got(tag_at.size(), 7);
    size_t done = 0;
    while (done < tag_at.size()) {
        const size_t k = std::min<size_t>(tag_at.size() - done, 1 + rng() % 12);
        std::vector<gr4pm_header_msg> msgs(k, gr4pm_header_msg{ packet_length, 0 });
        size_t used = 0;
        CHECK(sdf_gate(g, tag_at.data() + done, k, msgs.data(), k, 1, got.data() + done, &used) == GR4PM_OK, "gate failed: %s", gr4pm::g_error);
        done += k;
    }
    CHECK(got == want, "gate decisions differ from the block's (%zu tags)", tag_at.size());
    orc_sdf_destroy(ref);
}

// ---------------------------------------------------------------- SymbolFilter: replay -> run table -> symbols
static void symbol_filter(std::mt19937_64& rng)
{
    const size_t sps = 1 + rng() % 5, arms = 1 + rng() % 33, n_taps = arms * (1 + rng() % 12) - (rng() % arms), delay = rng() % 60;
    std::vector<float> taps(std::max<size_t>(n_taps, 1));
    std::normal_distribution<float> g(0.f, 1.f);
    for (auto& t : taps) t = g(rng);
    SymfHostState h;
    h.sps = sps, h.num_arms = arms, h.delay = delay;
    h.reset_clock_phase = (sps - (delay % sps)) % sps; // symbol_filter.hpp:106-107
    orc_symf* ref = orc_symf_create(sps, taps.data(), taps.size(), arms, delay);
    const size_t arm_size = (taps.size() + arms - 1) / arms;
    const size_t n = 30000;
    const auto x = noise(rng, n);
    std::vector<gr4pm_tag> tags;
    for (uint64_t p = rng() % 500; p < n; p += 1 + rng() % 1500) {
        gr4pm_tag t{};
        t.index = p;
        t.amplitude = 0.5f + static_cast<float>(rng() % 1000) / 500.f;
        t.phase = g(rng);
        t.freq = 0.01 * g(rng);
        t.time_est = static_cast<float>(static_cast<int>(rng() % 1001) - 500) / 1000.f;
        t.flags = (rng() % 7 == 0) ? GR4PM_TAG_OTHER : GR4PM_TAG_SYNCWORD;
        tags.push_back(t);
    }
    size_t pos = 0;
    while (pos < n) {
        const size_t m = std::min<size_t>(n - pos, 1 + rng() % 6000), cap = 1 + rng() % (m / sps + 8);
        std::vector<gr4pm_tag> tin, tout(64), rtout(64);
        for (auto t : tags)
            if (t.index >= pos && t.index < pos + m) {
                t.index -= pos;
                tin.push_back(t);
            }
        SymReplay rp;
        symf_replay(h, m, cap, tin.data(), tin.size(), tout.data(), tout.size(), rp);
        // what the filter kernels do with the run table (symbol_filter.hpp:208-214: inner_product over the arm, m
        // ascending, hist[0] newest, then the scale)
        std::vector<c64> got(cap, c64{ -1, -1 }), want(cap, c64{ -1, -1 });
        for (const auto& r : rp.runs)
            for (unsigned k = 0; k < r.count; ++k) {
                std::complex<float> acc(0.f, 0.f);
                const long long newest = static_cast<long long>(pos) + r.in0 + static_cast<long long>(k) * static_cast<long long>(sps);
                for (size_t j = 0; r.arm + arms * j < taps.size() && j < arm_size; ++j) {
                    const long long i = newest - static_cast<long long>(j);
                    const std::complex<float> v = i >= 0 ? std::complex<float>(x[static_cast<size_t>(i)].re, x[static_cast<size_t>(i)].im) : std::complex<float>(0.f, 0.f);
                    acc += taps[r.arm + arms * j] * v;
                }
                acc = r.scale * acc;
                got[r.out0 + k] = { acc.real(), acc.imag() };
            }
        size_t rn_tags = 0, rcons = 0;
        const size_t rprod = orc_symf_process_c64(ref, x.data() + pos, m, want.data(), cap, reinterpret_cast<const orc_tag*>(tin.data()), tin.size(),
                                                  reinterpret_cast<orc_tag*>(rtout.data()), rtout.size(), &rn_tags, &rcons);
        CHECK(rp.pos == rcons && rp.produced == rprod && rp.n_pub == rn_tags, "symf call at %zu: consumed %zu/%zu produced %zu/%zu tags %zu/%zu (sps %zu arms %zu taps %zu delay %zu)",
              pos, rp.pos, rcons, rp.produced, rprod, rp.n_pub, rn_tags, sps, arms, taps.size(), delay);
        if (getenv("HOSTLOGIC_DEBUG") && !(rp.pos == rcons && rp.produced == rprod)) {
            fprintf(stderr, "  call: m %zu cap %zu, tags:", m, cap);
            for (auto& t : tin) fprintf(stderr, " [%llu te %.3f fl %d]", (unsigned long long)t.index, t.time_est, t.flags);
            fprintf(stderr, "\n");
        }
        CHECK(same_bits(got.data(), want.data(), std::min(rp.produced, rprod)), "symf call at %zu: symbols differ (sps %zu arms %zu taps %zu)", pos, sps, arms, taps.size());
        CHECK(rp.n_pub > tout.size() || same_tags(tout.data(), rtout.data(), std::min(rp.n_pub, rn_tags)), "symf call at %zu: tags differ", pos);
        if (rcons == 0) break;
        pos += rcons;
    }
    orc_symf_destroy(ref);
}

// ---------------------------------------------------------------- PayloadMetadataInsert -> SyncwordRemove -> HeaderPayloadSplit
static void control_blocks(std::mt19937_64& rng)
{
    const size_t n = 60000;
    const auto x = noise(rng, n);
    std::vector<gr4pm_tag> tags;
    std::vector<gr4pm_header_msg> msgs;
    for (uint64_t p = rng() % 300; p < n; p += 150 + rng() % 2500) {
        gr4pm_tag t{};
        t.index = p;
        t.amplitude = 1.f, t.phase = 0.25f, t.flags = GR4PM_TAG_SYNCWORD;
        tags.push_back(t);
        msgs.push_back({ 1 + rng() % 200, rng() % 5 == 0 ? 1 : 0 });
    }
    PmiState h;
    h.syncword_bw = 0.02, h.header_bw = 0.01, h.payload_bw = 0.005;
    orc_pmi* ref = orc_pmi_create(h.syncword_size, h.header_size, h.syncword_bw, h.header_bw, h.payload_bw);
    std::vector<c64> pm(n, c64{ -1, -1 }), rpm(n, c64{ -1, -1 });
    std::vector<gr4pm_packet_tag> pt(3 * tags.size() + 8), rpt(3 * tags.size() + 8);
    PmiReplay rp;
    const gr4pm_status st = pmi_replay(h, n, n, tags.data(), tags.size(), msgs.data(), msgs.size(), 0, pt.data(), pt.size(), rp);
    apply(rp.spans, x.data(), pm.data());
    std::vector<uint64_t> pl;
    std::vector<uint8_t> inv;
    for (auto& m : msgs) pl.push_back(m.packet_length), inv.push_back(static_cast<uint8_t>(m.invalid_header));
    size_t rnt = 0, rc = 0, rprod = 0, rused = 0, rign = 0;
    const int rst = orc_pmi_process(ref, x.data(), n, rpm.data(), n, reinterpret_cast<const orc_tag*>(tags.data()), tags.size(), pl.data(), inv.data(), msgs.size(),
                                    reinterpret_cast<orc_ptag*>(rpt.data()), rpt.size(), &rnt, &rc, &rprod, &rused, &rign);
    CHECK(st == rst && rp.consumed == rc && rp.produced == rprod && rp.headers_used == rused && rp.ignored == rign && rp.n_pub == rnt,
          "pmi: status %d/%d consumed %zu/%zu produced %zu/%zu headers %zu/%zu ignored %zu/%zu tags %zu/%zu", st, rst, rp.consumed, rc, rp.produced, rprod,
          rp.headers_used, rused, rp.ignored, rign, rp.n_pub, rnt);
    CHECK(same_bits(pm.data(), rpm.data(), n), "pmi: items differ");
    CHECK(same_ptags(pt.data(), reinterpret_cast<const gr4pm_packet_tag*>(rpt.data()), std::min(rp.n_pub, rnt)), "pmi: tags differ");
    orc_pmi_destroy(ref);
    {
        // the receivers' way of calling it: ONE MESSAGE PER TAG (headers_per_tag), in several calls with random cuts, a
        // message now and then still pending (invalid_header == 2) when its tag arrives and resolved before the next
        // call -- against the oracle fed, in one call, with the messages of the packets that were actually opened
        PmiState hp;
        hp.syncword_bw = 0.02, hp.header_bw = 0.01, hp.payload_bw = 0.005;
        std::vector<c64> got(n, c64{ -1, -1 });
        std::vector<gr4pm_packet_tag> gt;
        std::vector<gr4pm_header_msg> opened;
        size_t pos = 0, opos = 0;
        bool owe_resolve = false;
        gr4pm_header_msg owed{};
        std::vector<c64> carry; // symbols a call could not take (it waits for a pending message), with their tags
        std::vector<gr4pm_tag> carry_tags;
        std::vector<gr4pm_header_msg> carry_msgs;
        while (pos < n || !carry.empty()) {
            if (owe_resolve) {
                if (hp.in_packet && !hp.has_held) {
                    hp.held = owed;
                    hp.has_held = true;
                }
                owe_resolve = false;
            }
            const size_t m = std::min<size_t>(n - pos, 1 + rng() % 7000);
            std::vector<c64> in(carry);
            in.insert(in.end(), x.begin() + static_cast<long>(pos), x.begin() + static_cast<long>(pos + m));
            std::vector<gr4pm_tag> tin(carry_tags);
            std::vector<gr4pm_header_msg> min_(carry_msgs);
            for (size_t i = 0; i < tags.size(); ++i)
                if (tags[i].index >= pos && tags[i].index < pos + m) {
                    tin.push_back(tags[i]);
                    tin.back().index = tags[i].index - pos + carry.size();
                    gr4pm_header_msg mm = msgs[i];
                    if (rng() % 6 == 0 && !owe_resolve && !(hp.in_packet && !hp.has_held)) { // pending until the next call
                        owed = mm;
                        mm.invalid_header = 2;
                        owe_resolve = true;
                    }
                    min_.push_back(mm);
                }
            if (in.empty()) break;
            std::vector<c64> out(in.size());
            std::vector<gr4pm_packet_tag> tout(3 * tin.size() + 8);
            PmiReplay r2;
            std::vector<gr4pm_header_msg> pad(std::max<size_t>(min_.size(), 1));
            std::copy(min_.begin(), min_.end(), pad.begin());
            const gr4pm_status s2 = pmi_replay(hp, in.size(), in.size(), tin.data(), tin.size(), pad.data(), tin.size(), 1, tout.data(), tout.size(), r2);
            CHECK(s2 == GR4PM_OK, "pmi per tag: status %d (%s)", s2, gr4pm::g_error);
            apply(r2.spans, in.data(), out.data());
            std::memcpy(got.data() + opos, out.data(), r2.produced * sizeof(c64));
            for (size_t i = 0; i < r2.n_pub; ++i) {
                gt.push_back(tout[i]);
                gt.back().index += opos;
                if (tout[i].kind == GR4PM_PKT_SYNCWORD) gt.back().syncword.index += pos - carry.size(); // call -> stream index
                if (tout[i].kind == GR4PM_PKT_SYNCWORD)
                    for (size_t j = 0; j < tin.size(); ++j)
                        if (tin[j].index == tout[i].syncword.index) opened.push_back(min_[j].invalid_header == 2 ? owed : min_[j]);
            }
            opos += r2.produced;
            // what the call did not take waits, with its tags and messages, in front of the next call (the receivers' stage 2)
            carry.assign(in.begin() + static_cast<long>(r2.consumed), in.end());
            carry_tags.clear();
            carry_msgs.clear();
            for (size_t j = 0; j < tin.size(); ++j)
                if (tin[j].index >= r2.consumed) {
                    carry_tags.push_back(tin[j]);
                    carry_tags.back().index -= r2.consumed;
                    carry_msgs.push_back(min_[j]);
                }
            CHECK(r2.consumed == in.size() || (hp.in_packet && !hp.has_held), "pmi per tag: stopped at %zu of %zu without waiting for a message", r2.consumed, in.size());
            pos += m;
            if (pos >= n && !carry.empty() && !owe_resolve) break; // (the stream ends inside a wait)
        }
        // the oracle with the list of the opened packets' messages
        orc_pmi* ref2 = orc_pmi_create(hp.syncword_size, hp.header_size, hp.syncword_bw, hp.header_bw, hp.payload_bw);
        std::vector<uint64_t> pl2;
        std::vector<uint8_t> inv2;
        for (auto& mm : opened) pl2.push_back(mm.packet_length), inv2.push_back(static_cast<uint8_t>(mm.invalid_header));
        std::vector<c64> want(n, c64{ -1, -1 });
        std::vector<gr4pm_packet_tag> wt(3 * tags.size() + 8);
        size_t wnt = 0, wc = 0, wp = 0, wu = 0, wi = 0;
        orc_pmi_process(ref2, x.data(), n, want.data(), n, reinterpret_cast<const orc_tag*>(tags.data()), tags.size(), pl2.data(), inv2.data(), opened.size(),
                        reinterpret_cast<orc_ptag*>(wt.data()), wt.size(), &wnt, &wc, &wp, &wu, &wi);
        // A call that ends exactly behind a packet's syncword leaves _position == syncword_size: the reference publishes the
        // header-start tag at the end of that call AND again at the start of the next one (payload_metadata_insert.hpp:
        // 185-194 has no "already published" state), and so does the replay, call for call.  The one-call oracle run has no
        // such cut: the second copy (same kind, same output index) is dropped before the comparison.
        for (size_t i = 1; i < gt.size();)
            if (gt[i].kind == GR4PM_PKT_HEADER_START && gt[i - 1].kind == GR4PM_PKT_HEADER_START && gt[i].index == gt[i - 1].index) gt.erase(gt.begin() + static_cast<long>(i));
            else ++i;
        CHECK(opos == wp && gt.size() == wnt, "pmi per tag: produced %zu/%zu tags %zu/%zu", opos, wp, gt.size(), wnt);
        CHECK(same_bits(got.data(), want.data(), std::min(opos, wp)), "pmi per tag: items differ");
        CHECK(same_ptags(gt.data(), wt.data(), std::min(gt.size(), wnt)), "pmi per tag: tags differ");
        orc_pmi_destroy(ref2);
    }
    // SyncwordRemove over PayloadMetadataInsert's output, in several calls
    SrState sh;
    orc_sr* sref = orc_sr_create(sh.syncword_size);
    const size_t n2 = rp.produced;
    std::vector<c64> data(n2 + 1, c64{ -1, -1 }), rdata(n2 + 1, c64{ -1, -1 });
    std::vector<gr4pm_packet_tag> dt(pt.size()), rdt(pt.size());
    size_t rndt = 0;
    const size_t rn_data = orc_sr_process(sref, rpm.data(), n2, rdata.data(), reinterpret_cast<const orc_ptag*>(rpt.data()), rnt,
                                          reinterpret_cast<orc_ptag*>(rdt.data()), rdt.size(), &rndt);
    size_t pos = 0, opos = 0, ntags = 0;
    while (pos < n2) {
        const size_t m = std::min<size_t>(n2 - pos, 1 + rng() % 9000);
        std::vector<gr4pm_packet_tag> tin, tout(pt.size());
        for (size_t i = 0; i < rp.n_pub; ++i)
            if (pt[i].index >= pos && pt[i].index < pos + m) {
                tin.push_back(pt[i]);
                tin.back().index -= pos;
            }
        SrReplay sr;
        sr_replay(sh, m, tin.data(), tin.size(), tout.data(), tout.size(), sr);
        apply(sr.spans, pm.data() + pos, data.data() + opos);
        for (size_t i = 0; i < sr.n_pub; ++i) {
            dt[ntags] = tout[i];
            dt[ntags++].index += opos;
        }
        pos += m;
        opos += sr.produced;
    }
    CHECK(opos == rn_data && ntags == rndt, "syncword remove: %zu/%zu items %zu/%zu tags", opos, rn_data, ntags, rndt);
    CHECK(same_bits(data.data(), rdata.data(), std::min(opos, rn_data)), "syncword remove: items differ");
    CHECK(same_ptags(dt.data(), rdt.data(), std::min(ntags, rndt)), "syncword remove: tags differ");
    orc_sr_destroy(sref);
    // HeaderPayloadSplit<float> over the soft symbols (here: the real parts twice, as the QPSK LLR decoder lays them out:
    // two floats per symbol, tags at twice the index -- only positions matter to the state machine)
    std::vector<float> llr(2 * opos);
    for (size_t i = 0; i < opos; ++i) llr[2 * i] = data[i].re, llr[2 * i + 1] = data[i].im;
    std::vector<gr4pm_packet_tag> lt(ntags);
    for (size_t i = 0; i < ntags; ++i) {
        lt[i] = dt[i];
        lt[i].index *= 2;
    }
    HpsState hh;
    orc_hps* href = orc_hps_create(hh.header_size);
    std::vector<float> hd(llr.size() + 1), pd(llr.size() + 1), rhd(llr.size() + 1), rpd(llr.size() + 1);
    std::vector<gr4pm_packet_tag> ht(ntags + 1), ptg(ntags + 1), rht(ntags + 1), rptg(ntags + 1);
    size_t rnh = 0, rnp = 0, rnht = 0, rnpt = 0;
    const int hrst = orc_hps_process(href, llr.data(), llr.size(), rhd.data(), &rnh, rpd.data(), &rnp, reinterpret_cast<const orc_ptag*>(lt.data()), ntags,
                                     reinterpret_cast<orc_ptag*>(rht.data()), &rnht, reinterpret_cast<orc_ptag*>(rptg.data()), &rnpt, ntags + 1);
    HpsReplay hr;
    const gr4pm_status hst = hps_replay(hh, llr.size(), lt.data(), ntags, ht.data(), ptg.data(), ntags + 1, hr);
    CHECK((hst == GR4PM_OK) == (hrst == 0), "header/payload split: status %d / %d", hst, hrst);
    if (hst == GR4PM_OK && hrst == 0) {
        for (const auto& s : hr.header_spans) std::memcpy(hd.data() + s.dst, llr.data() + s.src, s.len * sizeof(float));
        for (const auto& s : hr.payload_spans) std::memcpy(pd.data() + s.dst, llr.data() + s.src, s.len * sizeof(float));
        CHECK(hr.n_header == rnh && hr.n_payload == rnp && hr.n_header_tags == rnht && hr.n_payload_tags == rnpt, "header/payload split: counts differ");
        CHECK(std::memcmp(hd.data(), rhd.data(), std::min(hr.n_header, rnh) * sizeof(float)) == 0 &&
                  std::memcmp(pd.data(), rpd.data(), std::min(hr.n_payload, rnp) * sizeof(float)) == 0, "header/payload split: items differ");
        CHECK(same_ptags(ht.data(), rht.data(), std::min(hr.n_header_tags, rnht)) &&
                  same_ptags(ptg.data(), rptg.data(), std::min(hr.n_payload_tags, rnpt)), "header/payload split: tags differ");
    }
    orc_hps_destroy(href);
}

// ---------------------------------------------------------------- the receivers' slot rings and stage loop, mock stage bodies
static void slot_pipeline(std::mt19937_64& rng)
{
    constexpr int kSlots = 7, kStages = 5, kBatches = 400;
    using Q = SlotQueue<16>;
    Q free_slots, q[kStages], done;
    struct Slot {
        long value = 0;
        int trace = 0, failed_at = -1;
        unsigned char scratch[256];
    } slots[kSlots];
    for (int i = 0; i < kSlots; ++i) CHECK(free_slots.push(i), "push");
    const int throw_stage = static_cast<int>(rng() % kStages), throw_batch = static_cast<int>(rng() % kBatches);
    std::atomic<int> in_stage[kStages];
    for (auto& a : in_stage) a = 0;
    std::vector<std::thread> threads;
    for (int s = 0; s < kStages; ++s)
        threads.emplace_back([&, s] {
            run_stage(
                q[s], s + 1 < kStages ? &q[s + 1] : &done, /*forward_quit=*/s + 1 < kStages,
                [&](int i) {
                    CHECK(in_stage[s].fetch_add(1) == 0, "two batches inside stage %d", s); // one at a time, in order
                    Slot& sl = slots[i];
                    if (sl.failed_at < 0) {
                        if (s == throw_stage && sl.value / 1000 == throw_batch) {
                            in_stage[s].fetch_sub(1);
                            throw std::runtime_error("mock stage failure");
                        }
                        CHECK(sl.trace == s, "slot %d reached stage %d after %d stages", i, s, sl.trace);
                        sl.trace = s + 1;
                        sl.value += s + 1;
                        std::memset(sl.scratch, s, sizeof(sl.scratch)); // (a second owner of the slot would race here)
                    }
                    if ((sl.value & 15) == 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
                    in_stage[s].fetch_sub(1);
                },
                [&](int i) { slots[i].failed_at = s; });
        });
    int submitted = 0, collected = 0, failures = 0;
    long expect_next = 0;
    while (collected < kBatches) {
        while (submitted < kBatches && submitted - collected < kSlots - 1) {
            const int i = free_slots.pop();
            slots[i] = Slot{};
            slots[i].value = 1000L * submitted;
            ++submitted;
            CHECK(q[0].push(i), "ring full with %d in flight", submitted - collected);
        }
        const int i = done.pop();
        const Slot& sl = slots[i];
        CHECK(sl.value / 1000 == expect_next, "batches out of order: %ld, expected %ld", sl.value / 1000, expect_next);
        if (sl.failed_at >= 0) {
            ++failures;
            CHECK(sl.failed_at == throw_stage && sl.value / 1000 == throw_batch, "wrong batch failed");
        } else {
            CHECK(sl.trace == kStages && sl.value % 1000 == 15, "batch %ld incomplete: %d stages", sl.value / 1000, sl.trace);
        }
        ++expect_next;
        ++collected;
        CHECK(free_slots.push(i), "push");
    }
    CHECK(failures == 1, "%d failed batches (one injected)", failures);
    q[0].stop();
    for (auto& t : threads) t.join();
    CHECK(done.size() == 0 && free_slots.size() == kSlots, "slots lost");
    // a full ring refuses, it does not overwrite
    SlotQueue<4> small;
    for (int i = 0; i < 4; ++i) CHECK(small.push(i), "push");
    CHECK(!small.push(99) && small.pop() == 0, "a full ring must refuse");
}

// ---------------------------------------------------------------- the ZeroMQ PUB endpoint (hostlogic/zmtp_pub.hpp)
// A hand-written SUB peer over a blocking socket: greeting, READY, one subscription, then frames
struct RawSub {
    int fd = -1;
    bool open(int port)
    {
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        sockaddr_in a{};
        a.sin_family = AF_INET;
        a.sin_port = htons(static_cast<uint16_t>(port));
        a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
        timeval tv{ 5, 0 };
        (void)::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        return ::connect(fd, reinterpret_cast<sockaddr*>(&a), sizeof a) == 0;
    }
    bool rd(void* p, size_t n)
    {
        uint8_t* b = static_cast<uint8_t*>(p);
        while (n) {
            const ssize_t k = ::recv(fd, b, n, 0);
            if (k <= 0) return false;
            b += k, n -= static_cast<size_t>(k);
        }
        return true;
    }
    bool wr(const void* p, size_t n) { return ::send(fd, p, n, MSG_NOSIGNAL) == static_cast<ssize_t>(n); }
    bool handshake(const std::vector<uint8_t>& topic)
    {
        uint8_t g[64] = { 0xFF, 0, 0, 0, 0, 0, 0, 0, 1, 0x7F, 3, 0, 'N', 'U', 'L', 'L' };
        uint8_t peer[64], hdr[2], body[256];
        static const char ready[] = "\x04\x19\x05READY\x0BSocket-Type\x00\x00\x00\x03SUB";
        if (!wr(g, 64) || !rd(peer, 64) || !wr(ready, sizeof ready - 1) || !rd(hdr, 2) || !rd(body, hdr[1])) return false;
        const auto want_g = ZmtpPub::greeting();
        const auto want_r = ZmtpPub::ready();
        CHECK(std::memcmp(peer, want_g.data(), 64) == 0, "greeting");
        CHECK(hdr[0] == 4 && hdr[1] == want_r.size() - 2 && std::memcmp(body, want_r.data() + 2, hdr[1]) == 0, "READY");
        std::vector<uint8_t> sub{ 0, static_cast<uint8_t>(1 + topic.size()), 1 };
        sub.insert(sub.end(), topic.begin(), topic.end());
        return wr(sub.data(), sub.size());
    }
    bool message(std::vector<uint8_t>& out)
    {
        uint8_t flags, l1;
        if (!rd(&flags, 1)) return false;
        uint64_t len = 0;
        if (flags & 2) {
            uint8_t l8[8];
            if (!rd(l8, 8)) return false;
            for (uint8_t b : l8) len = (len << 8) | b;
        } else {
            if (!rd(&l1, 1)) return false;
            len = l1;
        }
        out.resize(len);
        return len == 0 || rd(out.data(), len);
    }
    ~RawSub()
    {
        if (fd >= 0) ::close(fd);
    }
};
static void zmtp_pub(std::mt19937_64& rng)
{
    ZmtpPub pub;
    CHECK(pub.bind("udp://127.0.0.1:1") != GR4PM_OK && pub.bind("tcp://nowhere:1") != GR4PM_OK, "bad endpoints must be refused");
    CHECK(pub.send("x", 1) != GR4PM_OK, "send before bind");
    CHECK(pub.bind("tcp://127.0.0.1:*") == GR4PM_OK && pub.port() > 0, "bind");
    CHECK(pub.send("nobody", 6) == GR4PM_OK && pub.dropped() == 0, "no subscriber: dropped silently, as a PUB socket does");
    const int n_msgs = 300;
    std::vector<std::vector<uint8_t>> msgs(n_msgs);
    for (auto& m : msgs) {
        m.resize(rng() % 3 ? rng() % 200 : 256 + rng() % 70000); // short and long frames
        for (auto& b : m) b = static_cast<uint8_t>(rng());
        if (!m.empty()) m[0] = static_cast<uint8_t>(rng() % 2 ? 'A' : 'B');
    }
    // three subscribers in threads of their own: everything / topic "A" / one that connects, subscribes and never reads
    std::atomic<int> ready{ 0 };
    auto reader = [&](std::vector<uint8_t> topic, std::vector<std::vector<uint8_t>>* got) {
        RawSub s;
        CHECK(s.open(pub.port()) && s.handshake(topic), "subscriber handshake");
        ready.fetch_add(1);
        if (!got) {
            std::this_thread::sleep_for(std::chrono::milliseconds(300));
            return;
        }
        std::vector<uint8_t> m;
        while (s.message(m)) {
            if (m.size() == 3 && std::memcmp(m.data(), "END", 3) == 0) break;
            got->push_back(m);
        }
    };
    std::vector<std::vector<uint8_t>> got_all, got_a;
    std::thread t_all(reader, std::vector<uint8_t>{}, &got_all), t_a(reader, std::vector<uint8_t>{ 'A' }, &got_a),
        t_mute(reader, std::vector<uint8_t>{}, nullptr);
    { // a peer that is no ZMTP peer at all
        RawSub junk;
        CHECK(junk.open(pub.port()) && junk.wr("GET / HTTP/1.0\r\n\r\n", 18), "junk peer");
    }
    for (int i = 0; i < 500 && (ready.load() < 3 || pub.subscribers() < 3); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    CHECK(pub.subscribers() == 3, "%zu subscribers", pub.subscribers());
    for (const auto& m : msgs) CHECK(pub.send(m.data(), m.size()) == GR4PM_OK, "send");
    CHECK(pub.send("END", 3) == GR4PM_OK && pub.send("AEND", 4) == GR4PM_OK, "send");
    t_all.join();
    std::vector<std::vector<uint8_t>> want_a;
    for (const auto& m : msgs)
        if (!m.empty() && m[0] == 'A') want_a.push_back(m);
    CHECK(got_all == msgs, "the subscriber of everything got %zu of %d messages", got_all.size(), n_msgs);
    // (the topic-A subscriber never sees "END": its stream ends with "AEND", or with the endpoint closing)
    t_mute.join();
    pub.close(500);
    t_a.join();
    if (!got_a.empty() && got_a.back() == std::vector<uint8_t>{ 'A', 'E', 'N', 'D' }) got_a.pop_back();
    CHECK(got_a == want_a, "the subscriber of topic A got %zu messages, expected %zu", got_a.size(), want_a.size());
}

// ---------------------------------------------------------------- the composed tail table (hostlogic/tail_plan.hpp)
// A PayloadMetadataInsert-shaped tag stream (syncword | header | payload of (len + 4) * 4 symbols, headers that did not
// decode among them) cut into calls at random symbols.  Reference: the blocks one by one with plain loops -- SyncwordRemove's
// gather, the QPSK LLRs, the descrambler over its runs, HeaderPayloadSplit's two gathers, then slicer + packer over the
// whole payload stream.  Under test: the same host state machines + compose_tail + the kernel's own per-symbol / per-byte
// functions (tail_llr_pair, tail_payload_byte), writing header LLRs and payload BITS in place, partial bytes at every cut.
static void tail_compose(std::mt19937_64& rng)
{
    // the stream and its tags (absolute symbol positions)
    std::vector<gr4pm_packet_tag> all;
    uint64_t pos = 0;
    const int n_packets = 40 + static_cast<int>(rng() % 40);
    for (int k = 0; k < n_packets; ++k) {
        gr4pm_packet_tag t{};
        t.index = pos;
        t.kind = GR4PM_PKT_SYNCWORD;
        t.constellation = 0;
        all.push_back(t);
        t = gr4pm_packet_tag{};
        t.index = pos + 64;
        t.kind = GR4PM_PKT_HEADER_START;
        t.constellation = 2;
        all.push_back(t);
        pos += 64 + 128;
        if (rng() % 6) { // the header decoded: a payload follows
            const uint64_t len = 1 + rng() % (rng() % 4 ? 40 : 700);
            t = gr4pm_packet_tag{};
            t.index = pos;
            t.kind = GR4PM_PKT_PAYLOAD;
            t.constellation = -1;
            t.packet_length = len;
            t.payload_symbols = (len + 4) * 4;
            t.payload_bits = 2 * t.payload_symbols;
            all.push_back(t);
            pos += t.payload_symbols;
        }
    }
    const uint64_t n_total = pos;
    std::vector<float> sym(2 * n_total);
    for (auto& v : sym) v = static_cast<float>(static_cast<double>(static_cast<int64_t>(rng() % 2001) - 1000) / 640.0); // zeros among them
    const float scale = 2.0f / (0.7f * 0.7f);
    ScrState scr_ref, scr_new;
    scr_ref.prefix = scr_new.prefix = 5;
    scr_ref.period = scr_new.period = 997;
    scr_ref.table_len = scr_new.table_len = 1002;
    std::vector<uint8_t> seq(1002);
    for (auto& b : seq) b = static_cast<uint8_t>(rng() & 1);
    SrState sr_ref, sr_new;
    HpsState hps_ref, hps_new;
    hps_ref.header_size = hps_new.header_size = 256;
    std::vector<float> hdr_ref, pay_ref, hdr_new;
    std::vector<uint8_t> packed_new;
    uint64_t pay_bits_new = 0;
    for (uint64_t start = 0; start < n_total;) {
        const uint64_t n = std::min<uint64_t>(n_total - start, 1 + rng() % (rng() % 3 ? 3000 : 90));
        std::vector<gr4pm_packet_tag> tags;
        for (auto t : all)
            if (t.index >= start && t.index < start + n) {
                t.index -= start;
                tags.push_back(t);
            }
        const float* x = sym.data() + 2 * start;
        std::vector<gr4pm_packet_tag> dt(tags.size() + 4), lt, ht(tags.size() + 4), pt(tags.size() + 4);
        // ---- the blocks one by one
        {
            SrReplay sr;
            sr_replay(sr_ref, n, tags.data(), tags.size(), dt.data(), dt.size(), sr);
            std::vector<float> data(2 * sr.produced), llr, desc;
            for (const auto& sp : sr.spans) std::memcpy(data.data() + 2 * sp.dst, x + 2 * sp.src, 8 * sp.len);
            llr.resize(data.size());
            for (size_t i = 0; i < data.size(); ++i) llr[i] = scale * data[i];
            lt.assign(dt.begin(), dt.begin() + static_cast<ptrdiff_t>(sr.n_pub));
            std::vector<uint64_t> resets;
            for (auto& t : lt) {
                t.index *= 2;
                if (t.kind == GR4PM_PKT_HEADER_START) resets.push_back(t.index);
            }
            std::vector<ScrambleRun> runs;
            CHECK(scramble_runs(scr_ref, llr.size(), resets.data(), resets.size(), runs) == GR4PM_OK, "runs");
            desc.resize(llr.size());
            for (const auto& r : runs)
                for (uint64_t i = 0; i < r.len; ++i) {
                    const float a = llr[r.start + i];
                    desc[r.start + i] = seq[scr_index(r.phase + i, scr_ref.prefix, scr_ref.period)] ? -a : a;
                }
            HpsReplay hp;
            CHECK(hps_replay(hps_ref, desc.size(), lt.data(), lt.size(), ht.data(), pt.data(), ht.size(), hp) == GR4PM_OK, "hps");
            const size_t h0 = hdr_ref.size(), p0 = pay_ref.size();
            hdr_ref.resize(h0 + hp.n_header);
            pay_ref.resize(p0 + hp.n_payload);
            for (const auto& sp : hp.header_spans) std::memcpy(hdr_ref.data() + h0 + sp.dst, desc.data() + sp.src, 4 * sp.len);
            for (const auto& sp : hp.payload_spans) std::memcpy(pay_ref.data() + p0 + sp.dst, desc.data() + sp.src, 4 * sp.len);
        }
        // ---- the host halves + the composed table + the kernel's functions
        {
            SrReplay sr;
            sr_replay(sr_new, n, tags.data(), tags.size(), dt.data(), dt.size(), sr);
            std::vector<gr4pm_packet_tag> lt2(dt.begin(), dt.begin() + static_cast<ptrdiff_t>(sr.n_pub));
            std::vector<uint64_t> resets;
            for (auto& t : lt2) {
                t.index *= 2;
                if (t.kind == GR4PM_PKT_HEADER_START) resets.push_back(t.index);
            }
            std::vector<ScrambleRun> runs;
            CHECK(scramble_runs(scr_new, 2 * sr.produced, resets.data(), resets.size(), runs) == GR4PM_OK, "runs");
            HpsReplay hp;
            CHECK(hps_replay(hps_new, 2 * sr.produced, lt2.data(), lt2.size(), ht.data(), pt.data(), ht.size(), hp) == GR4PM_OK, "hps");
            std::vector<TailSpan> spans;
            const bool ok = compose_tail(sr.spans, runs, hp, hdr_new.size(), pay_bits_new, spans);
            CHECK(ok, "compose_tail refused a regular tag stream");
            hdr_new.resize(hdr_new.size() + hp.n_header);
            packed_new.resize((pay_bits_new + hp.n_payload + 7) / 8 + 1, 0xAA); // (stale contents behind the stream's end)
            for (const auto& sp : spans) {
                if (sp.kind == 0) {
                    for (unsigned i = 0; i < sp.n_sym; ++i)
                        tail_llr_pair(x[2 * (sp.src + i)], x[2 * (sp.src + i) + 1], scale, seq.data(), sp.phase + 2ull * i,
                                      scr_new.prefix, scr_new.period, hdr_new[sp.dst + 2ull * i], hdr_new[sp.dst + 2ull * i + 1]);
                } else {
                    for (uint64_t B = sp.dst >> 3; B < (sp.dst + 2ull * sp.n_sym + 7) >> 3; ++B) {
                        unsigned v, mask;
                        tail_payload_byte(sp, B, x, scale, seq.data(), scr_new.prefix, scr_new.period, v, mask);
                        packed_new[B] = mask == 0xFFu ? static_cast<uint8_t>(v) : static_cast<uint8_t>((packed_new[B] & ~mask) | v);
                    }
                }
            }
            pay_bits_new += hp.n_payload;
        }
        start += n;
    }
    CHECK(hdr_ref.size() == hdr_new.size() && (hdr_ref.empty() || std::memcmp(hdr_ref.data(), hdr_new.data(), 4 * hdr_ref.size()) == 0),
          "header LLRs differ (%zu / %zu)", hdr_ref.size(), hdr_new.size());
    CHECK(pay_bits_new == pay_ref.size() && pay_ref.size() % 8 == 0, "payload stream %llu bits, reference %zu",
          static_cast<unsigned long long>(pay_bits_new), pay_ref.size());
    size_t wrong = 0;
    for (size_t B = 0; B < pay_ref.size() / 8; ++B) {
        unsigned v = 0;
        for (int k = 0; k < 8; ++k) v = (v << 1) | (pay_ref[8 * B + static_cast<size_t>(k)] < 0.0f ? 1u : 0u); // k_slice_pack
        wrong += packed_new[B] != v;
    }
    CHECK(wrong == 0, "%zu of %zu packed bytes differ", wrong, pay_ref.size() / 8);
}

int main(int argc, char** argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 20;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 4;
    std::mt19937_64 rng(seed);
    for (int c = 0; c < cases; ++c) {
        sdf_calls(rng);
        sdf_gate_vs_block(rng);
        symbol_filter(rng);
        control_blocks(rng);
        tail_compose(rng);
    }
    for (int c = 0; c < std::max(2, cases / 5); ++c) slot_pipeline(rng);
    for (int c = 0; c < std::max(2, cases / 10); ++c) zmtp_pub(rng);
    printf("hostlogic_san: %d cases, seed %llu: %d failures\n", cases, seed, g_failures);
    return g_failures ? 1 : 0;
}
