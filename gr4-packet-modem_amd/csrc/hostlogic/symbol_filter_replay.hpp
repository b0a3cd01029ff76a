// hostlogic/symbol_filter_replay.hpp -- SymbolFilter's tag-driven state machine (symbol_filter.hpp:130-238) without HIP:
// which outputs a call produces, from which input item, with which polyphase arm and scale (the run table the filter
// kernels work from), and where the queued tags leave.
#pragma once
#include <algorithm>
#include <cmath>
#include <deque>
#include <vector>

#include "base.hpp"

namespace gr4pm {
namespace hostlogic {

// SymbolFilter (symbol_filter.hpp:208-214): y = scale * sum_m arm[m] * x[idx - m]
struct SymRun {
    long long in0;  // input index of the newest sample of the run's first output
    unsigned out0;  // first output index
    unsigned count; // outputs, spaced samples_per_symbol apart
    unsigned arm;
    float scale;
    unsigned wg0;   // first workgroup of the run (workgroups never straddle runs)
    unsigned chan;  // channel of the run (launches that span channels: SymChan table)
};
struct SymQueued {
    long value; // gr::Tag::index as counted down in symbol_filter.hpp:183-185,235-237
    gr4pm_tag tag;
};
// host replica of the tag-driven state (symbol_filter.hpp:44-50) and the settings it depends on
struct SymfHostState {
    size_t sps = 4, num_arms = 32, delay = 0;
    size_t clock_phase = 0, reset_clock_phase = 0, pfb_arm = 0;
    float scale = 1.0f;
    std::deque<SymQueued> queue;
};

// Host replay of the tag-driven state machine of symbol_filter.hpp:130-238 over one call: which outputs exist, from
// which input, with which arm and scale (runs), and where the tags leave.
struct SymReplay {
    std::vector<SymRun> runs;
    size_t pos = 0, produced = 0, n_pub = 0;
    bool tag_overflow = false;
};
inline void symf_replay(SymfHostState& h, size_t n_in, size_t out_cap, const gr4pm_tag* tags_in,
                        size_t n_tags_in, gr4pm_tag* tags_out, size_t tags_cap, SymReplay& rp)
{
    const size_t sps = h.sps;
    const long half = static_cast<long>(sps / 2);
    std::vector<SymRun>& runs = rp.runs;
    size_t& pos = rp.pos;
    size_t& produced = rp.produced;
    size_t& n_pub = rp.n_pub;
    bool& tag_overflow = rp.tag_overflow;
    auto publish = [&](const gr4pm_tag& t, size_t out_index) {
        if (tags_out && n_pub < tags_cap) {
            tags_out[n_pub] = t;
            tags_out[n_pub].index = out_index;
        } else {
            tag_overflow = true;
        }
        ++n_pub;
    };
    // main loop of :208-238 over k items without tag events; returns items actually consumed
    auto advance = [&](size_t k) -> size_t {
        size_t done = 0;
        while (done < k && produced < out_cap) {
            if (h.clock_phase >= sps) { // only reachable through the sps <= 2 corner of :182,:194
                ++h.clock_phase;
                if (h.clock_phase >= sps) h.clock_phase = 0;
                for (auto& q : h.queue) --q.value;
                ++done;
                ++pos;
                continue;
            }
            const size_t span = k - done;
            const size_t u0 = (sps - h.clock_phase) % sps; // offset of the first output
            size_t count = u0 < span ? (span - u0 + sps - 1) / sps : 0;
            size_t eff = span;
            if (produced + count >= out_cap) { // :208 stops once the output is full: right behind the item of the last
                count = out_cap - produced;    // output, also when the outputs of this span fit exactly (the items
                eff = u0 + (count - 1) * sps + 1; // behind it wait for the next call; found by tests/hostlogic)
            }
            if (count > 0) {
                SymRun r;
                r.in0 = static_cast<long long>(pos + u0);
                r.out0 = static_cast<unsigned>(produced);
                r.count = static_cast<unsigned>(count);
                r.arm = static_cast<unsigned>(h.pfb_arm);
                r.scale = h.scale;
                runs.push_back(r);
                // tags leave on the first output whose countdown is below sps/2 (:218-228)
                while (!h.queue.empty()) {
                    const long v = h.queue.front().value;
                    const long umin = std::max<long>(0, v - half + 1);
                    size_t tix = 0;
                    if (static_cast<size_t>(umin) > u0) tix = (static_cast<size_t>(umin) - u0 + sps - 1) / sps;
                    if (tix >= count) break;
                    publish(h.queue.front().tag, produced + tix);
                    h.queue.pop_front();
                }
            }
            h.clock_phase = (h.clock_phase + eff) % sps;
            for (auto& q : h.queue) q.value -= static_cast<long>(eff);
            produced += count;
            pos += eff;
            done += eff;
            if (eff < span) break; // output full
        }
        return done;
    };
    size_t t = 0;
    bool full = false;
    while (pos < n_in && !full) {
        // no chunk -- and no tag of a chunk -- is started without room for at least one output: the runtime does not call
        // processBulk() with an empty output span (the tag's special cases :160-195 consume an item unconditionally)
        if (produced >= out_cap) break;
        while (t < n_tags_in && tags_in[t].index < pos) ++t; // tags inside consumed specials
        if (t < n_tags_in && tags_in[t].index == pos) {
            gr4pm_tag tag = tags_in[t++];
            long adjust = 0;
            if (tag.flags & GR4PM_TAG_SYNCWORD) { // :130-203
                size_t new_cp = h.reset_clock_phase;
                h.scale = 1.0f / tag.amplitude;
                float time_est = tag.time_est;
                if (time_est < 0.0f) { // :148-156
                    new_cp = (new_cp + 1) % sps;
                    time_est += 1.0f;
                    tag.phase = static_cast<float>(static_cast<double>(tag.phase) - tag.freq);
                }
                if (h.clock_phase == 0 && new_cp == 1) { // :160-189
                    SymRun r;
                    r.in0 = static_cast<long long>(pos);
                    r.out0 = static_cast<unsigned>(produced);
                    r.count = 1;
                    r.arm = static_cast<unsigned>(h.pfb_arm); // arm not yet updated (:199)
                    r.scale = h.scale;
                    runs.push_back(r);
                    while (!h.queue.empty() && h.queue.front().value < half) {
                        publish(h.queue.front().tag, produced);
                        h.queue.pop_front();
                    }
                    ++produced;
                    ++new_cp;
                    for (auto& q : h.queue) --q.value;
                    adjust = -1;
                    ++pos;
                } else if (h.clock_phase == 1 && new_cp == 0) { // :192-195
                    ++pos;
                    ++new_cp;
                }
                h.clock_phase = new_cp;
                const float a = std::round(static_cast<float>(h.num_arms) * time_est);
                h.pfb_arm = std::min(static_cast<size_t>(a), h.num_arms - 1); // :199-202
            }
            h.queue.push_back({ static_cast<long>(h.delay) + adjust, tag }); // :204-205
        }
        size_t end = n_in;
        if (t < n_tags_in && tags_in[t].index < end) end = std::max<size_t>(pos, tags_in[t].index);
        if (end > pos) {
            const size_t want = end - pos;
            if (advance(want) < want) full = true;
        }
        if (produced >= out_cap && pos < n_in) full = true;
    }
}

} // namespace hostlogic
} // namespace gr4pm
