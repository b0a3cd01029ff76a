// hostlogic/sdf_gate.hpp -- SyncwordDetectionFilter (syncword_detection_filter.hpp:54-210) without HIP: the decisions
// of one processBulk() call (process_plan: which items pass) and the same decisions over a list of tags for
// device-resident chains (gate / gate_resolve: which detections open a packet).
#pragma once
#include <algorithm>

#include "base.hpp"

namespace gr4pm {
namespace hostlogic {

struct SdfState {
    size_t sps = 4, syncword_size = 64, header_size = 128, allowed_margin = 16; // :44-47
    bool in_packet = false;                                                     // :35-37
    size_t position = 0, block_until = 0;
    // gate(): absolute item index where the current packet span ends (exclusive)
    bool gate_in_packet = false;
    uint64_t gate_start = 0, gate_end = 0;
    bool gate_end_known = false;
    size_t gate_hdr_idx = static_cast<size_t>(-1); // per-tag mode: header slot of the open packet
};

// One processBulk() call.  The items that pass are in[0 .. *consumed) -> out[0 .. *consumed): at most two runs, which
// the caller copies (`runs`: up to two (offset, length) pairs, n_runs of them).
inline gr4pm_status sdf_process_plan(SdfState& h, size_t n_in, size_t out_cap, int head_tag_flags,
                                     const gr4pm_header_msg* headers, size_t n_headers, size_t n_ignored,
                                     size_t* consumed_, size_t* headers_consumed, size_t* ignored_consumed,
                                     int* tag_out_flags, CopySpan runs[2], int* n_runs)
{
    *consumed_ = *headers_consumed = *ignored_consumed = 0;
    *tag_out_flags = 0;
    *n_runs = 0;
    auto copy = [&](size_t off, size_t n) {
        if (n) runs[(*n_runs)++] = CopySpan{ off, off, n };
    };
    if (head_tag_flags) { // :75-105
        int of = 0;
        bool new_in_packet = false;
        if ((head_tag_flags & GR4PM_TAG_SYNCWORD) && !h.in_packet) {
            new_in_packet = true;
            of |= GR4PM_TAG_SYNCWORD;
        }
        if (head_tag_flags & GR4PM_TAG_OTHER) of |= GR4PM_TAG_OTHER;
        if (new_in_packet) {
            h.in_packet = true;
            h.position = 0;
            h.block_until = 0;
        }
        *tag_out_flags = of;
    }
    if (!h.in_packet) { // :107-130
        const size_t n = std::min(n_in, out_cap);
        copy(0, n);
        *consumed_ = n;
        return GR4PM_OK;
    }
    if (h.block_until == 0 && n_headers > 0) { // :134-153
        *headers_consumed = 1;
        if (headers[0].invalid_header) {
            h.block_until = 1;
        } else {
            if (headers[0].packet_length == 0) {
                set_error("received packet_length = 0"); // :143-145
                return GR4PM_ERR_INVALID;
            }
            const size_t payload_symbols = (headers[0].packet_length + 4) * 4;
            h.block_until = h.sps * (h.header_size + h.syncword_size - h.allowed_margin + payload_symbols);
        }
    }
    if (h.block_until == 0 && n_ignored > 0) { // :157-160
        *ignored_consumed = 1;
        h.block_until = 1;
    }
    size_t consumed = 0;
    const size_t allowed = h.sps * (h.syncword_size + h.header_size + h.allowed_margin);
    if (h.position < allowed) { // :166-172
        const size_t n = std::min({ n_in, out_cap, allowed - h.position });
        copy(0, n);
        h.position += n;
        consumed = n;
    }
    if (h.position >= allowed && h.block_until != 0) { // :174-185
        const size_t n = std::min(n_in, out_cap) - consumed;
        copy(consumed, n);
        h.position += n;
        consumed += n;
        if (h.position >= h.block_until) h.in_packet = false;
    }
    *consumed_ = consumed;
    return GR4PM_OK;
}

// where the span of a packet with this header ends, relative to its start (:146-151; 1: invalid / ignored, :139,:159)
inline gr4pm_status sdf_block_until(const SdfState& h, const gr4pm_header_msg& m, uint64_t* block_until)
{
    *block_until = 1;
    if (!m.invalid_header) {
        if (m.packet_length == 0) {
            set_error("received packet_length = 0");
            return GR4PM_ERR_INVALID;
        }
        *block_until = h.sps * (h.header_size + h.syncword_size - h.allowed_margin + (m.packet_length + 4) * 4);
    }
    return GR4PM_OK;
}

inline gr4pm_status sdf_gate(SdfState& h, const uint64_t* tag_index, size_t n_tags, const gr4pm_header_msg* headers,
                             size_t n_headers, int headers_per_tag, uint8_t* accepted, size_t* headers_used)
{
    if (headers_per_tag && n_headers != n_tags) return GR4PM_ERR_INVALID;
    *headers_used = 0;
    const uint64_t allowed = h.sps * (h.syncword_size + h.header_size + h.allowed_margin); // :164-165
    size_t hu = 0;
    for (size_t i = 0; i < n_tags; ++i) {
        const uint64_t at = tag_index[i];
        if (h.gate_in_packet) {
            // resolve the pending header before looking at this tag: the reference cannot get
            // past `allowed` items of the packet without it (:166-185)
            if (!h.gate_end_known &&
                (headers_per_tag ? (h.gate_hdr_idx < n_headers && headers[h.gate_hdr_idx].invalid_header != 2)
                                 : hu < n_headers)) {
                const gr4pm_header_msg& m = headers_per_tag ? headers[h.gate_hdr_idx] : headers[hu];
                ++hu;
                uint64_t block_until = 1;
                if (const gr4pm_status st = sdf_block_until(h, m, &block_until); st != GR4PM_OK) return st;
                h.gate_end = h.gate_start + std::max<uint64_t>(allowed, block_until);
                h.gate_end_known = true;
            }
            // inside the first `allowed` items the span is open whatever the header says
            const uint64_t end = h.gate_end_known ? h.gate_end : h.gate_start + allowed;
            if (at < end) {
                accepted[i] = 0; // :83-88 dropped while _in_packet
                continue;
            }
            if (!h.gate_end_known) {
                // a tag beyond `allowed` with the header still unknown: the caller has not
                // supplied the message the reference would be waiting for
                set_error("parsed_header message missing for the packet at item %llu",
                          static_cast<unsigned long long>(h.gate_start));
                return GR4PM_INSUFFICIENT_INPUT_ITEMS;
            }
            h.gate_in_packet = false;
        }
        accepted[i] = 1; // :85-97
        h.gate_in_packet = true;
        h.gate_start = at;
        h.gate_end_known = false;
        h.gate_hdr_idx = i;
    }
    if (headers_per_tag && h.gate_in_packet && !h.gate_end_known && h.gate_hdr_idx < n_headers &&
        headers[h.gate_hdr_idx].invalid_header != 2) {
        // resolve the last accepted tag of this call now: its header will not be re-presented
        // (a pending one, invalid_header == 2, is resolved later by sdf_gate_resolve)
        uint64_t block_until = 1;
        if (const gr4pm_status st = sdf_block_until(h, headers[h.gate_hdr_idx], &block_until); st != GR4PM_OK) return st;
        h.gate_end = h.gate_start + std::max<uint64_t>(allowed, block_until);
        h.gate_end_known = true;
        ++hu;
    }
    h.gate_hdr_idx = static_cast<size_t>(-1);
    *headers_used = hu;
    return GR4PM_OK;
}

inline gr4pm_status sdf_gate_resolve(SdfState& h, const gr4pm_header_msg& msg)
{
    if (!h.gate_in_packet || h.gate_end_known) return GR4PM_OK; // nothing is waiting
    const uint64_t allowed = h.sps * (h.syncword_size + h.header_size + h.allowed_margin);
    uint64_t block_until = 1;
    if (const gr4pm_status st = sdf_block_until(h, msg, &block_until); st != GR4PM_OK) return st;
    h.gate_end = h.gate_start + std::max<uint64_t>(allowed, block_until);
    h.gate_end_known = true;
    return GR4PM_OK;
}

} // namespace hostlogic
} // namespace gr4pm
