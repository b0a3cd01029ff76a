// hostlogic/packet_control.hpp -- the symbol-rate control blocks without HIP: PayloadMetadataInsert
// (payload_metadata_insert.hpp:77-307), SyncwordRemove (syncword_remove.hpp:39-105) and HeaderPayloadSplit
// (header_payload_split.hpp:38-135).  Each is a state machine over tags plus a copy of item spans; the replay of one
// call -- chunk by chunk, chunks cut at the tags exactly as the runtime presents them -- yields the span table the
// gather kernels work from and the output tags.
#pragma once
#include <algorithm>
#include <vector>

#include "base.hpp"

namespace gr4pm {
namespace hostlogic {

struct PmiState {
    size_t syncword_size = 64, header_size = 128;
    double syncword_bw = 0, header_bw = 0, payload_bw = 0;
    bool in_packet = false;     // payload_metadata_insert.hpp:37
    uint64_t position = 0;      // :38
    size_t payload_symbols = 0; // :39
    uint64_t num_packet = 0;    // :40
    // headers_per_tag mode: the message that answers the syncword which opened the packet
    gr4pm_header_msg held{};
    bool has_held = false;
};
struct PmiReplay {
    std::vector<CopySpan> spans;
    size_t n_pub = 0, consumed = 0, produced = 0, headers_used = 0, ignored = 0;
    bool tag_overflow = false;
};
inline gr4pm_status pmi_replay(PmiState& h, size_t n_in, size_t out_cap, const gr4pm_tag* tags_in, size_t n_tags_in,
                               const gr4pm_header_msg* headers, size_t n_headers, int headers_per_tag,
                               gr4pm_packet_tag* tags_out, size_t tags_cap, PmiReplay& rp)
{
    std::vector<CopySpan>& spans = rp.spans;
    size_t& n_pub = rp.n_pub;
    size_t& hdr = rp.headers_used;
    size_t& ignored = rp.ignored;
    bool& tag_overflow = rp.tag_overflow;
    auto publish = [&](const gr4pm_packet_tag& t) {
        if (tags_out && n_pub < tags_cap) tags_out[n_pub] = t;
        else tag_overflow = true;
        ++n_pub;
    };
    size_t ipos = 0, opos = 0; // items consumed / produced so far
    spans.reserve(n_tags_in + 8);
    auto pass = [&](size_t want) { // move up to `want` items from the input to the output
        // (syncword, header and payload of a packet follow each other on both sides: one span, not three)
        if (want && !spans.empty() && spans.back().src + spans.back().len == ipos && spans.back().dst + spans.back().len == opos)
            spans.back().len += want;
        else
            spans.push_back({ ipos, opos, want });
        ipos += want;
        opos += want;
        h.position += want;
    };
    const size_t sw = h.syncword_size, hs = h.header_size;
    size_t t = 0;
    bool stop = false;
    while (ipos < n_in && !stop) {
        // one processBulk() call: the chunk [ipos, end) with at most one tag, at its head
        while (t < n_tags_in && tags_in[t].index < ipos) ++t;
        const bool head_tag = t < n_tags_in && tags_in[t].index == ipos;
        const bool has_tag = head_tag && (tags_in[t].flags & GR4PM_TAG_SYNCWORD); // syncword_amplitude key
        size_t end = n_in;
        if (const size_t u = head_tag ? t + 1 : t; u < n_tags_in) end = std::min<size_t>(end, tags_in[u].index);
        const size_t chunk0 = ipos;
        if (has_tag) { // :96-149
            if (!h.in_packet) {
                h.in_packet = true;
                h.position = 0;
                ++h.num_packet;
                gr4pm_packet_tag pt{};
                pt.index = opos;
                pt.kind = GR4PM_PKT_SYNCWORD;
                pt.constellation = 0; // the syncword modulation has been wiped off: pure pilot
                pt.loop_bandwidth = h.syncword_bw;
                pt.syncword = tags_in[t];
                publish(pt);
                if (headers_per_tag) {
                    h.held = headers[t];
                    h.has_held = headers[t].invalid_header != 2; // 2 = pending, see ..._resolve
                }
            } else {
                ++ignored;
            }
        }
        if (!h.in_packet) { // :150-169
            ipos = end;
            if (head_tag) ++t;
            continue;
        }
        while (opos < out_cap && ipos < end) { // :174
            if (h.position < sw) pass(std::min({ end - ipos, out_cap - opos, static_cast<size_t>(sw - h.position) }));
            if (h.position == sw) { // :186-194
                gr4pm_packet_tag pt{};
                pt.index = opos;
                pt.kind = GR4PM_PKT_HEADER_START;
                pt.constellation = 2;
                pt.loop_bandwidth = h.header_bw;
                publish(pt);
            }
            if (sw <= h.position && h.position < sw + hs)
                pass(std::min({ end - ipos, out_cap - opos, static_cast<size_t>(sw + hs - h.position) }));
            if (h.position == sw + hs && opos < out_cap && ipos < end) {
                if (headers_per_tag ? h.has_held : hdr < n_headers) { // :207-242
                    const gr4pm_header_msg msg = headers_per_tag ? h.held : headers[hdr];
                    h.has_held = false;
                    if (msg.invalid_header) {
                        h.in_packet = false;
                        ipos = end;
                        ++hdr;
                        break;
                    }
                    const uint64_t packet_length = msg.packet_length;
                    if (packet_length == 0) {
                        set_error("received packet_length = 0"); // :224-226
                        return GR4PM_ERR_INVALID;
                    }
                    h.payload_symbols = static_cast<size_t>((packet_length + 4) * 4); // + CRC-32, QPSK
                    gr4pm_packet_tag pt{};
                    pt.index = opos;
                    pt.kind = GR4PM_PKT_PAYLOAD;
                    pt.constellation = -1;
                    pt.loop_bandwidth = h.payload_bw;
                    pt.packet_length = packet_length;
                    pt.payload_symbols = h.payload_symbols;
                    pt.payload_bits = 2 * static_cast<uint64_t>(h.payload_symbols);
                    publish(pt);
                    pass(std::min({ end - ipos, out_cap - opos, h.payload_symbols }));
                    ++hdr;
                } else { // :243-247: return and wait for the header
                    stop = true;
                    break;
                }
            }
            if (sw + hs < h.position && h.position < sw + hs + h.payload_symbols)
                pass(std::min({ end - ipos, out_cap - opos,
                                static_cast<size_t>(sw + hs + h.payload_symbols - h.position) }));
            if (h.position >= sw + hs + h.payload_symbols) { // :263-267
                h.in_packet = false;
                ipos = end;
            }
        }
        if (head_tag && ipos > chunk0) ++t;
        if (ipos == chunk0) stop = true; // no progress: waiting for a header or output full
        else if (opos >= out_cap && ipos < end) stop = true;
    }
    // spans of length 0 come from the min() above when a stage has nothing to move
    spans.erase(std::remove_if(spans.begin(), spans.end(), [](const CopySpan& c) { return c.len == 0; }), spans.end());
    rp.consumed = ipos;
    rp.produced = opos;
    return GR4PM_OK;
}

struct SrState {
    size_t syncword_size = 64;
    bool in_syncword = false; // syncword_remove.hpp:25
    size_t position = 0;      // :26
};
struct SrReplay {
    std::vector<CopySpan> spans;
    size_t produced = 0, n_pub = 0;
    bool tag_overflow = false;
};
inline void sr_replay(SrState& h, size_t n, const gr4pm_packet_tag* tags_in, size_t n_tags_in, gr4pm_packet_tag* tags_out,
                      size_t tags_cap, SrReplay& rp)
{
    std::vector<CopySpan>& spans = rp.spans;
    size_t& opos = rp.produced;
    size_t& n_pub = rp.n_pub;
    bool& tag_overflow = rp.tag_overflow;
    size_t pos = 0, t = 0;
    while (pos < n) {
        while (t < n_tags_in && tags_in[t].index < pos) ++t;
        // the tags sitting on the chunk's first item (one merged map in the reference)
        size_t t1 = t;
        bool syncword = false;
        while (t1 < n_tags_in && tags_in[t1].index == pos) syncword |= tags_in[t1++].kind == GR4PM_PKT_SYNCWORD;
        const size_t end = t1 < n_tags_in ? std::min<size_t>(n, tags_in[t1].index) : n;
        if (!h.in_syncword && t1 > t) { // :51-64
            if (syncword) {
                h.in_syncword = true;
                h.position = 0;
            } else {
                for (size_t u = t; u < t1; ++u) {
                    if (tags_out && n_pub < tags_cap) {
                        tags_out[n_pub] = tags_in[u];
                        tags_out[n_pub].index = opos;
                    } else {
                        tag_overflow = true;
                    }
                    ++n_pub;
                }
            }
        }
        size_t from = pos;
        if (h.in_syncword) { // :67-74
            const size_t m = std::min(end - pos, h.syncword_size - h.position);
            from += m;
            h.position += m;
            if (h.position >= h.syncword_size) h.in_syncword = false;
        }
        if (!h.in_syncword && end > from) { // :76-81
            spans.push_back({ from, opos, end - from });
            opos += end - from;
        }
        t = t1;
        pos = end;
    }
}

struct HpsState {
    size_t header_size = 128;
    bool in_payload = false;    // header_payload_split.hpp:25
    uint64_t position = 0;      // :26
    uint64_t payload_items = 0; // :27
};
struct HpsReplay {
    std::vector<CopySpan> header_spans, payload_spans;
    size_t n_header = 0, n_payload = 0, n_header_tags = 0, n_payload_tags = 0;
    bool tag_overflow = false;
};
inline gr4pm_status hps_replay(HpsState& h, size_t n, const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                               gr4pm_packet_tag* header_tags, gr4pm_packet_tag* payload_tags, size_t tags_cap,
                               HpsReplay& rp)
{
    using FSpan = CopySpan;
    std::vector<CopySpan>& hs = rp.header_spans;
    std::vector<CopySpan>& ps = rp.payload_spans;
    size_t& hp = rp.n_header;
    size_t& pp = rp.n_payload;
    size_t& nht = rp.n_header_tags;
    size_t& npt = rp.n_payload_tags;
    bool& overflow = rp.tag_overflow;
    size_t pos = 0, t = 0;
    while (pos < n) {
        while (t < n_tags_in && tags_in[t].index < pos) ++t;
        size_t t1 = t;
        while (t1 < n_tags_in && tags_in[t1].index == pos) ++t1;
        const size_t end = t1 < n_tags_in ? std::min<size_t>(n, tags_in[t1].index) : n;
        for (size_t u = t; u < t1; ++u) // :68-82
            if (tags_in[u].kind == GR4PM_PKT_PAYLOAD) {
                if (h.in_payload || h.position != h.header_size) {
                    set_error("received unexpected payload_bits tag"); // :75-78
                    return GR4PM_ERR_INVALID;
                }
                h.in_payload = true;
                h.position = 0;
                h.payload_items = tags_in[u].payload_bits;
            }
        for (size_t u = t; u < t1; ++u) { // :83-87
            gr4pm_packet_tag o = tags_in[u];
            if (h.in_payload) {
                o.index = pp;
                if (payload_tags && npt < tags_cap) payload_tags[npt] = o;
                else overflow = true;
                ++npt;
            } else {
                o.index = hp;
                if (header_tags && nht < tags_cap) header_tags[nht] = o;
                else overflow = true;
                ++nht;
            }
        }
        t = t1;
        size_t cur = pos;
        while (cur < end) {
            if (!h.in_payload && h.position == h.header_size) h.position = 0; // :90-95
            if (!h.in_payload) { // :97-109
                const size_t m = std::min<size_t>(end - cur, h.header_size - h.position);
                hs.push_back({ cur, hp, m });
                hp += m;
                cur += m;
                h.position += m;
            } else { // :110-123
                const size_t m = std::min<size_t>(end - cur, static_cast<size_t>(h.payload_items - h.position));
                ps.push_back({ cur, pp, m });
                pp += m;
                cur += m;
                h.position += m;
                if (h.position >= h.payload_items) {
                    h.in_payload = false;
                    h.position = 0;
                }
            }
        }
        pos = end;
    }
    // neighbouring spans of one output are contiguous on both sides: merge them
    auto merge = [](std::vector<FSpan>& v) {
        std::vector<FSpan> o;
        for (const auto& s : v) {
            if (s.len == 0) continue;
            if (!o.empty() && o.back().src + o.back().len == s.src && o.back().dst + o.back().len == s.dst)
                o.back().len += s.len;
            else
                o.push_back(s);
        }
        v.swap(o);
    };
    merge(hs);
    merge(ps);
    return GR4PM_OK;
}

} // namespace hostlogic
} // namespace gr4pm
