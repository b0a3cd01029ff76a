// hostlogic/tail_plan.hpp -- the symbol-rate tail of the receiver behind the Costas loop as ONE table (round 6).
//
// packet_receiver.hpp:126-147,208-240 wires SyncwordRemove -> ConstellationLLRDecoder (QPSK) -> AdditiveScrambler ->
// HeaderPayloadSplit -> { HeaderFecDecoder | BinarySlicer -> PackBits -> CrcCheck }.  Block by block that is seven
// passes over the symbol stream (a gather, the LLR scaling, the descrambler, two more gathers, the slicer / packer), each
// reading and writing 8 .. 16 bytes per symbol.  None of the blocks in between looks at a VALUE: which symbol ends up
// where, and with which scrambler bit, is decided by the tags alone.  Every block's state machine is still replayed on
// the host exactly as before (hostlogic/packet_control.hpp, the scrambler's run list) -- this file only COMPOSES their
// span tables into one: for every run of Costas-loop output symbols that reaches the header decoder or the packer,
// where it comes from, where its LLRs / bits go and where the descrambler's sequence stands.  One kernel
// (k_tail_fused, header_blocks.hip) then reads each symbol once and writes header LLRs (floats, bit for bit the
// unfused chain's) or packed payload bytes.
#pragma once
#include <algorithm>
#include <vector>

#include "base.hpp"
#include "packet_control.hpp"

namespace gr4pm {
namespace hostlogic {

// items [start, start + len) of the scrambler's input use its sequence at index phase, phase + 1, ...
// (additive_scrambler.hpp:76-94 replayed over the reset tags: gr4pm_additive_scrambler_process's run list)
struct ScrambleRun {
    unsigned long long start, len, phase;
};

// AdditiveScrambler's state as far as the run list needs it (additive_scrambler.hpp:49,64,76-83): the LFSR output
// depends only on the number of items since the last reset, so it is tabulated once (csrc/header_blocks.hip): `prefix`
// items, then a cycle of `period` items -- or, where no repeat was found, just `table_len` items.
struct ScrState {
    uint64_t count = 0;              // :64, 0 = never reset by count
    uint64_t position = 0;           // items since the last reset (_current_count, :49)
    uint64_t prefix = 0, period = 1;
    bool cyclic = true;
    uint64_t table_len = 0;
};
// the host half of AdditiveScrambler::processOne over one call: runs of items between resets (tag resets :78-80, count
// resets :81); advances the block's position
inline gr4pm_status scramble_runs(ScrState& h, size_t n, const uint64_t* reset_index, size_t n_resets,
                                  std::vector<ScrambleRun>& runs)
{
    size_t pos = 0, t = 0;
    while (pos < n) {
        while (t < n_resets && reset_index[t] < pos) ++t;
        if (t < n_resets && reset_index[t] == pos) {
            h.position = 0;
            ++t;
        }
        if (h.count != 0 && h.position == h.count) h.position = 0;
        size_t end = n;
        if (t < n_resets) end = std::min<size_t>(end, reset_index[t]);
        if (h.count != 0) end = std::min<size_t>(end, pos + static_cast<size_t>(h.count - h.position));
        if (!h.cyclic && h.position + (end - pos) > h.table_len) {
            set_error("LFSR output needed beyond the %llu tabulated items", static_cast<unsigned long long>(h.table_len));
            return GR4PM_ERR_INVALID;
        }
        runs.push_back({ pos, end - pos, h.position });
        h.position += end - pos;
        pos = end;
    }
    if (h.cyclic && h.position >= h.prefix + h.period) // keep the counter small
        h.position = h.prefix + (h.position - h.prefix) % h.period;
    return GR4PM_OK;
}

struct TailSpan {
    unsigned long long src;   // first symbol of the run in the Costas loop's output
    unsigned long long dst;   // kind 0: index of its first LLR in the batch's header LLR stream;
                              // kind 1: index of its first BIT in the batch's payload stream (MSB-first bytes)
    unsigned long long phase; // scrambler sequence index of its first LLR
    unsigned n_sym;           // symbols (two LLRs each: QPSK)
    unsigned kind;            // 0 header, 1 payload
};

#if defined(__HIPCC__)
#define GR4PM_HOST_DEVICE __host__ __device__
#else
#define GR4PM_HOST_DEVICE
#endif
// What k_tail_fused does with one symbol / one output byte, as functions of their own: the kernel (header_blocks.hip) and
// the sanitizer build of this host logic (tests/hostlogic) run the very same lines.
GR4PM_HOST_DEVICE inline unsigned long long scr_index(unsigned long long q, unsigned long long prefix, unsigned long long period)
{
    return q >= prefix + period ? prefix + (q - prefix) % period : q; // (k_scramble's table index)
}
// one QPSK symbol -> its two descrambled LLRs: scale * re, scale * im (constellation_llr_decoder.hpp:106-116), negated where
// the LFSR bit is 1 (additive_scrambler.hpp:92-93); q: sequence index of the first of the two
GR4PM_HOST_DEVICE inline void tail_llr_pair(float re, float im, float scale, const uint8_t* seq, unsigned long long q,
                                            unsigned long long prefix, unsigned long long period, float& l0, float& l1)
{
    const float a = scale * re, b = scale * im;
    l0 = seq[scr_index(q, prefix, period)] ? -a : a;
    l1 = seq[scr_index(q + 1, prefix, period)] ? -b : b;
}
// byte B of the packer's stream as far as the payload run `sp` owns it: BinarySlicer<true> (binary_slicer.hpp:28-33: llr <
// 0) + PackBits<MSB> (pack_bits.hpp: the first bit is the top one).  sym: the Costas output, interleaved floats.  The first
// and the last byte of a run may be partial (a run that continues a packet of the batch before starts where that batch's
// bits end): mask says which bits are the run's.
GR4PM_HOST_DEVICE inline void tail_payload_byte(const TailSpan& sp, unsigned long long B, const float* sym, float scale,
                                                const uint8_t* seq, unsigned long long prefix, unsigned long long period,
                                                unsigned& v, unsigned& mask)
{
    const unsigned long long bit0 = sp.dst, bit_end = sp.dst + 2ull * sp.n_sym;
    const unsigned long long lo = B * 8 > bit0 ? B * 8 : bit0, hi = B * 8 + 8 < bit_end ? B * 8 + 8 : bit_end;
    v = mask = 0;
    for (unsigned long long b = lo; b < hi; b += 2) {
        const unsigned long long k = (b - bit0) >> 1;
        float l0, l1;
        tail_llr_pair(sym[2 * (sp.src + k)], sym[2 * (sp.src + k) + 1], scale, seq, sp.phase + 2ull * k, prefix, period, l0, l1);
        const unsigned sh = 6u - static_cast<unsigned>(b - B * 8); // bit position 0 of the byte is 0x80
        v |= ((l0 < 0.0f ? 1u : 0u) << (sh + 1)) | ((l1 < 0.0f ? 1u : 0u) << sh);
        mask |= 3u << sh;
    }
}

// sr:   SyncwordRemove's spans, symbols: Costas output [src, src + len) -> data [dst, dst + len), ascending in dst
// scr:  the descrambler's runs over the LLR stream (LLR j = 2 * data symbol + (0 | 1)), ascending, covering it
// hps:  HeaderPayloadSplit's spans over the LLR stream (header_spans -> header stream, payload_spans -> payload stream)
// payload_bit0: where the batch's payload stream starts in the packer's bit stream (the bits of an unfinished packet
//       carried from the batches before lie in front of it)
// Returns false when a boundary does not fall between two symbols (an odd LLR index: cannot happen with the tags
// PayloadMetadataInsert publishes; the caller then runs the blocks one by one).
inline bool compose_tail(const std::vector<CopySpan>& sr, const std::vector<ScrambleRun>& scr, const HpsReplay& hps,
                         unsigned long long header_llr0, unsigned long long payload_bit0, std::vector<TailSpan>& out)
{
    out.clear();
    auto emit = [&](const std::vector<CopySpan>& spans, unsigned kind, unsigned long long dst0) -> bool {
        size_t is = 0, ir = 0; // cursors into sr and scr: every list ascends, the spans of one kind ascend in src
        for (const CopySpan& s : spans) {
            unsigned long long a = s.src, end = s.src + s.len, d = dst0 + s.dst;
            if ((a | end | s.dst) & 1ull) return false;
            while (a < end) {
                while (is < sr.size() && 2 * (sr[is].dst + sr[is].len) <= a) ++is;
                while (ir < scr.size() && scr[ir].start + scr[ir].len <= a) ++ir;
                if (is >= sr.size() || ir >= scr.size() || 2 * sr[is].dst > a || scr[ir].start > a) return false;
                const unsigned long long stop = std::min({ end, 2 * (sr[is].dst + sr[is].len), scr[ir].start + scr[ir].len });
                if ((stop & 1ull) || stop <= a) return false;
                unsigned long long left = (stop - a) / 2, sym = sr[is].src + (a / 2 - sr[is].dst), ph = scr[ir].phase + (a - scr[ir].start);
                while (left) { // (n_sym is 32 bits wide)
                    const unsigned long long m = std::min<unsigned long long>(left, 1ull << 30);
                    out.push_back({ sym, d, ph, static_cast<unsigned>(m), kind });
                    sym += m, d += 2 * m, ph += 2 * m, left -= m;
                }
                a = stop;
            }
        }
        return true;
    };
    return emit(hps.header_spans, 0, header_llr0) && emit(hps.payload_spans, 1, payload_bit0);
}

} // namespace hostlogic
} // namespace gr4pm
