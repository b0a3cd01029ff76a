// packet_receiver.hip -- gr::packet_modem::PacketReceiver (packet_receiver.hpp:34-147,191-247) as
// one native object over the block-level C ABI: the reference composes the blocks in a C++
// flowgraph and lets the multi-threaded scheduler run them concurrently (benchmarks/README.md:8-26);
// here three stages -- detector | gate + frequency correction + symbol filter + wipe-off |
// Costas loop (+ PayloadMetadataInsert / SyncwordRemove / LLR decoder) -- run on three HIP streams
// driven by the caller's thread and two worker threads, several batches in flight.
// Host code only: every kernel is reached through the gr4pm_* entry points.
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

using namespace gr4pm;

namespace {

constexpr int kSlots = 4; // detector | stage 1 | stage 2 | held by the caller

struct Slot {
    // inputs of the batch
    const gr4pm_c64* in = nullptr;
    size_t n_in = 0;
    const gr4pm_c64* delayed = nullptr; // the delayed stream read in place, or nullptr: y below
    gr4pm_c64* out_symbols = nullptr;   // caller's buffers
    size_t out_cap = 0;
    float* out_llr = nullptr;
    size_t llr_cap = 0;
    uint64_t packet_length = 0; // parsed_header answer for every packet (0: "invalid_header")
    // products
    gr4pm_status status = GR4PM_OK;
    char error[256] = { 0 };
    size_t consumed = 0, n_symbols = 0, n_llr = 0;
    uint64_t base = 0; // absolute index of the first item of this batch's delayed stream
    std::vector<gr4pm_tag> det_tags, tags, sym_tags;
    std::vector<uint8_t> accepted;
    std::vector<gr4pm_header_msg> msgs;
    std::vector<gr4pm_packet_tag> packet_tags, data_tags, llr_tags;
    size_t n_det = 0, n_tags = 0, n_sym_tags = 0, n_packet_tags = 0, n_llr_tags = 0, ignored = 0;
    DevBuf<gr4pm_c64> y, sym, w, pm, z, data;
};

template <typename T>
class Channel { // blocking FIFO between two threads
    std::deque<T> q_;
    std::mutex m_;
    std::condition_variable cv_;

public:
    void push(T v)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            q_.push_back(v);
        }
        cv_.notify_one();
    }
    T pop()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop_front();
        return v;
    }
    size_t size()
    {
        std::lock_guard<std::mutex> l(m_);
        return q_.size();
    }
};

} // namespace

struct gr4pm_packet_receiver {
    gr4pm_packet_receiver_params p;
    hipStream_t streams[3] = { nullptr, nullptr, nullptr };
    gr4pm_syncword_detection* sd = nullptr;
    gr4pm_syncword_detection_filter* sdf = nullptr;
    gr4pm_rotator* cfc = nullptr;
    gr4pm_symbol_filter* symf = nullptr;
    gr4pm_syncword_wipeoff* wipe = nullptr;
    gr4pm_costas_loop* costas = nullptr;
    gr4pm_payload_metadata_insert* pmi = nullptr;
    gr4pm_syncword_remove* remove = nullptr;
    gr4pm_constellation_llr_decoder* llr = nullptr;
    uint64_t hist = 0;
    std::deque<gr4pm_header_msg> hdr_fifo; // gate -> PayloadMetadataInsert (stage 1 -> 2, in the slot)
    Slot slots[kSlots];
    Channel<int> free_slots, to_stage1, to_stage2, done;
    std::thread workers[2];
    int held = -1; // slot whose result the caller is looking at
    size_t inflight = 0;

    void fail(Slot& s, gr4pm_status st)
    {
        s.status = st;
        std::strncpy(s.error, gr4pm_last_error(), sizeof(s.error) - 1);
    }
    void stage0(Slot& s, const gr4pm_c64* next_in, size_t next_n);
    void stage1(Slot& s);
    void stage2(Slot& s);
};

void gr4pm_packet_receiver::stage0(Slot& s, const gr4pm_c64* next_in, size_t next_n)
{
    if (next_in) (void)gr4pm_syncword_detection_hint_next(sd, next_in, next_n, next_n);
    gr4pm_c64* out = nullptr;
    if (!s.delayed) {
        if (s.y.n < s.n_in && s.y.alloc(s.n_in) != GR4PM_OK) return fail(s, GR4PM_ERR_NOMEM);
        out = s.y.p;
    }
    size_t n_done = 0, n_tags = 0;
    const gr4pm_status st = gr4pm_syncword_detection_process(sd, s.in, s.n_in, s.n_in, out, s.n_in, &n_done,
                                                             s.det_tags.data(), s.det_tags.size(), &n_tags);
    if (st != GR4PM_OK) return fail(s, st);
    s.consumed = n_done;
    s.n_det = n_tags;
    s.base = gr4pm_syncword_detection_items_consumed(sd) - n_done;
}

void gr4pm_packet_receiver::stage1(Slot& s)
{
    if (s.status != GR4PM_OK) return;
    const gr4pm_c64* y = s.delayed ? s.delayed : s.y.p;
    // SyncwordDetectionFilter: the samples pass unchanged, the tags are gated
    std::vector<uint64_t> idx(s.n_det);
    s.msgs.assign(std::max<size_t>(s.n_det, 1), gr4pm_header_msg{ s.packet_length, s.packet_length == 0 ? 1 : 0 });
    for (size_t i = 0; i < s.n_det; ++i) idx[i] = s.base + s.det_tags[i].index;
    s.accepted.assign(std::max<size_t>(s.n_det, 1), 0);
    size_t used = 0;
    gr4pm_status st = gr4pm_syncword_detection_filter_gate(sdf, idx.data(), s.n_det, s.msgs.data(), s.n_det, 1,
                                                           s.accepted.data(), &used);
    if (st != GR4PM_OK) return fail(s, st);
    s.n_tags = 0;
    for (size_t i = 0; i < s.n_det; ++i)
        if (s.accepted[i]) s.tags[s.n_tags++] = s.det_tags[i];
    const size_t cap = s.consumed / p.samples_per_symbol + s.n_tags + 2;
    if (s.sym.n < cap && s.sym.alloc(cap) != GR4PM_OK) return fail(s, GR4PM_ERR_NOMEM);
    if (s.w.n < cap && s.w.alloc(cap) != GR4PM_OK) return fail(s, GR4PM_ERR_NOMEM);
    size_t n_out_tags = 0, consumed = 0, produced = 0;
    st = gr4pm_cfc_symbol_filter_process(cfc, symf, y, s.consumed, s.sym.p, cap, s.tags.data(), s.n_tags,
                                         s.sym_tags.data(), s.sym_tags.size(), &n_out_tags, &consumed, &produced);
    if (st != GR4PM_OK) return fail(s, st);
    s.n_sym_tags = n_out_tags;
    s.n_symbols = produced;
    st = gr4pm_syncword_wipeoff_process(wipe, s.sym.p, produced, s.w.p, s.sym_tags.data(), n_out_tags);
    if (st != GR4PM_OK) return fail(s, st);
}

void gr4pm_packet_receiver::stage2(Slot& s)
{
    if (s.status != GR4PM_OK) return;
    if (!p.soft_bits) {
        if (s.out_cap < s.n_symbols) {
            set_error("out_cap %zu < %zu symbols", s.out_cap, s.n_symbols);
            return fail(s, GR4PM_INSUFFICIENT_OUTPUT_ITEMS);
        }
        const gr4pm_status st = gr4pm_costas_loop_process(costas, s.w.p, s.n_symbols, s.n_symbols, s.out_symbols,
                                                          s.sym_tags.data(), nullptr, s.n_sym_tags);
        if (st != GR4PM_OK) fail(s, st);
        return;
    }
    // the symbol filter may hold a tag of the last samples back until the next batch: the
    // message of every accepted tag waits in a FIFO until its tag arrives here
    for (size_t i = 0; i < s.n_det; ++i)
        if (s.accepted[i]) hdr_fifo.push_back(s.msgs[i]);
    std::vector<gr4pm_header_msg> hdrs(std::max<size_t>(s.n_sym_tags, 1));
    for (size_t i = 0; i < s.n_sym_tags; ++i) {
        hdrs[i] = hdr_fifo.front();
        hdr_fifo.pop_front();
    }
    const size_t n = s.n_symbols;
    if (s.pm.n < n + 1 && s.pm.alloc(n + 1) != GR4PM_OK) return fail(s, GR4PM_ERR_NOMEM);
    if (s.z.n < n + 1 && s.z.alloc(n + 1) != GR4PM_OK) return fail(s, GR4PM_ERR_NOMEM);
    if (s.data.n < n + 1 && s.data.alloc(n + 1) != GR4PM_OK) return fail(s, GR4PM_ERR_NOMEM);
    s.packet_tags.resize(3 * s.n_sym_tags + 8);
    s.data_tags.resize(s.packet_tags.size());
    s.llr_tags.resize(s.packet_tags.size());
    size_t n_pt = 0, consumed = 0, produced = 0, used = 0, ignored = 0;
    gr4pm_status st = gr4pm_payload_metadata_insert_process(pmi, s.w.p, n, s.pm.p, n + 1, s.sym_tags.data(),
                                                            s.n_sym_tags, hdrs.data(), s.n_sym_tags, 1,
                                                            s.packet_tags.data(), s.packet_tags.size(), &n_pt,
                                                            &consumed, &produced, &used, &ignored);
    if (st != GR4PM_OK) return fail(s, st);
    s.n_packet_tags = n_pt;
    s.ignored = ignored;
    if (s.out_cap < produced) {
        set_error("out_cap %zu < %zu symbols", s.out_cap, produced);
        return fail(s, GR4PM_INSUFFICIENT_OUTPUT_ITEMS);
    }
    st = gr4pm_costas_loop_process_packets(costas, s.pm.p, produced, s.out_symbols, s.packet_tags.data(), n_pt);
    if (st != GR4PM_OK) return fail(s, st);
    s.n_symbols = produced;
    size_t n_dt = 0, n_data = 0;
    st = gr4pm_syncword_remove_process(remove, s.out_symbols, produced, s.data.p, s.packet_tags.data(), n_pt,
                                       s.data_tags.data(), s.data_tags.size(), &n_dt, &n_data);
    if (st != GR4PM_OK) return fail(s, st);
    size_t n_lt = 0, n_llr = 0;
    st = gr4pm_constellation_llr_decoder_process(llr, s.data.p, n_data, s.out_llr, s.llr_cap, s.data_tags.data(), n_dt,
                                                 s.llr_tags.data(), s.llr_tags.size(), &n_lt, &n_llr);
    if (st != GR4PM_OK) return fail(s, st);
    s.n_llr_tags = n_lt;
    s.n_llr = n_llr;
}

extern "C" {

gr4pm_status gr4pm_packet_receiver_create(const gr4pm_packet_receiver_params* p, gr4pm_packet_receiver** out)
{
    if (!p || !out || p->samples_per_symbol == 0 || p->max_items < 2048) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_packet_receiver;
    if (!h) return GR4PM_ERR_NOMEM;
    h->p = *p;
    auto bail = [&](gr4pm_status st) {
        gr4pm_packet_receiver_destroy(h);
        return st;
    };
    for (auto& s : h->streams)
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return bail(GR4PM_ERR_HIP);
    const size_t sps = p->samples_per_symbol;
    // packet_receiver.hpp:60-74: RRC taps normalised to unit RMS norm (float accumulation)
    std::vector<float> rrc(((sps * 11) | 1));
    const size_t n_rrc = gr4pm_firdes_root_raised_cosine(1.0, static_cast<double>(sps), 1.0, 0.35, sps * 11, rrc.data());
    rrc.resize(n_rrc);
    float norm = 0.0f;
    for (float v : rrc) norm += v * v;
    norm = std::sqrt(norm);
    for (float& v : rrc) v /= norm;
    static const uint8_t syncword[64] = { // the 64-bit CCSDS syncword 0x034776C7272895B0, packet_receiver.hpp:37-44
        0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 1, 1, 1,
        0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0 };
    const gr4pm_c64 bpsk[2] = { { 1.0f, 0.0f }, { -1.0f, 0.0f } };
    gr4pm_syncword_detection_params sp{};
    sp.fft_size = 2048;
    sp.samples_per_symbol = sps;
    sp.rrc_taps = rrc.data();
    sp.n_rrc_taps = rrc.size();
    sp.syncword = syncword;
    sp.n_syncword = 64;
    sp.constellation = bpsk;
    sp.n_constellation = 2;
    sp.min_freq_bin = -p->syncword_freq_bins;
    sp.max_freq_bin = p->syncword_freq_bins;
    sp.time_threshold = 768;
    sp.power_threshold = p->syncword_threshold;
    sp.n_channels = 1;
    sp.max_items = p->max_items;
    sp.stream = h->streams[0];
    gr4pm_status st = gr4pm_syncword_detection_create(&sp, &h->sd); // :76-83
    if (st != GR4PM_OK) return bail(st);
    h->hist = 2 * 768 + 1;
    gr4pm_syncword_detection_filter_params fp{ sps, 64, 128, h->streams[1] }; // :84-85
    if ((st = gr4pm_syncword_detection_filter_create(&fp, &h->sdf)) != GR4PM_OK) return bail(st);
    gr4pm_rotator_params rp{ 1, 0.0f, (rrc.size() - 1) / 2 + sps, 1, h->streams[1] }; // :94-95
    if ((st = gr4pm_rotator_create(&rp, &h->cfc)) != GR4PM_OK) return bail(st);
    const size_t arms = 32; // :96-115
    std::vector<float> pfb(((arms * sps * 11) | 1));
    const size_t n_pfb = gr4pm_firdes_root_raised_cosine(static_cast<double>(arms) / static_cast<double>(norm),
                                                         static_cast<double>(arms * sps), 1.0, 0.35, arms * sps * 11,
                                                         pfb.data());
    pfb.resize(n_pfb - 1); // the design is odd-length: drop the last tap (:108-110)
    gr4pm_symbol_filter_params fsp{ sps, pfb.data(), pfb.size(), arms, rrc.size() - 1, 0, h->streams[1] };
    if ((st = gr4pm_symbol_filter_create(&fsp, &h->symf)) != GR4PM_OK) return bail(st);
    float bipolar[64];
    for (int i = 0; i < 64; ++i) bipolar[i] = syncword[i] ? -1.0f : 1.0f; // :117-122
    gr4pm_syncword_wipeoff_params wp{ bipolar, 64, h->streams[1] };
    if ((st = gr4pm_syncword_wipeoff_create(&wp, &h->wipe)) != GR4PM_OK) return bail(st);
    gr4pm_costas_loop_params cp{ 0.01, p->soft_bits ? 1 : p->costas_constellation, 1, h->streams[2] }; // :125
    if ((st = gr4pm_costas_loop_create(&cp, &h->costas)) != GR4PM_OK) return bail(st);
    if (p->soft_bits) {
        gr4pm_payload_metadata_insert_params pp{ 64, 128, 0.02, 0.01, 0.005, h->streams[2] }; // :123-124
        if ((st = gr4pm_payload_metadata_insert_create(&pp, &h->pmi)) != GR4PM_OK) return bail(st);
        gr4pm_syncword_remove_params sr{ 64, h->streams[2] }; // :126
        if ((st = gr4pm_syncword_remove_create(&sr, &h->remove)) != GR4PM_OK) return bail(st);
        gr4pm_constellation_llr_decoder_params lp{ 0.7f, 2, h->streams[2] }; // :127-130
        if ((st = gr4pm_constellation_llr_decoder_create(&lp, &h->llr)) != GR4PM_OK) return bail(st);
    }
    const size_t tags_cap = std::max<size_t>(p->tags_cap, 64);
    for (int i = 0; i < kSlots; ++i) {
        h->slots[i].det_tags.resize(tags_cap);
        h->slots[i].tags.resize(tags_cap);
        h->slots[i].sym_tags.resize(tags_cap + 64);
        h->free_slots.push(i);
    }
    if (p->pipelined) {
        h->workers[0] = std::thread([h] {
            for (;;) {
                const int i = h->to_stage1.pop();
                if (i < 0) break;
                h->stage1(h->slots[i]);
                h->to_stage2.push(i);
            }
            h->to_stage2.push(-1);
        });
        h->workers[1] = std::thread([h] {
            for (;;) {
                const int i = h->to_stage2.pop();
                if (i < 0) break;
                h->stage2(h->slots[i]);
                h->done.push(i);
            }
        });
    }
    *out = h;
    return GR4PM_OK;
}

void gr4pm_packet_receiver_destroy(gr4pm_packet_receiver* h)
{
    if (!h) return;
    if (h->workers[0].joinable()) {
        h->to_stage1.push(-1);
        h->workers[0].join();
        h->workers[1].join();
    }
    gr4pm_syncword_detection_destroy(h->sd);
    gr4pm_syncword_detection_filter_destroy(h->sdf);
    gr4pm_rotator_destroy(h->cfc);
    gr4pm_symbol_filter_destroy(h->symf);
    gr4pm_syncword_wipeoff_destroy(h->wipe);
    gr4pm_costas_loop_destroy(h->costas);
    gr4pm_payload_metadata_insert_destroy(h->pmi);
    gr4pm_syncword_remove_destroy(h->remove);
    gr4pm_constellation_llr_decoder_destroy(h->llr);
    for (auto s : h->streams)
        if (s) (void)hipStreamDestroy(s);
    delete h;
}

size_t gr4pm_packet_receiver_inflight(const gr4pm_packet_receiver* h) { return h ? h->inflight : 0; }

gr4pm_status gr4pm_packet_receiver_submit(gr4pm_packet_receiver* h, const gr4pm_c64* in, size_t n_in,
                                          const gr4pm_c64* delayed, const gr4pm_c64* next_in, size_t next_n,
                                          uint64_t packet_length, gr4pm_c64* out_symbols, size_t out_cap,
                                          float* out_llr, size_t llr_cap)
{
    if (!h || !in || !out_symbols || (h->p.soft_bits && !out_llr)) return GR4PM_ERR_INVALID;
    if (h->inflight >= static_cast<size_t>(kSlots - 1)) {
        set_error("%zu batches in flight: collect one first", h->inflight);
        return GR4PM_ERR_INVALID;
    }
    const int i = h->free_slots.pop();
    Slot& s = h->slots[i];
    s.in = in;
    s.n_in = n_in;
    s.delayed = delayed;
    s.out_symbols = out_symbols;
    s.out_cap = out_cap;
    s.out_llr = out_llr;
    s.llr_cap = llr_cap;
    s.packet_length = packet_length;
    s.status = GR4PM_OK;
    s.error[0] = 0;
    s.consumed = s.n_symbols = s.n_llr = s.n_det = s.n_tags = s.n_sym_tags = s.n_packet_tags = s.n_llr_tags = 0;
    h->stage0(s, next_in, next_n);
    ++h->inflight;
    if (h->p.pipelined) {
        h->to_stage1.push(i);
    } else {
        h->stage1(s);
        h->stage2(s);
        h->done.push(i);
    }
    return GR4PM_OK;
}

gr4pm_status gr4pm_packet_receiver_collect(gr4pm_packet_receiver* h, gr4pm_packet_receiver_result* r)
{
    if (!h || !r) return GR4PM_ERR_INVALID;
    if (h->held >= 0) { // the previous result is handed back now
        h->free_slots.push(h->held);
        h->held = -1;
    }
    if (h->inflight == 0) {
        set_error("nothing in flight");
        return GR4PM_ERR_INVALID;
    }
    const int i = h->done.pop();
    --h->inflight;
    h->held = i;
    Slot& s = h->slots[i];
    std::memset(r, 0, sizeof(*r));
    r->consumed = s.consumed;
    r->n_symbols = s.n_symbols;
    r->n_llr = s.n_llr;
    r->detector_tags = s.det_tags.data();
    r->n_detector_tags = s.n_det;
    r->accepted = s.accepted.data();
    r->tags = s.sym_tags.data();
    r->n_tags = s.n_sym_tags;
    r->packet_tags = s.packet_tags.data();
    r->n_packet_tags = s.n_packet_tags;
    r->llr_tags = s.llr_tags.data();
    r->n_llr_tags = s.n_llr_tags;
    r->ignored_syncwords = s.ignored;
    r->symbols = s.out_symbols;
    r->llr = s.out_llr;
    if (s.status != GR4PM_OK) set_error("%s", s.error);
    return s.status;
}

} // extern "C"
