// packet_receiver.hip -- gr::packet_modem::PacketReceiver (packet_receiver.hpp:34-147,191-247) as
// one native object over the block-level C ABI: the reference composes the blocks in a C++
// flowgraph and lets the multi-threaded scheduler run them concurrently (benchmarks/README.md:8-26);
// here three stages -- detector | gate + frequency correction + symbol filter + wipe-off |
// Costas loop (+ PayloadMetadataInsert / SyncwordRemove / LLR decoder) -- run on three HIP streams
// driven by the caller's thread and two worker threads, several batches in flight.
// Host code only: every kernel is reached through the gr4pm_* entry points.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "hostlogic/slot_queue.hpp"

using namespace gr4pm;

namespace {

// detector | pass A (decode_headers) | stage 1 | stage 1b | stage 2 | stage 3 (decode_headers) | held by the caller, and
// (round 6) three more that wait in front of stage 1b while their phasor chains run side by side on the CFC's own streams
// (one packet per 2^20 samples: a chain is 16.7 ms long, a batch every 4 ms)
constexpr int kSlots = 10;
static_assert(kSlots <= GR4PM_CFC_PLANS, "a slot keeps its CFC plan from stage 1 to stage 1b: the plan ring must cover every slot");

struct Slot {
    uint64_t seq = 0; // ordinal of the batch (1, 2, ...)
    // inputs of the batch
    const gr4pm_c64* in = nullptr;
    size_t n_in = 0;
    const gr4pm_c64* delayed = nullptr; // the delayed stream read in place, or nullptr: y below
    gr4pm_c64* out_symbols = nullptr;   // caller's buffers
    size_t out_cap = 0;
    float* out_llr = nullptr;
    size_t llr_cap = 0;
    uint64_t packet_length = 0; // parsed_header answer for every packet (0: "invalid_header")
    int plan = -1;              // rotation plan made by stage 1, used by stage 1b
    // decode_headers, pass A -> stage 1: per detection of this batch, the header pass A decoded for
    // it ({0, 1} when none) or invalid_header == 2 while its window is still on its way; and every
    // header pass A finished during this batch (sorted by detection index), for detections of
    // earlier batches that were still pending when the gate saw them
    std::vector<gr4pm_header_msg> pre_msgs;
    std::vector<std::pair<uint64_t, gr4pm_header_msg>> newly_known;
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
    DevBuf<gr4pm_c64> y, sym, pm, z, data; // sym: symbol filter output, wiped off in place
    // decode_headers
    std::vector<gr4pm_header_msg> hdrs;      // per symbol-rate tag, for PayloadMetadataInsert
    std::vector<gr4pm_header_msg> opened;    // messages of the packets PayloadMetadataInsert opened (stage 2 -> 3)
    bool has_resolve = false;
    gr4pm_header_msg resolve{};
    std::vector<gr4pm_header_msg> header_messages;
    std::vector<int32_t> packet_type;
    std::vector<uint64_t> packet_lengths;
    size_t header_mismatches = 0, n_packet_bytes = 0;
    uint8_t* out_packets = nullptr;
    size_t packets_cap = 0;
    DevBuf<float> payload_llr;
    size_t n_payload_llr = 0;
    std::vector<gr4pm_packet_tag> payload_tags;
    size_t n_payload_tags = 0;
    // symbol PDU tap (soft_bits): SyncwordRemove's output is s.data[0 .. n_data), cut into header / payload pieces
    size_t n_data = 0;
    // packets_only: SyncwordRemove's spans of this batch (stage 2 -> stage 3) and the LLR decoder's scale
    std::vector<hostlogic::CopySpan> sr_spans;
    float llr_scale = 0.0f;
    std::vector<hostlogic::CopySpan> pm_spans; // decode_headers: PayloadMetadataInsert's spans (the PLL reads through them)
    std::vector<gr4pm_symbol_pdu> pdus;
    size_t pdu_resyncs = 0;
};

// slot indices between the stage threads: hostlogic/slot_queue.hpp (fixed ring, push() cannot throw)
using SlotRing = hostlogic::SlotQueue<16>;

// pass A of the header loop: one window of W items per detection, gathered into a compact stream.
// starts[j] is relative to y[0]; negative starts read the saved tail of the previous batch, which
// `head` holds in front of the first W items of y.
__global__ __launch_bounds__(256) void k_gather_windows(const gr4pm_c64* __restrict__ y,
                                                        const gr4pm_c64* __restrict__ head, long long tail_len,
                                                        const long long* __restrict__ starts, unsigned W,
                                                        gr4pm_c64* __restrict__ out)
{
    const long long st = starts[blockIdx.x];
    const gr4pm_c64* src = st >= 0 ? y + st : head + (st + tail_len);
    gr4pm_c64* dst = out + static_cast<size_t>(blockIdx.x) * W;
    for (unsigned i = threadIdx.x; i < W; i += blockDim.x) dst[i] = src[i];
}

// everything one of the two header loops needs: descrambler -> split -> FEC decoder -> parser,
// with the header LLRs of an unfinished codeword carried to the next batch
struct HeaderLoop {
    gr4pm_additive_scrambler* scr = nullptr;
    gr4pm_header_payload_split* split = nullptr;
    gr4pm_header_fec_decoder* fec = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<float> desc, hdr, pay, acc; // acc: header LLRs waiting for a complete codeword
    size_t acc_n = 0;
    std::vector<uint8_t> bytes, invalid;
    std::vector<gr4pm_packet_tag> hdr_tags, pay_tags;
    size_t n_pay = 0, n_pay_tags = 0;

    gr4pm_status create(const char* alist, hipStream_t s)
    {
        stream = s;
        gr4pm_additive_scrambler_params sp{ 0x4001, 0x18E38, 16, 0, 1, s }; // packet_receiver.hpp:131-135
        GR4PM_TRY(gr4pm_additive_scrambler_create(&sp, &scr));
        gr4pm_header_payload_split_params hp{ 256, s }; // :136-137
        GR4PM_TRY(gr4pm_header_payload_split_create(&hp, &split));
        gr4pm_header_fec_decoder_params fp{ alist, 25, s }; // :138
        return gr4pm_header_fec_decoder_create(&fp, &fec);
    }
    void destroy()
    {
        gr4pm_additive_scrambler_destroy(scr);
        gr4pm_header_payload_split_destroy(split);
        gr4pm_header_fec_decoder_destroy(fec);
    }
    // llr + tags in; appends the parsed_header messages of the codewords finished by this batch.  pay_dst (room for
    // n + 1 items): where the payload LLRs go instead of this loop's own buffer
    gr4pm_status run(const float* llr, size_t n, const gr4pm_packet_tag* tags, size_t n_tags,
                     std::vector<gr4pm_header_msg>& msgs, std::vector<int32_t>& ptype, float* pay_dst = nullptr)
    {
        n_pay = n_pay_tags = 0;
        std::vector<uint64_t> resets;
        for (size_t i = 0; i < n_tags; ++i)
            if (tags[i].kind == GR4PM_PKT_HEADER_START) resets.push_back(tags[i].index);
        if (desc.n < n + 1) GR4PM_TRY(desc.alloc(n + 1));
        if (hdr.n < n + 1) GR4PM_TRY(hdr.alloc(n + 1));
        if (!pay_dst) {
            if (pay.n < n + 1) GR4PM_TRY(pay.alloc(n + 1));
            pay_dst = pay.p;
        }
        GR4PM_TRY(gr4pm_additive_scrambler_process(scr, llr, n, desc.p, resets.data(), resets.size()));
        hdr_tags.resize(n_tags + 1);
        pay_tags.resize(n_tags + 1);
        size_t n_hdr = 0, nht = 0;
        GR4PM_TRY(gr4pm_header_payload_split_process(split, desc.p, n, hdr.p, &n_hdr, pay_dst, &n_pay, tags, n_tags,
                                                     hdr_tags.data(), &nht, pay_tags.data(), &n_pay_tags, n_tags + 1));
        GR4PM_TRY(reserve_acc(n_hdr));
        if (n_hdr) GR4PM_HIP_TRY(hipMemcpyAsync(acc.p + acc_n, hdr.p, n_hdr * sizeof(float), hipMemcpyDeviceToDevice, stream));
        return decode_acc(n_hdr, msgs, ptype);
    }
    // room for n_hdr more header LLRs behind the acc_n waiting ones
    gr4pm_status reserve_acc(size_t n_hdr)
    {
        if (acc.n >= acc_n + n_hdr + 256) return GR4PM_OK;
        DevBuf<float> bigger;
        GR4PM_TRY(bigger.alloc((acc_n + n_hdr + 256) * 2));
        if (acc_n) GR4PM_HIP_TRY(hipMemcpyAsync(bigger.p, acc.p, acc_n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        GR4PM_HIP_TRY(hipStreamSynchronize(stream));
        std::swap(acc.p, bigger.p);
        std::swap(acc.n, bigger.n);
        return GR4PM_OK;
    }
    // n_hdr header LLRs have been put behind the waiting ones (in stream order): decode every complete codeword
    gr4pm_status decode_acc(size_t n_hdr, std::vector<gr4pm_header_msg>& msgs, std::vector<int32_t>& ptype)
    {
        acc_n += n_hdr;
        const size_t cw = acc_n / 256;
        bytes.resize(std::max<size_t>(cw, 1) * 4);
        invalid.resize(std::max<size_t>(cw, 1));
        GR4PM_TRY(gr4pm_header_fec_decoder_process(fec, acc.p, cw, bytes.data(), invalid.data()));
        const size_t rest = acc_n - cw * 256;
        if (cw && rest) { // move the unfinished codeword to the front (regions cannot overlap: rest < 256 <= cw * 256)
            GR4PM_HIP_TRY(hipMemcpyAsync(acc.p, acc.p + cw * 256, rest * sizeof(float), hipMemcpyDeviceToDevice, stream));
            GR4PM_HIP_TRY(hipStreamSynchronize(stream));
        }
        acc_n = rest;
        const size_t at = msgs.size();
        msgs.resize(at + cw);
        ptype.resize(at + cw);
        gr4pm_header_parse(bytes.data(), invalid.data(), cw, msgs.data() + at, ptype.data() + at);
        return GR4PM_OK;
    }
    // packets_only (round 6): the same loop with the blocks' host halves alone and ONE kernel over the Costas loop's output
    // (hostlogic/tail_plan.hpp, k_tail_fused): header LLRs go straight behind the waiting ones, payload bits straight into
    // the packer's byte stream at bit `payload_bit0` of `packed`.  sr_spans: SyncwordRemove's spans of this batch.
    std::vector<hostlogic::ScrambleRun> scr_runs;
    std::vector<hostlogic::TailSpan> tail_spans;
    DevBuf<hostlogic::TailSpan> tail_table;
    gr4pm_status run_fused(const gr4pm_c64* costas_out, float llr_scale, const std::vector<hostlogic::CopySpan>& sr_spans,
                           size_t n_llr, const gr4pm_packet_tag* tags, size_t n_tags, uint8_t* packed, size_t payload_bit0,
                           std::vector<gr4pm_header_msg>& msgs, std::vector<int32_t>& ptype)
    {
        n_pay = n_pay_tags = 0;
        std::vector<uint64_t> resets;
        for (size_t i = 0; i < n_tags; ++i)
            if (tags[i].kind == GR4PM_PKT_HEADER_START) resets.push_back(tags[i].index);
        GR4PM_TRY(gr4pm::scrambler_plan(scr, n_llr, resets.data(), resets.size(), scr_runs));
        hdr_tags.resize(n_tags + 1);
        pay_tags.resize(n_tags + 1);
        hostlogic::HpsReplay rp;
        GR4PM_TRY(gr4pm::header_payload_split_plan(split, n_llr, tags, n_tags, hdr_tags.data(), pay_tags.data(), n_tags + 1, rp));
        n_pay = rp.n_payload;
        n_pay_tags = rp.n_payload_tags;
        GR4PM_TRY(reserve_acc(rp.n_header));
        if (!hostlogic::compose_tail(sr_spans, scr_runs, rp, acc_n, payload_bit0, tail_spans)) {
            set_error("packets_only: the tag stream behind the Costas loop does not compose (a boundary inside a symbol)");
            return GR4PM_ERR_INTERNAL;
        }
        GR4PM_TRY(gr4pm::tail_fused(scr, tail_table, tail_spans, costas_out, llr_scale, acc.p, packed, stream));
        return decode_acc(rp.n_header, msgs, ptype);
    }
};

} // namespace

#ifdef GR4PM_TIMING
// wall time a stage spends working and waiting for its input queue (make EXTRA=-DGR4PM_TIMING)
struct StageClock {
    const char* name;
    double busy = 0, idle = 0;
    int n = 0;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void got_work()
    {
        const auto now = std::chrono::steady_clock::now();
        idle += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
    }
    void done_work()
    {
        const auto now = std::chrono::steady_clock::now();
        busy += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
        if (++n % 8 == 0) {
            fprintf(stderr, "[gr4pm timing] %s: busy %.0f us, idle %.0f us (mean of the last 8)\n", name, busy / 8, idle / 8);
            busy = idle = 0;
        }
    }
};
#define CLK_GOT(c) (c).got_work()
#define CLK_DONE(c) (c).done_work()
#else
struct StageClock { const char* name; };
#define CLK_GOT(c) (void)(c)
#define CLK_DONE(c) (void)(c)
#endif

struct gr4pm_packet_receiver {
    StageClock clk[6] = { { "stage 0" }, { "stage 1" }, { "stage 2" }, { "stage 3" }, { "stage 1b" }, { "stage A" } };
    gr4pm_packet_receiver_params p;
    // [3]: header loop + payload tail (stage 3); [4]: symbol filter + wipe-off (stage 1b);
    // [5]: pass A of decode_headers (stage A)
    hipStream_t streams[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    gr4pm_syncword_detection* sd = nullptr;
    gr4pm_syncword_detection_filter* sdf = nullptr;
    gr4pm_rotator* cfc = nullptr;
    gr4pm_symbol_filter* symf = nullptr;
    gr4pm_syncword_wipeoff* wipe = nullptr;
    gr4pm_costas_loop* costas = nullptr;
    gr4pm_payload_metadata_insert* pmi = nullptr;
    gr4pm_syncword_remove* remove = nullptr;
    gr4pm_constellation_llr_decoder* llr = nullptr;
    // HeaderPayloadSplit<c64> of the symbol tap (header_payload_split.hpp:30-45): state carried across batches
    struct {
        bool in_payload = false;
        uint64_t position = 0, payload_items = 0;
    } tap;
    gr4pm_symbol_pdu_fn pdu_fn = nullptr;
    void* pdu_user = nullptr;
    gr4pm_zmq_pub* pdu_pub[2] = { nullptr, nullptr }; // gr4pm_packet_receiver_publish_symbol_pdus: headers, payloads
    std::vector<gr4pm_c64> pdu_host, pdu_acc[2];
    bool pdu_open[2] = { false, false }; // the sink has seen the first piece of the PDU it is collecting
    gr4pm_status split_symbol_pdus(Slot& s, size_t n_data, size_t n_dt);
    uint64_t hist = 0;
    std::deque<gr4pm_header_msg> hdr_fifo; // gate -> PayloadMetadataInsert (stage 1 -> 2, in the slot)
    // ---- decode_headers: pass A (stage 1 thread) ----
    static constexpr unsigned kPre = 64, kW = 912; // window: 64 items before the tag, 228 symbols in all
    gr4pm_rotator* a_cfc = nullptr;
    gr4pm_symbol_filter* a_symf = nullptr;
    gr4pm_syncword_wipeoff* a_wipe = nullptr;
    gr4pm_payload_metadata_insert* a_pmi = nullptr;
    gr4pm_costas_loop* a_costas = nullptr;
    gr4pm_syncword_remove* a_remove = nullptr;
    gr4pm_constellation_llr_decoder* a_llr = nullptr;
    HeaderLoop a_loop, b_loop; // b_loop: the real chain's own header loop (stage 2 thread)
    gr4pm_crc_check* crc = nullptr;
    DevBuf<gr4pm_c64> a_tail, a_head, a_compact, a_sym, a_pm, a_z, a_data;
    DevBuf<float> a_llrbuf;
    DevBuf<uint8_t> a_packed;
    DevBuf<long long> a_starts;
    std::vector<uint64_t> awaiting_idx;     // detections whose window continues in the next batch
    std::vector<gr4pm_tag> awaiting_tags;
    std::deque<uint64_t> a_order;           // detections inside pass A whose header is not out yet
    size_t a_fifo = 0;                      // invalid messages owed to pass A's PayloadMetadataInsert
    bool pending_real = false;              // an accepted packet waits for its header
    uint64_t pending_idx = 0;
    std::deque<gr4pm_header_msg> s1_fifo;   // accepted tags' messages on their way to stage 2
    std::mutex s1_fifo_mutex;               // stage 1 pushes and patches, stage 1b pops
    // batches stage 1 has started / stage 1b has finished (or seen failed).  Stage 1 patches a pending message inside
    // s1_fifo only after stage 1b is done with every EARLIER batch: whether the entry is still in the fifo then says
    // whether its tag left the symbol filter in the batch before (resolve path) or leaves it later (patch) -- the same
    // answer with and without the pipeline (without the wait it depended on which thread got there first: the two
    // forms of the receiver then cut the symbol stream differently around a pending header; tools/stress_receiver.py)
    uint64_t stage1_started = 0, stage1_earlier = 0;
    uint64_t stage1b_done = 0;
    bool stage1b_abort = false; // destroy(): nobody waits for stage 1b any more
    std::mutex stage1b_mutex;
    std::condition_variable stage1b_cv;
    // ---- decode_headers: stage 2 thread ----
    // symbols (and their tags / messages) that PayloadMetadataInsert could not take at the end of a batch because the
    // header of the packet they belong to arrives with the next batch (stage2_decode)
    static constexpr size_t kPmCarryMax = 1024;
    DevBuf<gr4pm_c64> pm_carry, pm_join;
    size_t pm_carry_n = 0;
    std::vector<gr4pm_tag> pm_carry_tags;
    std::vector<gr4pm_header_msg> pm_carry_hdrs;
    std::deque<gr4pm_header_msg> used_msgs; // given to PayloadMetadataInsert, not yet verified
    std::deque<gr4pm_header_msg> early_hdrs; // decoded by the chain before pass A's message for the packet arrived
    DevBuf<float> soft;                     // payload soft bits of packets not finished yet
    size_t soft_n = 0;
    std::deque<uint64_t> payload_bits;      // their lengths
    DevBuf<uint8_t> packed;
    // packets_only: the packer's byte stream of a batch starts with the bits of the packet the batch before left
    // unfinished -- carried as BYTES (two buffers in turn: the tail of one batch's stream is copied to the front of the next
    // one's; the last byte may be half full, k_tail_fused completes it)
    DevBuf<uint8_t> packed2[2];
    int packed_cur = 0;
    size_t carry_bits = 0;
    Slot slots[kSlots];
    SlotRing free_slots, to_stageA, to_stage1, to_stage1b, to_stage2, to_stage3, done;
    std::thread workers[5];
    int held = -1; // slot whose result the caller is looking at
    size_t inflight = 0;
    // what the PLL stage needs to know: is a correlator launch of a LATER batch running (or about to) beside it?
    std::atomic<uint64_t> n_submitted{ 0 };
    std::atomic<int> n_ahead{ 0 }; // announced and not yet submitted
    int costas_form = 0;            // the PLL's kernel form while that is the case (gr4pm_costas_loop_set_small_footprint)

    void fail(Slot& s, gr4pm_status st)
    {
        s.status = st;
        std::strncpy(s.error, gr4pm_last_error(), sizeof(s.error) - 1);
    }
    void stage0(Slot& s, const gr4pm_c64* next_in, size_t next_n);
    void stageA(Slot& s);
    void stage1(Slot& s);
    void stage1b(Slot& s);
    void stage2(Slot& s);
    gr4pm_status predecode(Slot& s, const gr4pm_c64* y);
    gr4pm_status stage1_decode(Slot& s, const gr4pm_c64* y);
    gr4pm_status filter_and_wipe(Slot& s);
    gr4pm_status stage2_decode(Slot& s);
    gr4pm_status stage3_decode(Slot& s);
    gr4pm_status stage3_soft(Slot& s);
    void stage3(Slot& s)
    {
        if (s.status != GR4PM_OK || !p.soft_bits) return;
        gr4pm_status st;
        try {
            st = p.decode_headers ? stage3_decode(s) : stage3_soft(s);
        } catch (...) {
            drop_unfinished_payload();
            (void)hipStreamSynchronize(streams[3]);
            throw; // hostlogic::run_stage fails the batch
        }
        (void)hipStreamSynchronize(streams[3]);
        if (st != GR4PM_OK) {
            drop_unfinished_payload();
            fail(s, st);
        }
    }
    // A batch that fails in stage3_decode has popped payload lengths without consuming their soft bits, or has not
    // appended its own: `soft` / `soft_n` / `payload_bits` no longer describe one stream.  The stage lives on after a
    // failed batch (run_stage), so the carried state starts empty again -- the packets in flight across the failed
    // batch are lost with it; everything behind is sliced from consistent state.
    void drop_unfinished_payload()
    {
        soft_n = 0;
        carry_bits = 0;
        payload_bits.clear();
    }
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

// Stage A (decode_headers only): pass A of the two-pass header loop, one batch ahead of the gate
void gr4pm_packet_receiver::stageA(Slot& s)
{
    if (s.status != GR4PM_OK || !p.decode_headers) return;
    const gr4pm_status st = predecode(s, s.delayed ? s.delayed : s.y.p);
    if (st != GR4PM_OK) fail(s, st);
}

// Stages 1-3 run with deferred synchronisation (gr4pm_set_deferred_sync): the blocks of a stage
// share one stream, so their kernels queue up behind each other without the host waiting in
// between; every stage waits once, at its end, before the slot moves on.
void gr4pm_packet_receiver::stage1(Slot& s)
{
    stage1_earlier = stage1_started++; // every batch counts, failed ones too (stage 1b counts them as well)
    if (s.status != GR4PM_OK) return;
    const gr4pm_c64* y = s.delayed ? s.delayed : s.y.p;
    struct SyncAtEnd {
        hipStream_t st;
        ~SyncAtEnd() { (void)hipStreamSynchronize(st); }
    } sync_at_end{ streams[1] };
    if (p.decode_headers) {
        const gr4pm_status st = stage1_decode(s, y);
        if (st != GR4PM_OK) fail(s, st);
        return;
    }
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
    st = gr4pm_cfc_symbol_filter_plan(cfc, s.consumed, s.tags.data(), s.n_tags, &s.plan);
    if (st != GR4PM_OK) return fail(s, st);
}

// CoarseFrequencyCorrection -> SymbolFilter (the filter half of the fused call) -> SyncwordWipeoff
gr4pm_status gr4pm_packet_receiver::filter_and_wipe(Slot& s)
{
    const gr4pm_c64* y = s.delayed ? s.delayed : s.y.p;
    const size_t cap = s.consumed / p.samples_per_symbol + s.n_tags + 2;
    if (s.sym.n < cap) GR4PM_TRY(s.sym.alloc(cap));
    size_t n_out_tags = 0, consumed = 0, produced = 0;
    GR4PM_TRY(gr4pm_cfc_symbol_filter_run(cfc, s.plan, symf, y, s.consumed, s.sym.p, cap, s.tags.data(), s.n_tags,
                                          s.sym_tags.data(), s.sym_tags.size(), &n_out_tags, &consumed, &produced));
    s.n_sym_tags = n_out_tags;
    s.n_symbols = produced;
    GR4PM_TRY(gr4pm_syncword_wipeoff_process(wipe, s.sym.p, produced, s.sym.p, s.sym_tags.data(), n_out_tags));
    if (p.decode_headers) {
        // the message of every accepted tag waits in a FIFO (filled, and patched, by stage 1) until
        // the symbol filter lets its tag through
        std::lock_guard<std::mutex> lk(s1_fifo_mutex);
        s.hdrs.assign(std::max<size_t>(n_out_tags, 1), gr4pm_header_msg{ 0, 1 });
        for (size_t i = 0; i < n_out_tags && !s1_fifo.empty(); ++i) {
            s.hdrs[i] = s1_fifo.front();
            s1_fifo.pop_front();
        }
    }
    return GR4PM_OK;
}

// Stage 1b: the symbol filter runs while stage 1 already plans the next batch (the phasor
// checkpoints are a serial, latency-bound kernel; the filter is a whole-chip one)
void gr4pm_packet_receiver::stage1b(Slot& s)
{
    struct Done { // also on the early return and on an exception: stage 1 may be waiting for this batch
        gr4pm_packet_receiver* h;
        ~Done()
        {
            {
                std::lock_guard<std::mutex> lk(h->stage1b_mutex);
                ++h->stage1b_done;
            }
            h->stage1b_cv.notify_all();
        }
    } done{ this };
    if (s.status != GR4PM_OK) return;
    const gr4pm_status st = filter_and_wipe(s);
    (void)hipStreamSynchronize(streams[4]);
    if (st != GR4PM_OK) fail(s, st);
}

void gr4pm_packet_receiver::stage2(Slot& s)
{
    if (s.status != GR4PM_OK) return;
    struct SyncAtEnd {
        hipStream_t st;
        ~SyncAtEnd() { (void)hipStreamSynchronize(st); }
    } sync_at_end{ streams[2] };
    if (p.decode_headers) {
        const gr4pm_status st = stage2_decode(s);
        if (st != GR4PM_OK) fail(s, st);
        return;
    }
    if (!p.soft_bits) {
        if (s.out_cap < s.n_symbols) {
            set_error("out_cap %zu < %zu symbols", s.out_cap, s.n_symbols);
            return fail(s, GR4PM_INSUFFICIENT_OUTPUT_ITEMS);
        }
        // the 32-VGPR form pays beside a correlator launch; with nothing behind this batch (the pipeline is draining)
        // the chip is the PLL's own and the fast form finishes a millisecond earlier
        const bool later_work = n_submitted.load(std::memory_order_relaxed) > s.seq || n_ahead.load(std::memory_order_relaxed) > 0;
        (void)gr4pm_costas_loop_set_small_footprint(costas, later_work ? costas_form : 0);
        const gr4pm_status st = gr4pm_costas_loop_process(costas, s.sym.p, s.n_symbols, s.n_symbols, s.out_symbols,
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
    gr4pm_status st = gr4pm_payload_metadata_insert_process(pmi, s.sym.p, n, s.pm.p, n + 1, s.sym_tags.data(),
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
    s.n_symbols = produced; // SyncwordRemove and the LLR decoder follow in stage 3
}

// The symbol tap's HeaderPayloadSplit<c64> (packet_receiver.hpp:159-162; header_payload_split.hpp:46-135, header_size
// 128, payload_length_key "payload_symbols"), replayed over the tags of SyncwordRemove's output: the runtime presents a
// tag at the head of a chunk, so the stream is walked chunk by chunk between tags.  No items move: the PDUs are spans of
// s.data (TaggedStreamToPdu, :163-164, collects exactly those spans).
gr4pm_status gr4pm_packet_receiver::split_symbol_pdus(Slot& s, size_t n_data, size_t n_dt)
{
    constexpr uint64_t header_size = 128;
    s.n_data = n_data;
    s.pdus.clear();
    s.pdu_resyncs = 0;
    size_t ti = 0;
    uint64_t pos = 0;
    while (pos < n_data) {
        while (ti < n_dt && s.data_tags[ti].index < pos) ++ti; // (tags are ascending)
        while (ti < n_dt && s.data_tags[ti].index == pos) {
            const gr4pm_packet_tag& t = s.data_tags[ti++];
            if (t.kind == GR4PM_PKT_PAYLOAD) { // :70-82
                if (tap.in_payload || tap.position != header_size) {
                    // header_payload_split.hpp:75-78 throws here.  The tap is the receiver's OPTIONAL side output
                    // (zmq_output, packet_receiver.hpp:159): it must not fail the batch's LLRs and packets, and it
                    // must not carry the broken state into the next batches.  The piece in progress is closed as it
                    // is (never `last`, so no sink delivers it), the tap starts over at this tag, the result counts it.
                    ++s.pdu_resyncs;
                }
                tap.in_payload = true;
                tap.position = 0;
                tap.payload_items = t.payload_symbols;
            }
        }
        const uint64_t chunk_end = ti < n_dt ? std::min<uint64_t>(s.data_tags[ti].index, n_data) : n_data;
        if (!tap.in_payload && tap.position == header_size) tap.position = 0; // :90-95: the header did not decode
        const uint64_t want = tap.in_payload ? tap.payload_items - tap.position : header_size - tap.position;
        const uint64_t m = std::min(chunk_end - pos, want);
        gr4pm_symbol_pdu pc{};
        pc.offset = pos;
        pc.length = m;
        pc.kind = tap.in_payload ? 1 : 0;
        pc.first = tap.position == 0;
        tap.position += m;
        pc.last = tap.position >= (tap.in_payload ? tap.payload_items : header_size);
        if (!s.pdus.empty() && s.pdus.back().kind == pc.kind && !s.pdus.back().last && !pc.first &&
            s.pdus.back().offset + s.pdus.back().length == pc.offset) {
            s.pdus.back().length += m; // pieces of one PDU separated only by a tag
            s.pdus.back().last = pc.last;
        } else if (m > 0 || pc.last) {
            s.pdus.push_back(pc);
        }
        pos += m;
        if (tap.in_payload && tap.position >= tap.payload_items) { // :123-126
            tap.in_payload = false;
            tap.position = 0;
        }
        if (m == 0 && chunk_end == pos && ti >= n_dt) break;
    }
    return GR4PM_OK;
}

// soft_bits without decode_headers: SyncwordRemove + ConstellationLLRDecoder as stage 3 (stage 2,
// the tag-driven Costas loop behind PayloadMetadataInsert, is the slowest one of this mode)
gr4pm_status gr4pm_packet_receiver::stage3_soft(Slot& s)
{
    size_t n_dt = 0, n_data = 0, n_lt = 0, n_llr = 0;
    GR4PM_TRY(gr4pm_syncword_remove_process(remove, s.out_symbols, s.n_symbols, s.data.p, s.packet_tags.data(),
                                            s.n_packet_tags, s.data_tags.data(), s.data_tags.size(), &n_dt, &n_data));
    GR4PM_TRY(gr4pm_constellation_llr_decoder_process(llr, s.data.p, n_data, s.out_llr, s.llr_cap, s.data_tags.data(),
                                                      n_dt, s.llr_tags.data(), s.llr_tags.size(), &n_lt, &n_llr));
    s.n_llr_tags = n_lt;
    s.n_llr = n_llr;
    return split_symbol_pdus(s, n_data, n_dt);
}

// pass A (see blocks.py PacketReceiver._predecode, which this mirrors): the header of every detection
gr4pm_status gr4pm_packet_receiver::predecode(Slot& s, const gr4pm_c64* y)
{
#ifdef GR4PM_TIMING
    static double lap_sum[12] = { 0 };
    static int lap_n = 0;
    int lap_i = 0;
    auto lap_t = std::chrono::steady_clock::now();
#define A_LAP()                                                                                              \
    do {                                                                                                     \
        const auto now_ = std::chrono::steady_clock::now();                                                  \
        lap_sum[lap_i++] += std::chrono::duration<double, std::micro>(now_ - lap_t).count();                \
        lap_t = now_;                                                                                        \
    } while (0)
#else
#define A_LAP() do { } while (0)
#endif
    hipStream_t st1 = streams[5];
    s.newly_known.clear();
    const long long n = static_cast<long long>(s.consumed);
    std::vector<long long> starts;
    std::vector<gr4pm_tag> tags;
    std::vector<uint64_t> order;
    for (size_t i = 0; i < awaiting_idx.size(); ++i) { // waiting detections of the last batch first
        starts.push_back(static_cast<long long>(awaiting_idx[i]) - static_cast<long long>(s.base) - kPre);
        tags.push_back(awaiting_tags[i]);
        order.push_back(awaiting_idx[i]);
    }
    awaiting_idx.clear();
    awaiting_tags.clear();
    for (size_t i = 0; i < s.n_det; ++i) {
        const long long st = static_cast<long long>(s.det_tags[i].index) - kPre;
        if (st + kW <= n) {
            starts.push_back(st);
            tags.push_back(s.det_tags[i]);
            order.push_back(s.base + s.det_tags[i].index);
        } else {
            awaiting_idx.push_back(s.base + s.det_tags[i].index);
            awaiting_tags.push_back(s.det_tags[i]);
        }
    }
    const size_t k = starts.size();
    const size_t tail_len = kW + kPre;
    if (k) {
        if (a_starts.n < k) GR4PM_TRY(a_starts.alloc(k * 2));
        if (a_compact.n < k * kW) GR4PM_TRY(a_compact.alloc(k * kW * 2));
        GR4PM_TRY(a_starts.upload_staged(starts.data(), k, st1));
        // [saved tail | first W items of y] for the windows that begin in the previous batch
        GR4PM_HIP_TRY(hipMemcpyAsync(a_head.p, a_tail.p, tail_len * sizeof(gr4pm_c64), hipMemcpyDeviceToDevice, st1));
        GR4PM_HIP_TRY(hipMemcpyAsync(a_head.p + tail_len, y, kW * sizeof(gr4pm_c64), hipMemcpyDeviceToDevice, st1));
        hipLaunchKernelGGL(k_gather_windows, dim3(static_cast<unsigned>(k)), dim3(256), 0, st1, y, a_head.p,
                           static_cast<long long>(tail_len), a_starts.p, kW, a_compact.p);
        GR4PM_HIP_TRY(hipGetLastError());
        A_LAP(); // 0 gather
        for (size_t j = 0; j < k; ++j) tags[j].index = j * kW + kPre;
        const size_t n_items = k * kW, cap = n_items / p.samples_per_symbol + k + 2;
        if (a_sym.n < cap) {
            GR4PM_TRY(a_sym.alloc(cap * 2));
            GR4PM_TRY(a_pm.alloc(cap * 2));
            GR4PM_TRY(a_z.alloc(cap * 2));
            GR4PM_TRY(a_data.alloc(cap * 2));
            GR4PM_TRY(a_llrbuf.alloc(cap * 4));
        }
        std::vector<gr4pm_tag> sym_tags(k + 64);
        size_t n_st = 0, consumed = 0, produced = 0;
        GR4PM_TRY(gr4pm_cfc_symbol_filter_process(a_cfc, a_symf, a_compact.p, n_items, a_sym.p, cap, tags.data(), k,
                                                  sym_tags.data(), sym_tags.size(), &n_st, &consumed, &produced));
        A_LAP(); // 1 cfc + symbol filter
        GR4PM_TRY(gr4pm_syncword_wipeoff_process(a_wipe, a_sym.p, produced, a_sym.p, sym_tags.data(), n_st));
        A_LAP(); // 2 wipe-off
        a_fifo += k; // one "invalid_header" per detection: only syncword + header pass
        std::vector<gr4pm_header_msg> inv(std::max<size_t>(n_st, 1), gr4pm_header_msg{ 0, 1 });
        a_fifo -= std::min(a_fifo, n_st);
        std::vector<gr4pm_packet_tag> ptags(3 * n_st + 8), dtags(3 * n_st + 8), ltags(3 * n_st + 8);
        size_t n_pt = 0, c2 = 0, n_pm = 0, used = 0, ignored = 0;
        // Round 6: pass A the way the packets_only receiver runs its tail -- every block's state machine on the host, the
        // PLL reading PayloadMetadataInsert's input through its span table, then ONE kernel (k_tail_fused) from the PLL's
        // output to the header LLRs: six launches less per batch on the stream whose latency the gate waits for.  Pass A's
        // streams are nobody's result; the header LLRs are the blocks' bit for bit (the composition is the tested one).
        std::vector<hostlogic::CopySpan> a_pm_spans, a_sr_spans;
        GR4PM_TRY(gr4pm::payload_metadata_insert_plan(a_pmi, produced, cap, sym_tags.data(), n_st, inv.data(), n_st, 1,
                                                      ptags.data(), ptags.size(), &n_pt, &c2, &n_pm, &used, &ignored, a_pm_spans));
        A_LAP(); // 3 pmi
        GR4PM_TRY(gr4pm::costas_loop_process_packets_from(a_costas, a_sym.p, a_pm_spans.data(), a_pm_spans.size(), n_pm, a_z.p,
                                                          ptags.data(), n_pt));
        A_LAP(); // 4 costas
        size_t n_dt = 0, n_data = 0, n_lt = 0, n_llr = 0;
        GR4PM_TRY(gr4pm::syncword_remove_plan(a_remove, n_pm, ptags.data(), n_pt, dtags.data(), dtags.size(), &n_dt, &n_data,
                                              a_sr_spans));
        bool all_qpsk = true;
        float a_scale = 0.0f;
        GR4PM_TRY(gr4pm::llr_decoder_plan(a_llr, n_data, dtags.data(), n_dt, ltags.data(), ltags.size(), &n_lt, &n_llr, &all_qpsk,
                                          &a_scale));
        if (!all_qpsk) {
            set_error("pass A: a run behind SyncwordRemove is not QPSK");
            return GR4PM_ERR_INVALID;
        }
        A_LAP(); // 5 remove + llr
        std::vector<gr4pm_header_msg> done;
        std::vector<int32_t> ptype;
        if (a_packed.n < 64) GR4PM_TRY(a_packed.alloc(64)); // (pass A's PayloadMetadataInsert passes no payload: nothing is packed)
        GR4PM_TRY(a_loop.run_fused(a_z.p, a_scale, a_sr_spans, n_llr, ltags.data(), n_lt, a_packed.p, 0, done, ptype));
        if (a_loop.n_pay != 0) {
            set_error("pass A: %zu payload LLRs behind a header that was declared invalid", a_loop.n_pay);
            return GR4PM_ERR_INTERNAL;
        }
        A_LAP(); // 6 header loop
        for (uint64_t o : order) a_order.push_back(o);
        for (const auto& m : done) {
            if (a_order.empty()) break;
            s.newly_known.emplace_back(a_order.front(), m);
            a_order.pop_front();
        }
        std::sort(s.newly_known.begin(), s.newly_known.end(),
                  [](const auto& a, const auto& b) { return a.first < b.first; });
    }
    // what stage 1 will want to know about this batch's own detections
    s.pre_msgs.assign(std::max<size_t>(s.n_det, 1), gr4pm_header_msg{ 0, 1 });
    for (size_t i = 0; i < s.n_det; ++i) {
        const uint64_t idx = s.base + s.det_tags[i].index;
        auto it = std::lower_bound(s.newly_known.begin(), s.newly_known.end(), idx,
                                   [](const auto& a, uint64_t v) { return a.first < v; });
        if (it != s.newly_known.end() && it->first == idx) s.pre_msgs[i] = it->second;
        else if (std::find(awaiting_idx.begin(), awaiting_idx.end(), idx) != awaiting_idx.end() ||
                 std::find(a_order.begin(), a_order.end(), idx) != a_order.end())
            s.pre_msgs[i].invalid_header = 2;
    }
    if (n >= static_cast<long long>(tail_len))
        GR4PM_HIP_TRY(hipMemcpyAsync(a_tail.p, y + (n - tail_len), tail_len * sizeof(gr4pm_c64), hipMemcpyDeviceToDevice,
                                     st1));
    A_LAP(); // 7 bookkeeping
    GR4PM_HIP_TRY(hipStreamSynchronize(st1));
    A_LAP(); // 8 final wait
#ifdef GR4PM_TIMING
    if (++lap_n % 8 == 0) {
        fprintf(stderr, "[gr4pm timing] pass A laps (us, mean of 8): gather %.0f cfc+symf %.0f wipe %.0f pmi %.0f costas %.0f "
                        "remove+llr %.0f header_loop %.0f bookkeeping %.0f final_wait %.0f\n",
                lap_sum[0] / 8, lap_sum[1] / 8, lap_sum[2] / 8, lap_sum[3] / 8, lap_sum[4] / 8, lap_sum[5] / 8,
                lap_sum[6] / 8, lap_sum[7] / 8, lap_sum[8] / 8);
        for (double& v : lap_sum) v = 0;
    }
#endif
#undef A_LAP
    return GR4PM_OK;
}

gr4pm_status gr4pm_packet_receiver::stage1_decode(Slot& s, const gr4pm_c64* y)
{
    (void)y;
    auto lookup = [&](uint64_t idx) -> const gr4pm_header_msg* {
        auto it = std::lower_bound(s.newly_known.begin(), s.newly_known.end(), idx,
                                   [](const auto& a, uint64_t v) { return a.first < v; });
        return it != s.newly_known.end() && it->first == idx ? &it->second : nullptr;
    };
    s.has_resolve = false;
    const uint64_t earlier = stage1_earlier; // batches in front of this one (stage 1 is one thread, or the caller's)
    if (pending_real)
        if (const gr4pm_header_msg* m = lookup(pending_idx)) {
            GR4PM_TRY(gr4pm_syncword_detection_filter_gate_resolve(sdf, m));
            pending_real = false;
            {   // (rare: a detection within the last ~850 items of the batch before) stage 1b is done with every earlier batch
                std::unique_lock<std::mutex> w(stage1b_mutex);
                stage1b_cv.wait(w, [&] { return stage1b_done >= earlier || stage1b_abort; });
                if (stage1b_done < earlier) {
                    set_error("packet receiver: shut down while a pending header waited for the symbol filter stage");
                    return GR4PM_ERR_INVALID;
                }
            }
            bool patched = false;
            std::lock_guard<std::mutex> lk(s1_fifo_mutex);
            for (auto& f : s1_fifo)
                if (f.invalid_header == 2) { // its tag has not even reached PayloadMetadataInsert yet
                    f = *m;
                    patched = true;
                    break;
                }
            if (!patched) {
                s.has_resolve = true;
                s.resolve = *m;
            }
        }
    // per-tag messages: decoded / still on its way (2)
    std::vector<uint64_t> idx(s.n_det);
    s.msgs = s.pre_msgs;
    for (size_t i = 0; i < s.n_det; ++i) idx[i] = s.base + s.det_tags[i].index;
    s.accepted.assign(std::max<size_t>(s.n_det, 1), 0);
    size_t used = 0;
    GR4PM_TRY(gr4pm_syncword_detection_filter_gate(sdf, idx.data(), s.n_det, s.msgs.data(), s.n_det, 1,
                                                   s.accepted.data(), &used));
    s.n_tags = 0;
    std::unique_lock<std::mutex> lk(s1_fifo_mutex);
    for (size_t i = 0; i < s.n_det; ++i)
        if (s.accepted[i]) {
            s.tags[s.n_tags++] = s.det_tags[i];
            s1_fifo.push_back(s.msgs[i]);
            if (s.msgs[i].invalid_header == 2) {
                pending_real = true;
                pending_idx = idx[i];
            }
        }
    lk.unlock();
    GR4PM_TRY(gr4pm_cfc_symbol_filter_plan(cfc, s.consumed, s.tags.data(), s.n_tags, &s.plan));
    return GR4PM_OK;
}

#ifdef GR4PM_TIMING
#include <chrono>
namespace {
struct StageTimer { // wall time of the calls of one stage, printed every 8 batches (make EXTRA=-DGR4PM_TIMING)
    const char* name;
    std::vector<std::pair<const char*, double>> acc;
    std::chrono::steady_clock::time_point t;
    int batches = 0;
    size_t at = 0;
    void begin() { t = std::chrono::steady_clock::now(); at = 0; }
    void mark(const char* what)
    {
        const auto now = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(now - t).count();
        if (at >= acc.size()) acc.emplace_back(what, 0.0);
        acc[at++].second += us;
        t = now;
    }
    void end()
    {
        if (++batches % 8) return;
        fprintf(stderr, "[gr4pm timing] %s:", name);
        for (auto& a : acc) fprintf(stderr, " %s %.0f", a.first, a.second / batches);
        fprintf(stderr, " us (mean of %d)\n", batches);
    }
};
StageTimer g_t2{ "stage 2", {}, {}, 0, 0 }, g_t3{ "stage 3", {}, {}, 0, 0 };
}
#define T2_BEGIN() g_t2.begin()
#define T2_MARK(x) g_t2.mark(x)
#define T2_END() g_t2.end()
#define T3_BEGIN() g_t3.begin()
#define T3_MARK(x) g_t3.mark(x)
#define T3_END() g_t3.end()
#else
#define T2_BEGIN()
#define T2_MARK(x)
#define T2_END()
#define T3_BEGIN()
#define T3_MARK(x)
#define T3_END()
#endif

gr4pm_status gr4pm_packet_receiver::stage2_decode(Slot& s)
{
    T2_BEGIN();
    if (s.has_resolve) GR4PM_TRY(gr4pm_payload_metadata_insert_resolve(pmi, &s.resolve));
    // Symbols PayloadMetadataInsert could not take in the batch before (below) come first: [carry | this batch's symbols]
    const gr4pm_c64* pmi_in = s.sym.p;
    size_t n = s.n_symbols;
    if (pm_carry_n) {
        if (pm_join.n < pm_carry_n + n + 1) GR4PM_TRY(pm_join.alloc(2 * (pm_carry_n + n + 1)));
        GR4PM_HIP_TRY(hipMemcpyAsync(pm_join.p, pm_carry.p, pm_carry_n * sizeof(gr4pm_c64), hipMemcpyDeviceToDevice, streams[2]));
        GR4PM_HIP_TRY(hipMemcpyAsync(pm_join.p + pm_carry_n, s.sym.p, n * sizeof(gr4pm_c64), hipMemcpyDeviceToDevice, streams[2]));
        std::vector<gr4pm_tag> tj(pm_carry_tags);
        std::vector<gr4pm_header_msg> hj(pm_carry_hdrs);
        for (size_t i = 0; i < s.n_sym_tags; ++i) {
            tj.push_back(s.sym_tags[i]);
            tj.back().index += pm_carry_n;
            hj.push_back(s.hdrs[i]);
        }
        // (copied in, never swapped: the slot's tag array keeps its capacity -- the symbol filter of a later batch is told
        // sym_tags.size() as the room it has)
        if (s.sym_tags.size() < tj.size()) s.sym_tags.resize(tj.size());
        std::copy(tj.begin(), tj.end(), s.sym_tags.begin());
        s.n_sym_tags = tj.size();
        hj.resize(std::max<size_t>(hj.size(), 1));
        s.hdrs.swap(hj);
        pmi_in = pm_join.p;
        n += pm_carry_n;
        pm_carry_n = 0;
        pm_carry_tags.clear();
        pm_carry_hdrs.clear();
    }
    if (!p.packets_only && s.data.n < n + 1) GR4PM_TRY(s.data.alloc(n + 1));
    s.packet_tags.resize(3 * s.n_sym_tags + 8);
    s.data_tags.resize(s.packet_tags.size());
    s.llr_tags.resize(s.packet_tags.size());
    size_t n_pt = 0, consumed = 0, produced = 0, used = 0, ignored = 0;
    // (round 6) PayloadMetadataInsert on the host alone: its output stream is never written -- the Costas loop below reads
    // the block's INPUT through the span table (one read and one write of the symbol stream less per batch)
    GR4PM_TRY(gr4pm::payload_metadata_insert_plan(pmi, n, n + 1, s.sym_tags.data(), s.n_sym_tags, s.hdrs.data(), s.n_sym_tags, 1,
                                                  s.packet_tags.data(), s.packet_tags.size(), &n_pt, &consumed, &produced, &used,
                                                  &ignored, s.pm_spans));
    if (consumed != n) {
        // PayloadMetadataInsert has reached the payload of a packet whose header is still pending: pass A decodes a
        // header from a 912-item window, and a detection within the last ~848 items of a batch gets its message with
        // the NEXT batch (invalid_header == 2, gr4pm_payload_metadata_insert_resolve) -- while the real chain may
        // already have produced the packet's 192 syncword + header symbols when the batch ends 816 .. 848 items behind
        // the tag.  The reference block waits there (payload_metadata_insert.hpp:243-247); so does this stage: the few
        // symbols it could not take (and their tags) are carried to the front of the next batch.  (Round 3 failed the
        // batch here: one cut in ~200 random ones, found by tools/stress_receiver.py ... decode.)
        const size_t rest = n - consumed;
        if (rest > kPmCarryMax) {
            set_error("PayloadMetadataInsert stalled at symbol %zu of %zu: a header message is missing", consumed, n);
            return GR4PM_ERR_INVALID;
        }
        if (pm_carry.n < kPmCarryMax) GR4PM_TRY(pm_carry.alloc(kPmCarryMax));
        GR4PM_HIP_TRY(hipMemcpyAsync(pm_carry.p, pmi_in + consumed, rest * sizeof(gr4pm_c64), hipMemcpyDeviceToDevice, streams[2]));
        pm_carry_n = rest;
        for (size_t i = 0; i < s.n_sym_tags; ++i)
            if (s.sym_tags[i].index >= consumed) {
                pm_carry_tags.push_back(s.sym_tags[i]);
                pm_carry_tags.back().index -= consumed;
                pm_carry_hdrs.push_back(s.hdrs[i]);
            }
    }
    s.n_packet_tags = n_pt;
    s.ignored = ignored;
    T2_MARK("pmi");
    if (s.out_cap < produced) {
        set_error("out_cap %zu < %zu symbols", s.out_cap, produced);
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    {   // The PLL's kernel form.  Round 6 measured the 32-VGPR form (k_costas_chains_cap, which starts beside a correlator
        // workgroup) here too: 6.3 - 6.5 ms per 2^28 samples of packets back to back against 5.9 - 6.0 for the 121- and the
        // 71-register forms, same box (profiles/r6_dense_ab.txt) -- behind PayloadMetadataInsert this stage is the
        // pipeline's longest, and the slower kernel (2.6 instead of 1.2 ms) lengthens it.  GR4PM_COSTAS_SMALL_DECODE: A/B.
        static const char* form = experiment_env("GR4PM_COSTAS_SMALL_DECODE", false);
        (void)gr4pm_costas_loop_set_small_footprint(costas, form ? atoi(form) : 0);
    }
    GR4PM_TRY(gr4pm::costas_loop_process_packets_from(costas, pmi_in, s.pm_spans.data(), s.pm_spans.size(), produced,
                                                      s.out_symbols, s.packet_tags.data(), n_pt));
    T2_MARK("costas");
    s.n_symbols = produced;
    size_t n_dt = 0, n_data = 0, n_lt = 0, n_llr = 0;
    if (p.packets_only) {
        // SyncwordRemove and the LLR decoder on the host alone (state, tags, span table): stage 3's one kernel reads the
        // Costas loop's output through the composed table (hostlogic/tail_plan.hpp); neither `data` nor the LLR stream exists
        GR4PM_TRY(gr4pm::syncword_remove_plan(remove, produced, s.packet_tags.data(), n_pt, s.data_tags.data(), s.data_tags.size(),
                                              &n_dt, &n_data, s.sr_spans));
        bool all_qpsk = true;
        GR4PM_TRY(gr4pm::llr_decoder_plan(llr, n_data, s.data_tags.data(), n_dt, s.llr_tags.data(), s.llr_tags.size(), &n_lt,
                                          &n_llr, &all_qpsk, &s.llr_scale));
        if (!all_qpsk) { // (PayloadMetadataInsert names QPSK for the header and nothing for the payload: packet_receiver.hpp:127-130)
            set_error("packets_only: a run behind SyncwordRemove is not QPSK");
            return GR4PM_ERR_INVALID;
        }
        T2_MARK("remove+llr plans");
        s.n_llr_tags = n_lt;
        s.n_llr = n_llr;
        s.n_data = 0;
        s.pdus.clear();
        s.pdu_resyncs = 0;
    } else {
    GR4PM_TRY(gr4pm_syncword_remove_process(remove, s.out_symbols, produced, s.data.p, s.packet_tags.data(), n_pt,
                                            s.data_tags.data(), s.data_tags.size(), &n_dt, &n_data));
    T2_MARK("remove");
    GR4PM_TRY(gr4pm_constellation_llr_decoder_process(llr, s.data.p, n_data, s.out_llr, s.llr_cap, s.data_tags.data(),
                                                      n_dt, s.llr_tags.data(), s.llr_tags.size(), &n_lt, &n_llr));
    T2_MARK("llr");
    s.n_llr_tags = n_lt;
    s.n_llr = n_llr;
    GR4PM_TRY(split_symbol_pdus(s, n_data, n_dt));
    }
    s.opened.clear();
    for (size_t i = 0, j = 0; i < n_pt; ++i)
        if (s.packet_tags[i].kind == GR4PM_PKT_SYNCWORD) { // a packet PayloadMetadataInsert opened: both lists ascend
            while (j < s.n_sym_tags && s.sym_tags[j].index < s.packet_tags[i].syncword.index) ++j;
            if (j < s.n_sym_tags && s.sym_tags[j].index == s.packet_tags[i].syncword.index) s.opened.push_back(s.hdrs[j]);
        }
    T2_END();
    return GR4PM_OK;
}

// stage 3 (decode_headers): the chain's own header loop (verification of pass A, descrambled
// payload) and the payload tail, on their own stream and thread
gr4pm_status gr4pm_packet_receiver::stage3_decode(Slot& s)
{
    hipStream_t st2 = streams[3];
    T3_BEGIN();
    const size_t n_llr = s.n_llr, n_lt = s.n_llr_tags;
    auto same_header = [](const gr4pm_header_msg& given, const gr4pm_header_msg& got) {
        return given.invalid_header == got.invalid_header && (got.invalid_header == 1 || given.packet_length == got.packet_length);
    };
    s.header_mismatches = 0;
    if (s.has_resolve) {
        bool placed = false;
        for (auto& m : used_msgs)
            if (m.invalid_header == 2) {
                m = s.resolve;
                placed = true;
                break;
            }
        // the chain's own decode of that header came first (below): the message is checked against it now
        if (!placed && !early_hdrs.empty()) {
            if (!same_header(s.resolve, early_hdrs.front())) ++s.header_mismatches;
            early_hdrs.pop_front();
        }
    }
    for (const auto& m : s.opened) used_msgs.push_back(m);
    s.header_messages.clear();
    s.packet_type.clear();
    const size_t lean_bit0 = carry_bits; // packets_only: this batch's payload bits go behind the carried ones
    if (p.packets_only) {
        uint8_t* stream_bytes = nullptr;
        {   // the packer's byte stream of this batch: carried bits + at most one bit per LLR
            const size_t want = (carry_bits + n_llr) / 8 + 16;
            auto& pk = packed2[packed_cur];
            if (pk.n < want) {
                DevBuf<uint8_t> bigger;
                GR4PM_TRY(bigger.alloc(want * 2));
                if (carry_bits)
                    GR4PM_HIP_TRY(hipMemcpyAsync(bigger.p, pk.p, (carry_bits + 7) / 8, hipMemcpyDeviceToDevice, st2));
                GR4PM_HIP_TRY(hipStreamSynchronize(st2));
                std::swap(pk.p, bigger.p);
                std::swap(pk.n, bigger.n);
            }
            stream_bytes = pk.p;
        }
        GR4PM_TRY(b_loop.run_fused(s.out_symbols, s.llr_scale, s.sr_spans, n_llr, s.llr_tags.data(), n_lt, stream_bytes,
                                   lean_bit0, s.header_messages, s.packet_type));
    } else {
    // (the descrambled payload LLRs of this batch go straight to the slot's buffer, which the caller reads: round 5 --
    // they used to be copied there, and once more behind the unfinished packet's, 1.9 GB of traffic a step on a
    // packet-dense stream in the stage that sets the pace of the decode_headers pipeline)
    if (s.payload_llr.n < n_llr + 1) GR4PM_TRY(s.payload_llr.alloc(n_llr + 1));
    GR4PM_TRY(b_loop.run(s.out_llr, n_llr, s.llr_tags.data(), n_lt, s.header_messages, s.packet_type, s.payload_llr.p));
    }
    T3_MARK("header_loop");
    for (const auto& got : s.header_messages) {
        if (used_msgs.empty()) break;
        const gr4pm_header_msg given = used_msgs.front();
        used_msgs.pop_front();
        if (given.invalid_header == 2) {
            // pass A's message for this packet is still pending (it arrives with the next batch: a batch that ends 816 ..
            // 848 items behind the syncword, stage2_decode) while the chain has already decoded the header itself
            early_hdrs.push_back(got);
            continue;
        }
        if (!same_header(given, got)) ++s.header_mismatches;
    }
    const size_t n_pay = b_loop.n_pay;
    const float* pay = p.packets_only ? nullptr : s.payload_llr.p;
    s.n_payload_llr = n_pay;
    s.n_payload_tags = b_loop.n_pay_tags;
    s.payload_tags.assign(b_loop.pay_tags.begin(), b_loop.pay_tags.begin() + b_loop.n_pay_tags);
    // payload tail, packet_receiver.hpp:140-147 (whole packets only; the rest waits): the stream is
    // [soft[0 .. soft_n): the unfinished packet of the batches before | pay[0 .. n_pay)], sliced and packed in place
    for (size_t i = 0; i < b_loop.n_pay_tags; ++i) payload_bits.push_back(b_loop.pay_tags[i].payload_bits);
    const size_t total = (p.packets_only ? carry_bits : soft_n) + n_pay;
    std::vector<uint64_t> lens, offs;
    size_t used_bits = 0;
    while (!payload_bits.empty() && used_bits + payload_bits.front() <= total) {
        offs.push_back(used_bits / 8);
        lens.push_back(payload_bits.front() / 8);
        used_bits += payload_bits.front();
        payload_bits.pop_front();
    }
    s.packet_lengths.assign(lens.size(), 0);
    s.n_packet_bytes = 0;
    const uint8_t* packed_bytes = p.packets_only ? packed2[packed_cur].p : nullptr;
    if (!lens.empty()) {
        if (!p.packets_only) {
            if (packed.n < used_bits / 8 + 1) GR4PM_TRY(packed.alloc(used_bits / 8 * 2 + 1));
            GR4PM_TRY(gr4pm::slice_pack_two(soft.p, soft_n, pay, used_bits / 8, packed.p, st2));
            packed_bytes = packed.p;
        }
        if (s.packets_cap < used_bits / 8) {
            set_error("packets_cap %zu < %zu bytes", s.packets_cap, used_bits / 8);
            return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
        }
        GR4PM_TRY(gr4pm_crc_check_process(crc, packed_bytes, offs.data(), lens.data(), lens.size(), s.out_packets,
                                          s.packet_lengths.data(), &s.n_packet_bytes));
    }
    // what is left for the next batch
    const size_t rest = total - used_bits;
    if (p.packets_only) {
        // the unfinished packet's bytes (the last one perhaps half full) move to the front of the other buffer: the next
        // batch's kernel goes on writing behind bit `rest` there.  (used_bits is a whole number of bytes: packets are.)
        auto& next = packed2[packed_cur ^ 1];
        const size_t n_bytes = (rest + 7) / 8;
        if (next.n < n_bytes + 16) GR4PM_TRY(next.alloc((n_bytes + 16) * 2));
        if (n_bytes)
            GR4PM_HIP_TRY(hipMemcpyAsync(next.p, packed2[packed_cur].p + used_bits / 8, n_bytes, hipMemcpyDeviceToDevice, st2));
        packed_cur ^= 1;
        carry_bits = rest;
        GR4PM_HIP_TRY(hipStreamSynchronize(st2));
        T3_MARK("payload_tail");
        T3_END();
        return GR4PM_OK;
    }
    auto soft_room = [&](size_t want, size_t keep) -> gr4pm_status { // (keep: items at the front that must survive)
        if (soft.n >= want + 8) return GR4PM_OK;
        DevBuf<float> bigger;
        GR4PM_TRY(bigger.alloc((want + 8) * 2));
        if (keep) GR4PM_HIP_TRY(hipMemcpyAsync(bigger.p, soft.p, keep * sizeof(float), hipMemcpyDeviceToDevice, st2));
        GR4PM_HIP_TRY(hipStreamSynchronize(st2));
        std::swap(soft.p, bigger.p);
        std::swap(soft.n, bigger.n);
        return GR4PM_OK;
    };
    // (the carried LLRs belong to ONE unfinished packet -- every complete one was consumed by the batch that completed
    // it -- so a packet that ends in this batch ends behind them: used_bits is 0 or > soft_n)
    if (used_bits && used_bits <= soft_n) { // cannot happen while the state above is consistent: never index pay[] below 0
        set_error("payload tail: %zu bits end inside the %zu carried ones", used_bits, soft_n);
        return GR4PM_ERR_INTERNAL;
    }
    if (used_bits) { // the tail of this batch's LLRs: one short copy, in stream order behind the kernel that read soft
        GR4PM_TRY(soft_room(rest, 0));
        if (rest)
            GR4PM_HIP_TRY(hipMemcpyAsync(soft.p, pay + (used_bits - soft_n), rest * sizeof(float), hipMemcpyDeviceToDevice,
                                         st2));
    } else { // no packet ended in this batch (a batch with little payload): its LLRs go behind the carried ones
        GR4PM_TRY(soft_room(rest, soft_n));
        if (n_pay)
            GR4PM_HIP_TRY(hipMemcpyAsync(soft.p + soft_n, pay, n_pay * sizeof(float), hipMemcpyDeviceToDevice, st2));
    }
    soft_n = rest;
    GR4PM_HIP_TRY(hipStreamSynchronize(st2));
    T3_MARK("payload_tail");
    T3_END();
    return GR4PM_OK;
}

extern "C" {

gr4pm_status gr4pm_packet_receiver_create(const gr4pm_packet_receiver_params* p, gr4pm_packet_receiver** out)
try {
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
    // the detector's small, latency-bound kernels set the pace of the pipeline: they get the highest
    // stream priority (its big kernel, the correlator, runs on the detector's own low-priority
    // look-ahead stream)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return bail(GR4PM_ERR_HIP);
    for (int i = 0; i < 6; ++i) {
        // pass A of the header loop (streams[5]) is a chain of fifteen small dependent kernels: on a chip that the
        // correlator and the filters keep full, each of them waits for a free slot -- at the highest priority it is
        // dispatched first
        // (GR4PM_HIGH_STREAMS: the digits of the stage streams that get the highest priority, default "05")
        static const char* high = getenv("GR4PM_HIGH_STREAMS") ? getenv("GR4PM_HIGH_STREAMS") : "05";
        const int prio = strchr(high, '0' + i) != nullptr ? greatest : 0;
        if (hipStreamCreateWithPriority(&h->streams[i], hipStreamNonBlocking, prio) != hipSuccess)
            return bail(GR4PM_ERR_HIP);
    }
    const size_t sps = p->samples_per_symbol;
    // packet_receiver.hpp:60-74: RRC taps normalised to unit RMS norm (float accumulation)
    std::vector<float> rrc(((sps * 11) | 1));
    const size_t n_rrc = gr4pm_firdes_root_raised_cosine(1.0, static_cast<double>(sps), 1.0, 0.35, sps * 11, rrc.data());
    if (n_rrc == 0) return bail(GR4PM_ERR_NOMEM); // the design ran out of host memory (its entry point reports that as 0 taps)
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
    gr4pm::sd_set_coresident(h->sd, true);
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
    if (n_pfb == 0) return bail(GR4PM_ERR_NOMEM);
    pfb.resize(n_pfb - 1); // the design is odd-length: drop the last tap (:108-110)
    gr4pm_symbol_filter_params fsp{ sps, pfb.data(), pfb.size(), arms, rrc.size() - 1, 0, h->streams[4] };
    if ((st = gr4pm_symbol_filter_create(&fsp, &h->symf)) != GR4PM_OK) return bail(st);
    float bipolar[64];
    for (int i = 0; i < 64; ++i) bipolar[i] = syncword[i] ? -1.0f : 1.0f; // :117-122
    gr4pm_syncword_wipeoff_params wp{ bipolar, 64, h->streams[4] };
    if ((st = gr4pm_syncword_wipeoff_create(&wp, &h->wipe)) != GR4PM_OK) return bail(st);
    gr4pm_costas_loop_params cp{ 0.01, p->soft_bits ? 1 : p->costas_constellation, 1, h->streams[2] }; // :125
    if ((st = gr4pm_costas_loop_create(&cp, &h->costas)) != GR4PM_OK) return bail(st);
    // Pipelined, the PLL runs while the correlator of a later batch has the chip: the 32-VGPR form of k_costas fits
    // beside two correlator waves (2 x 240 of a SIMD's 512 registers), so its waves -- one per 64 packets, alive for as
    // long as a packet's chain takes -- no longer keep a correlator workgroup off their CU.  Round 4: +2.5 % for the
    // receiver although the kernel alone is slower (GR4PM_COSTAS_SMALL = 0 / 1 / 2 for A/B: 112 / 62 / 32 VGPRs).
    if (p->pipelined) {
        static const char* small = experiment_env("GR4PM_COSTAS_SMALL", false);
        h->costas_form = small ? atoi(small) : 2;
        (void)gr4pm_costas_loop_set_small_footprint(h->costas, h->costas_form);
    }
    if (p->soft_bits) {
        gr4pm_payload_metadata_insert_params pp{ 64, 128, 0.02, 0.01, 0.005, h->streams[2] }; // :123-124
        if ((st = gr4pm_payload_metadata_insert_create(&pp, &h->pmi)) != GR4PM_OK) return bail(st);
        hipStream_t tail = p->decode_headers ? h->streams[2] : h->streams[3]; // stage 3 of the soft_bits mode
        gr4pm_syncword_remove_params sr{ 64, tail }; // :126
        if ((st = gr4pm_syncword_remove_create(&sr, &h->remove)) != GR4PM_OK) return bail(st);
        gr4pm_constellation_llr_decoder_params lp{ 0.7f, 2, tail }; // :127-130
        if ((st = gr4pm_constellation_llr_decoder_create(&lp, &h->llr)) != GR4PM_OK) return bail(st);
    }
    if (p->decode_headers) {
        if (!p->soft_bits || !p->header_alist) {
            set_error("decode_headers needs soft_bits and the header code's alist");
            return bail(GR4PM_ERR_INVALID);
        }
    }
    if (p->packets_only && !p->decode_headers) {
        set_error("packets_only is a form of the decode_headers receiver");
        return bail(GR4PM_ERR_INVALID);
    }
    if (p->decode_headers) {
        // pass A: the same blocks a second time, on a stream (and pipeline stage) of their own
        hipStream_t s1 = h->streams[5], s2 = h->streams[3];
        gr4pm_rotator_params rp2{ 1, 0.0f, (rrc.size() - 1) / 2 + sps, 1, s1 };
        if ((st = gr4pm_rotator_create(&rp2, &h->a_cfc)) != GR4PM_OK) return bail(st);
        gr4pm_symbol_filter_params fsp2{ sps, pfb.data(), pfb.size(), arms, rrc.size() - 1, 0, s1 };
        if ((st = gr4pm_symbol_filter_create(&fsp2, &h->a_symf)) != GR4PM_OK) return bail(st);
        gr4pm_syncword_wipeoff_params wp2{ bipolar, 64, s1 };
        if ((st = gr4pm_syncword_wipeoff_create(&wp2, &h->a_wipe)) != GR4PM_OK) return bail(st);
        gr4pm_payload_metadata_insert_params pp2{ 64, 128, 0.02, 0.01, 0.005, s1 };
        if ((st = gr4pm_payload_metadata_insert_create(&pp2, &h->a_pmi)) != GR4PM_OK) return bail(st);
        gr4pm_costas_loop_params cp2{ 0.01, 1, 1, s1 };
        if ((st = gr4pm_costas_loop_create(&cp2, &h->a_costas)) != GR4PM_OK) return bail(st);
        gr4pm_syncword_remove_params sr2{ 64, s1 };
        if ((st = gr4pm_syncword_remove_create(&sr2, &h->a_remove)) != GR4PM_OK) return bail(st);
        gr4pm_constellation_llr_decoder_params lp2{ 0.7f, 2, s1 };
        if ((st = gr4pm_constellation_llr_decoder_create(&lp2, &h->a_llr)) != GR4PM_OK) return bail(st);
        if ((st = h->a_loop.create(p->header_alist, s1)) != GR4PM_OK) return bail(st);
        if ((st = h->b_loop.create(p->header_alist, s2)) != GR4PM_OK) return bail(st); // :131-139
        gr4pm_crc_check_params cc{ 32, 0x4C11DB7, 0xFFFFFFFF, 0xFFFFFFFF, 1, 1, 0, 1, 0, s2 }; // :145-147
        if ((st = gr4pm_crc_check_create(&cc, &h->crc)) != GR4PM_OK) return bail(st);
        const size_t tail_len = gr4pm_packet_receiver::kW + gr4pm_packet_receiver::kPre;
        if ((st = h->a_tail.alloc(tail_len)) != GR4PM_OK) return bail(st);
        if ((st = h->a_tail.zero(s1)) != GR4PM_OK) return bail(st);
        if ((st = h->a_head.alloc(tail_len + gr4pm_packet_receiver::kW)) != GR4PM_OK) return bail(st);
        if (hipStreamSynchronize(s1) != hipSuccess) return bail(GR4PM_ERR_HIP);
    }
    const size_t tags_cap = std::max<size_t>(p->tags_cap, 64);
    // every slot's buffers at the size of a full batch, now: a slot is first used by one of the first
    // kSlots batches, and a device allocation in the middle of a stream stalls all stages
    const size_t sym_cap = p->max_items / sps + tags_cap + 2;
    for (int i = 0; i < kSlots; ++i) {
        auto& sl = h->slots[i];
        sl.det_tags.resize(tags_cap);
        sl.tags.resize(tags_cap);
        sl.sym_tags.resize(tags_cap + 64);
        if ((st = sl.sym.alloc(sym_cap)) != GR4PM_OK) return bail(st);
        if (p->soft_bits) {
            if (!p->decode_headers && (st = sl.pm.alloc(sym_cap + 1)) != GR4PM_OK) return bail(st); // (decode: the PLL gathers)
            if (!p->packets_only && (st = sl.data.alloc(sym_cap + 1)) != GR4PM_OK) return bail(st);
            if (!p->decode_headers && (st = sl.z.alloc(sym_cap + 1)) != GR4PM_OK) return bail(st);
        }
        h->free_slots.push(i);
    }
    // HIP's current device is a per-thread setting that starts at device 0: the stage threads work on the device the
    // handle was created on (one process per GPU under torch.distributed: rank k creates it on device k)
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return bail(GR4PM_ERR_HIP);
    if (p->pipelined) {
        // one thread per stage (hostlogic::run_stage): a C++ exception inside a stage body (std::bad_alloc from a table
        // that grows, std::system_error) fails that batch -- collect() returns its status -- and the thread lives on
        struct StageDef {
            int worker, clk;
            SlotRing *from, *to;
            void (gr4pm_packet_receiver::*body)(Slot&);
            const char* name;
        };
        const StageDef stages[5] = {
            { 4, 5, &h->to_stageA, &h->to_stage1, &gr4pm_packet_receiver::stageA, "packet receiver, stage A" },
            { 0, 1, &h->to_stage1, &h->to_stage1b, &gr4pm_packet_receiver::stage1, "packet receiver, stage 1" },
            { 3, 4, &h->to_stage1b, &h->to_stage2, &gr4pm_packet_receiver::stage1b, "packet receiver, stage 1b" },
            { 1, 2, &h->to_stage2, &h->to_stage3, &gr4pm_packet_receiver::stage2, "packet receiver, stage 2" },
            { 2, 3, &h->to_stage3, &h->done, &gr4pm_packet_receiver::stage3, "packet receiver, stage 3" },
        };
        try {
            for (const StageDef& sd : stages)
                h->workers[sd.worker] = std::thread([h, device, sd] {
                    (void)hipSetDevice(device);
                    gr4pm_set_deferred_sync(getenv("GR4PM_NO_DEFER") ? 0 : 1);
                    hostlogic::run_stage(
                        *sd.from, sd.to, /*forward_quit=*/sd.to != &h->done,
                        [&](int i) {
                            CLK_GOT(h->clk[sd.clk]);
                            (h->*sd.body)(h->slots[i]);
                            CLK_DONE(h->clk[sd.clk]);
                        },
                        [&](int i) { h->fail(h->slots[i], exception_status(sd.name)); });
                });
        } catch (...) { // std::thread could not start: wind down the ones that did
            const gr4pm_status ts = exception_status("gr4pm_packet_receiver_create (stage threads)");
            gr4pm_packet_receiver_destroy(h);
            return ts;
        }
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

void gr4pm_packet_receiver_destroy(gr4pm_packet_receiver* h)
try {
    if (!h) return;
    // Only the head of the chain is told to stop: run_stage hands the end on behind the batches still queued, so every
    // stage sees its predecessors' deliveries first (an idle stage 1b that stopped at once would leave stage 1 waiting
    // for a batch stage 1b never counts).  A stage whose predecessor's thread never started (create failed half way)
    // would not see a forwarded end: its queue is stopped here.
    {
        SlotRing* const queues[5] = { &h->to_stageA, &h->to_stage1, &h->to_stage1b, &h->to_stage2, &h->to_stage3 };
        const int order[5] = { 4, 0, 3, 1, 2 }; // worker index of the stage reading queues[k]
        for (int k = 0; k < 5; ++k)
            if (k == 0 || !h->workers[order[k - 1]].joinable()) queues[k]->stop();
        if (!h->workers[3].joinable()) { // stage 1b never started: nobody may wait for it
            {
                std::lock_guard<std::mutex> lk(h->stage1b_mutex);
                h->stage1b_abort = true;
            }
            h->stage1b_cv.notify_all();
        }
        for (int k = 0; k < 5; ++k) {
            if (!h->workers[order[k]].joinable()) continue;
            h->workers[order[k]].join();
            if (order[k] == 3) { // stage 1b is gone: release whoever might still wait for it (belt and braces)
                {
                    std::lock_guard<std::mutex> lk(h->stage1b_mutex);
                    h->stage1b_abort = true;
                }
                h->stage1b_cv.notify_all();
            }
        }
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
    gr4pm_rotator_destroy(h->a_cfc);
    gr4pm_symbol_filter_destroy(h->a_symf);
    gr4pm_syncword_wipeoff_destroy(h->a_wipe);
    gr4pm_payload_metadata_insert_destroy(h->a_pmi);
    gr4pm_costas_loop_destroy(h->a_costas);
    gr4pm_syncword_remove_destroy(h->a_remove);
    gr4pm_constellation_llr_decoder_destroy(h->a_llr);
    h->a_loop.destroy();
    h->b_loop.destroy();
    gr4pm_crc_check_destroy(h->crc);
    for (auto pub : h->pdu_pub) gr4pm_zmq_pub_destroy(pub);
    for (auto s : h->streams)
        if (s) (void)hipStreamDestroy(s);
    delete h;
}
GR4PM_ABI_CATCH_VOID

size_t gr4pm_packet_receiver_inflight(const gr4pm_packet_receiver* h) { return h ? h->inflight : 0; }

gr4pm_status gr4pm_packet_receiver_submit(gr4pm_packet_receiver* h, const gr4pm_c64* in, size_t n_in,
                                          const gr4pm_c64* delayed, const gr4pm_c64* next_in, size_t next_n,
                                          uint64_t packet_length, gr4pm_c64* out_symbols, size_t out_cap,
                                          float* out_llr, size_t llr_cap, uint8_t* out_packets, size_t packets_cap)
try {
    if (!h || !in || !out_symbols || (h->p.soft_bits && !h->p.packets_only && !out_llr) || (h->p.decode_headers && !out_packets))
        return GR4PM_ERR_INVALID;
    if (h->p.decode_headers && n_in < gr4pm_packet_receiver::kW + gr4pm_packet_receiver::kPre + 2048) {
        set_error("decode_headers needs batches of at least %u items",
                  gr4pm_packet_receiver::kW + gr4pm_packet_receiver::kPre + 2048);
        return GR4PM_ERR_INVALID;
    }
    if (h->inflight >= static_cast<size_t>(kSlots - 1)) {
        set_error("%zu batches in flight: collect one first", h->inflight);
        return GR4PM_ERR_INVALID;
    }
    const int i = h->free_slots.pop();
    // until the batch counts as in flight, an exception (stage 0 runs in this thread) hands the slot back
    struct SlotReturn {
        SlotRing& q;
        int slot;
        bool armed = true;
        ~SlotReturn()
        {
            if (armed) (void)q.push(slot);
        }
    } slot_return{ h->free_slots, i };
    Slot& s = h->slots[i];
    s.in = in;
    s.n_in = n_in;
    s.delayed = delayed;
    s.out_symbols = out_symbols;
    s.out_cap = out_cap;
    s.out_llr = out_llr;
    s.llr_cap = llr_cap;
    s.packet_length = packet_length;
    s.out_packets = out_packets;
    s.packets_cap = packets_cap;
    s.header_mismatches = s.n_packet_bytes = s.n_payload_llr = s.n_payload_tags = 0;
    s.header_messages.clear();
    s.packet_type.clear();
    s.packet_lengths.clear();
    s.status = GR4PM_OK;
    s.error[0] = 0;
    s.consumed = s.n_symbols = s.n_llr = s.n_det = s.n_tags = s.n_sym_tags = s.n_packet_tags = s.n_llr_tags = 0;
    s.n_data = 0;
    s.pdus.clear();
    CLK_GOT(h->clk[0]);
    h->stage0(s, next_in, next_n);
    CLK_DONE(h->clk[0]);
    if (h->n_ahead.load(std::memory_order_relaxed) > 0) h->n_ahead.fetch_sub(1, std::memory_order_relaxed);
    if (next_in) h->n_ahead.store(1, std::memory_order_relaxed); // hint_next: the one batch named with this submit
    s.seq = h->n_submitted.fetch_add(1, std::memory_order_relaxed) + 1;
    ++h->inflight;
    slot_return.armed = false;
    if (h->p.pipelined) {
        if (h->p.decode_headers) (void)h->to_stageA.push(i);
        else (void)h->to_stage1.push(i);
    } else {
        const gr4pm_status gs = guarded("packet receiver, stages in the caller's thread", [&] {
            DeferredSyncScope defer;
            h->stageA(s);
            h->stage1(s);
            h->stage1b(s);
            h->stage2(s);
            h->stage3(s);
        });
        if (gs != GR4PM_OK) h->fail(s, gs);
        {   // an exception between stage 1 and stage 1b must not leave the next pending-header batch waiting for it
            std::lock_guard<std::mutex> lk(h->stage1b_mutex);
            if (h->stage1b_done < h->stage1_started) h->stage1b_done = h->stage1_started;
        }
        (void)h->done.push(i);
    }
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_packet_receiver_announce(gr4pm_packet_receiver* h, const gr4pm_c64* in, size_t n_in)
try {
    if (!h || !in) return GR4PM_ERR_INVALID;
    h->n_ahead.fetch_add(1, std::memory_order_relaxed);
    return gr4pm_syncword_detection_announce(h->sd, in, n_in, n_in);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_packet_receiver_collect(gr4pm_packet_receiver* h, gr4pm_packet_receiver_result* r)
try {
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
    r->llr = h->p.packets_only ? nullptr : s.out_llr; // packets_only: the LLR stream is never materialised (n_llr: its length)
    r->header_messages = s.header_messages.data();
    r->packet_type = s.packet_type.data();
    r->n_header_messages = s.header_messages.size();
    r->header_mismatches = s.header_mismatches;
    r->payload_llr = h->p.packets_only ? nullptr : s.payload_llr.p;
    r->n_payload_llr = s.n_payload_llr;
    r->payload_tags = s.payload_tags.data();
    r->n_payload_tags = s.n_payload_tags;
    r->packets = s.out_packets;
    r->n_packet_bytes = s.n_packet_bytes;
    r->packet_lengths = s.packet_lengths.data();
    r->n_packets = s.packet_lengths.size();
    r->pdu_symbols = h->p.packets_only ? nullptr : s.data.p;
    r->n_pdu_symbols = s.n_data;
    r->symbol_pdus = s.pdus.data();
    r->n_symbol_pdus = s.pdus.size();
    r->symbol_pdu_resyncs = s.pdu_resyncs;
    if (s.status != GR4PM_OK) set_error("%s", s.error);
    if (s.status != GR4PM_OK || !h->pdu_fn) {
        // a failed batch, or nobody listening: whatever the sink had collected so far is not continued (a callback
        // registered in the middle of a PDU must not get its tail as if it were a whole one)
        for (int k = 0; k < 2; ++k) {
            h->pdu_acc[k].clear();
            h->pdu_open[k] = false;
        }
    }
    if (s.status == GR4PM_OK && h->pdu_fn && s.n_data) {
        // the tap's sink side: one copy of the batch's symbols to the host, then one call per complete PDU
        h->pdu_host.resize(s.n_data);
        GR4PM_HIP_TRY(hipMemcpy(h->pdu_host.data(), s.data.p, s.n_data * sizeof(gr4pm_c64), hipMemcpyDeviceToHost));
        for (const auto& pc : s.pdus) {
            auto& acc = h->pdu_acc[pc.kind];
            if (pc.first) {
                acc.clear();
                h->pdu_open[pc.kind] = true;
            }
            if (!h->pdu_open[pc.kind]) continue; // the start of this PDU was never seen: not delivered
            acc.insert(acc.end(), h->pdu_host.begin() + static_cast<ptrdiff_t>(pc.offset),
                       h->pdu_host.begin() + static_cast<ptrdiff_t>(pc.offset + pc.length));
            if (pc.last) {
                h->pdu_fn(h->pdu_user, pc.kind, acc.data(), acc.size());
                acc.clear();
                h->pdu_open[pc.kind] = false;
            }
        }
    }
    return s.status;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_packet_receiver_set_symbol_pdu_callback(gr4pm_packet_receiver* h, gr4pm_symbol_pdu_fn fn, void* user)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (fn && !h->p.soft_bits) {
        set_error("the symbol PDU tap hangs off SyncwordRemove: soft_bits receivers only");
        return GR4PM_ERR_INVALID;
    }
    if (fn && h->p.packets_only) {
        set_error("the symbol PDU tap needs SyncwordRemove's output stream: not a packets_only receiver");
        return GR4PM_ERR_INVALID;
    }
    h->pdu_fn = fn;
    h->pdu_user = user;
    for (auto& pub : h->pdu_pub) { // the library's own sink (below) gives way to the caller's
        gr4pm_zmq_pub_destroy(pub);
        pub = nullptr;
    }
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

// zmq_pdu_pub_sink.hpp:31-41 behind the tap: kind 0 -> the header endpoint, 1 -> the payload endpoint
static void publish_symbol_pdu(void* user, int kind, const gr4pm_c64* symbols, size_t n)
{
    auto* h = static_cast<gr4pm_packet_receiver*>(user);
    if (kind >= 0 && kind < 2 && h->pdu_pub[kind]) (void)gr4pm_zmq_pub_send(h->pdu_pub[kind], symbols, n * sizeof(gr4pm_c64));
}

gr4pm_status gr4pm_packet_receiver_publish_symbol_pdus(gr4pm_packet_receiver* h, const char* header_endpoint,
                                                       const char* payload_endpoint, int ports[2])
try {
    if (!h || (header_endpoint == nullptr) != (payload_endpoint == nullptr)) return GR4PM_ERR_INVALID;
    GR4PM_TRY(gr4pm_packet_receiver_set_symbol_pdu_callback(h, nullptr, nullptr)); // (closes endpoints bound before)
    if (!header_endpoint) return GR4PM_OK;
    if (!h->p.soft_bits) {
        set_error("the symbol PDU tap hangs off SyncwordRemove: soft_bits receivers only");
        return GR4PM_ERR_INVALID;
    }
    if (h->p.packets_only) {
        set_error("the symbol PDU tap needs SyncwordRemove's output stream: not a packets_only receiver");
        return GR4PM_ERR_INVALID;
    }
    gr4pm_zmq_pub* pubs[2] = { nullptr, nullptr };
    gr4pm_status st = gr4pm_zmq_pub_create(header_endpoint, &pubs[0]);
    if (st == GR4PM_OK) st = gr4pm_zmq_pub_create(payload_endpoint, &pubs[1]);
    if (st != GR4PM_OK) {
        gr4pm_zmq_pub_destroy(pubs[0]);
        gr4pm_zmq_pub_destroy(pubs[1]);
        return st;
    }
    h->pdu_pub[0] = pubs[0], h->pdu_pub[1] = pubs[1];
    h->pdu_fn = publish_symbol_pdu;
    h->pdu_user = h;
    if (ports) ports[0] = gr4pm_zmq_pub_port(pubs[0]), ports[1] = gr4pm_zmq_pub_port(pubs[1]);
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"
