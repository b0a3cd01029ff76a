// multichannel_receiver.hip -- gr4pm_multichannel_receiver: BASELINE configs[2], n_channels
// independent receive chains on one GPU (packet_receiver.hpp:191-265 couples nothing across
// receivers).  What is serial per packet and costs the same whatever the batch holds runs ONCE
// for all channels: one batched SyncwordDetection handle (blockIdx.y = channel), one
// CoarseFrequencyCorrection handle with n_channels channels (the phasor checkpoints of every
// channel in one launch, gr4pm_cfc_symbol_filter_plan_channels) and one CostasLoop handle with
// n_channels channels (gr4pm_costas_loop_process_ragged).  In between, every channel has its
// own SyncwordDetectionFilter, SymbolFilter (fused with its share of the CFC plan) and
// SyncwordWipeoff; these are spread over worker threads with a HIP stream each, whose calls
// only queue kernels (gr4pm_set_deferred_sync).  Front-end mode of gr4pm_packet_receiver,
// channel by channel: same constants, same results (tests compare with one receiver per
// channel, bit for bit).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "hostlogic/slot_queue.hpp"

using namespace gr4pm;

// Batches in flight: the detector of batch n + 2 (caller's thread), the tag gates + CFC plan of batch n + 1
// (stage 1), the symbol filters of batch n + 1 / n (stage 2 and its workers) and the Costas loop of batch n
// (stage 3) run at the same time; the two serial kernels (phasor checkpoints, PLL), whose time is that of the
// longest segment of the batch, overlap the correlator of the next batches.
constexpr int kMcSlots = 4;

struct gr4pm_multichannel_receiver {
    gr4pm_multichannel_receiver_params p{};
    int device = 0; // the device the handle was created on: its threads select it (hipSetDevice is per thread)
    gr4pm_syncword_detection* sd = nullptr;
    hipStream_t sd_stream = nullptr;
    struct Chain { // per-channel blocks (their state carries from batch to batch)
        gr4pm_syncword_detection_filter* sdf = nullptr;
        gr4pm_symbol_filter* symf = nullptr;
        gr4pm_syncword_wipeoff* wipe = nullptr;
        std::vector<uint64_t> idx;
        std::vector<gr4pm_header_msg> msgs;
        std::vector<uint8_t> accepted;
    };
    std::unique_ptr<Chain[]> chains; // [n_channels] (DevBuf members: neither copied nor moved)
    gr4pm_rotator* cfc = nullptr;        // n_channels channels
    gr4pm_costas_loop* costas = nullptr; // n_channels channels
    hipStream_t batch_stream = nullptr, costas_stream = nullptr;
    std::vector<hipStream_t> streams; // one per worker
    size_t y_stride = 0;
    struct ChanBatch { // what one batch holds per channel
        std::vector<gr4pm_tag> tags, sym_tags; // accepted detector tags, re-timed symbol tags
        size_t n_acc = 0, n_sym_tags = 0, produced = 0;
    };
    struct Slot { // one batch
        DevBuf<gr4pm_c64> y;      // SyncwordDetection's delayed output, [n_channels][y_stride]
        DevBuf<gr4pm_c64> symall; // symbol filter outputs, [n_channels][out_stride]
        std::vector<gr4pm_tag> det_tags; // [n_channels][tags_cap]
        std::vector<size_t> n_det;
        std::vector<ChanBatch> ch;
        std::vector<gr4pm_tag> all_tags;
        std::vector<uint32_t> all_channel;
        std::vector<size_t> produced;
        int plan = -1;
        const gr4pm_c64* in = nullptr; // in place: the batch's own input and the tail of the batch before
        size_t in_stride = 0;
        const gr4pm_c64* head = nullptr;
        size_t consumed = 0, out_stride = 0;
        uint64_t base = 0, packet_length = 0;
        gr4pm_c64* out_symbols = nullptr;
        gr4pm_status status = GR4PM_OK;
        char error[256] = { 0 };
        std::chrono::steady_clock::time_point t_submit, t_done;
    } slots[kMcSlots];
    // slot indices travel through the stages in submission order
    using Queue = hostlogic::SlotQueue<16>; // fixed ring, push() cannot throw (hostlogic/slot_queue.hpp)
    Queue to_stage1, to_stage2, to_stage3, done;
    std::thread t_stage1, t_stage2, t_stage3;
    int next_slot = 0, inflight = 0;
    // stage 2 fans a batch out to the workers
    Slot* cur = nullptr;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;
    unsigned pending = 0;
    bool quit = false;
    gr4pm_status wstatus = GR4PM_OK;
    char werror[256] = { 0 };

    // input in place (gr4pm_multichannel_receiver_set_input_in_place): no delayed copy; the last `hist` items of every
    // batch's input are kept (tails: one more than batches in flight) for the batch after it
    bool in_place = false;
    size_t hist = 0;
    size_t sd_fft = 0, sd_stride = 0;      // the detector's block size and overlap-save stride (syncword_detection.hpp:236-238)
    DevBuf<gr4pm_c64> tails[kMcSlots + 1]; // [n_channels][hist]
    int tail_next = 0;                     // where the NEXT submitted batch saves its tail
    // stage 2 as one launch per kernel: argument vectors of the ..._channels calls
    bool per_channel_launches = false;
    std::vector<gr4pm_symbol_filter*> v_symf;
    std::vector<gr4pm_syncword_wipeoff*> v_wipe;
    std::vector<const gr4pm_tag*> v_tags_in;
    std::vector<gr4pm_tag*> v_tags_out;
    std::vector<size_t> v_n_tags_in, v_n_tags_out, v_produced;

    gr4pm_status run_channel(Slot& s, size_t c);
    void worker(unsigned w);
    gr4pm_status stage1(Slot& s);
    gr4pm_status stage2(Slot& s);
    gr4pm_status stage3(Slot& s);
    void stage_loop(int which);
};

gr4pm_status gr4pm_multichannel_receiver::run_channel(Slot& s, size_t c)
{
    Chain& ch = chains[c];
    ChanBatch& cb = s.ch[c];
    const size_t cap = s.out_stride;
    size_t n_out_tags = 0, consumed = 0, produced_c = 0;
    GR4PM_TRY(gr4pm_cfc_symbol_filter_run_channel(cfc, s.plan, c, ch.symf, s.y.p + c * y_stride, s.consumed,
                                                  s.symall.p + c * s.out_stride, cap, cb.tags.data(), cb.n_acc,
                                                  cb.sym_tags.data(), cb.sym_tags.size(), &n_out_tags, &consumed,
                                                  &produced_c));
    GR4PM_TRY(gr4pm_syncword_wipeoff_process(ch.wipe, s.symall.p + c * s.out_stride, produced_c,
                                             s.symall.p + c * s.out_stride, cb.sym_tags.data(), n_out_tags));
    cb.n_sym_tags = n_out_tags;
    cb.produced = produced_c;
    return GR4PM_OK;
}

void gr4pm_multichannel_receiver::worker(unsigned w)
{
    (void)hipSetDevice(device);
    gr4pm_set_deferred_sync(1);
    uint64_t seen = 0;
    for (;;) {
        Slot* s = nullptr;
        {
            std::unique_lock<std::mutex> l(m);
            cv_go.wait(l, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
            s = cur;
        }
        gr4pm_status st = GR4PM_OK;
        const gr4pm_status gs = guarded("multichannel receiver, channel worker", [&] {
            for (size_t c = w; c < p.n_channels && st == GR4PM_OK; c += streams.size()) st = run_channel(*s, c);
        });
        if (gs != GR4PM_OK) st = gs;
        if (hipStreamSynchronize(streams[w]) != hipSuccess && st == GR4PM_OK) st = GR4PM_ERR_HIP;
        {
            std::lock_guard<std::mutex> l(m);
            if (st != GR4PM_OK && wstatus == GR4PM_OK) {
                wstatus = st;
                std::strncpy(werror, gr4pm_last_error(), sizeof(werror) - 1);
            }
            if (--pending == 0) cv_done.notify_all();
        }
    }
}

// stage 1: SyncwordDetectionFilter of every channel (host only: the samples pass unchanged, the tags are gated),
// then the CoarseFrequencyCorrection plan of all channels: one launch of the serial phasor checkpoints
gr4pm_status gr4pm_multichannel_receiver::stage1(Slot& s)
{
    const size_t C = p.n_channels;
    s.all_tags.clear();
    s.all_channel.clear();
    for (size_t c = 0; c < C; ++c) {
        auto& ch = chains[c];
        auto& cb = s.ch[c];
        const gr4pm_tag* dt = s.det_tags.data() + c * p.tags_cap;
        const size_t nd = s.n_det[c];
        ch.idx.resize(nd);
        ch.msgs.assign(std::max<size_t>(nd, 1), gr4pm_header_msg{ s.packet_length, s.packet_length == 0 ? 1 : 0 });
        for (size_t i = 0; i < nd; ++i) ch.idx[i] = s.base + dt[i].index;
        ch.accepted.assign(std::max<size_t>(nd, 1), 0);
        size_t used = 0;
        GR4PM_TRY(gr4pm_syncword_detection_filter_gate(ch.sdf, ch.idx.data(), nd, ch.msgs.data(), nd, 1,
                                                       ch.accepted.data(), &used));
        cb.n_acc = 0;
        for (size_t i = 0; i < nd; ++i)
            if (ch.accepted[i]) {
                cb.tags[cb.n_acc++] = dt[i];
                s.all_tags.push_back(dt[i]);
                s.all_channel.push_back(static_cast<uint32_t>(c));
            }
    }
    return gr4pm_cfc_symbol_filter_plan_channels(cfc, s.consumed, s.all_tags.data(), s.all_channel.data(),
                                                 s.all_tags.size(), &s.plan);
}

// stage 2: every channel's SymbolFilter (fused with its share of the CFC plan) + SyncwordWipeoff: ONE launch of
// each kernel for all channels (per_channel_launches: on the workers, channel by channel -- GR4PM_MC_PER_CHANNEL=1,
// kept for comparison: 64 channels of 2^22 items are 64 x 4 small kernels and 64 x 2 table uploads that way)
gr4pm_status gr4pm_multichannel_receiver::stage2(Slot& s)
{
    if (!per_channel_launches) {
        const size_t C = p.n_channels;
        for (size_t c = 0; c < C; ++c) {
            v_tags_in[c] = s.ch[c].tags.data();
            v_n_tags_in[c] = s.ch[c].n_acc;
            v_tags_out[c] = s.ch[c].sym_tags.data();
        }
        DeferredSyncScope defer;
        if (s.in)
            GR4PM_TRY(gr4pm_cfc_symbol_filter_run_channels(cfc, s.plan, v_symf.data(), C, s.in, s.in_stride, s.consumed,
                                                           s.symall.p, s.out_stride, v_tags_in.data(),
                                                           v_n_tags_in.data(), v_tags_out.data(),
                                                           s.ch[0].sym_tags.size(), v_n_tags_out.data(),
                                                           v_produced.data(), s.head, hist, hist));
        else
            GR4PM_TRY(gr4pm_cfc_symbol_filter_run_channels(cfc, s.plan, v_symf.data(), C, s.y.p, y_stride, s.consumed,
                                                           s.symall.p, s.out_stride, v_tags_in.data(),
                                                           v_n_tags_in.data(), v_tags_out.data(),
                                                           s.ch[0].sym_tags.size(), v_n_tags_out.data(),
                                                           v_produced.data(), nullptr, 0, 0));
        for (size_t c = 0; c < C; ++c) {
            s.ch[c].n_sym_tags = v_n_tags_out[c];
            s.ch[c].produced = v_produced[c];
            v_tags_in[c] = s.ch[c].sym_tags.data();
        }
        GR4PM_TRY(gr4pm_syncword_wipeoff_process_channels(v_wipe.data(), C, s.symall.p, s.out_stride, v_produced.data(),
                                                          v_tags_in.data(), v_n_tags_out.data()));
        GR4PM_HIP_TRY(hipStreamSynchronize(streams[0]));
        return GR4PM_OK;
    }
    std::unique_lock<std::mutex> l(m);
    wstatus = GR4PM_OK;
    cur = &s;
    pending = static_cast<unsigned>(workers.size());
    ++generation;
    cv_go.notify_all();
    cv_done.wait(l, [&] { return pending == 0; });
    if (wstatus != GR4PM_OK) {
        set_error("%s", werror);
        return wstatus;
    }
    return GR4PM_OK;
}

// stage 3: CostasLoop of all channels: one launch
gr4pm_status gr4pm_multichannel_receiver::stage3(Slot& s)
{
    const size_t C = p.n_channels;
    s.all_tags.clear();
    s.all_channel.clear();
    for (size_t c = 0; c < C; ++c) {
        auto& cb = s.ch[c];
        s.produced[c] = cb.produced;
        for (size_t i = 0; i < cb.n_sym_tags; ++i) {
            s.all_tags.push_back(cb.sym_tags[i]);
            s.all_channel.push_back(static_cast<uint32_t>(c));
        }
    }
    return gr4pm_costas_loop_process_ragged(costas, s.symall.p, s.out_stride, s.produced.data(), s.out_symbols,
                                            s.all_tags.data(), s.all_channel.data(), s.all_tags.size());
}

void gr4pm_multichannel_receiver::stage_loop(int which)
{
    (void)hipSetDevice(device);
    Queue& in = which == 1 ? to_stage1 : which == 2 ? to_stage2 : to_stage3;
    Queue& out = which == 1 ? to_stage2 : which == 2 ? to_stage3 : done;
    // hostlogic::run_stage: an exception inside a stage body fails that batch (collect() returns its status); the
    // end of the input is NOT forwarded -- destroy() stops every queue itself
    hostlogic::run_stage(
        in, &out, /*forward_quit=*/false,
        [&](int i) {
            Slot& s = slots[i];
            if (s.status == GR4PM_OK) { // a failed batch just travels on to collect()
                static const bool timing = getenv("GR4PM_MC_TIMING") != nullptr;
                const auto ta = std::chrono::steady_clock::now();
                const gr4pm_status st = which == 1 ? stage1(s) : which == 2 ? stage2(s) : stage3(s);
                if (timing)
                    fprintf(stderr, "[gr4pm multichannel] stage %d: %.0f us\n", which,
                            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ta).count());
                if (st != GR4PM_OK) {
                    s.status = st;
                    std::strncpy(s.error, gr4pm_last_error(), sizeof(s.error) - 1);
                }
            }
            if (which == 3) s.t_done = std::chrono::steady_clock::now();
        },
        [&](int i) {
            Slot& s = slots[i];
            s.status = exception_status(which == 1 ? "multichannel receiver, stage 1"
                                                   : which == 2 ? "multichannel receiver, stage 2" : "multichannel receiver, stage 3");
            std::strncpy(s.error, gr4pm_last_error(), sizeof(s.error) - 1);
            if (which == 3) s.t_done = std::chrono::steady_clock::now();
        });
}

extern "C" {

gr4pm_status gr4pm_multichannel_receiver_create(const gr4pm_multichannel_receiver_params* p,
                                                gr4pm_multichannel_receiver** out)
try {
    if (!p || !out || p->n_channels == 0 || p->samples_per_symbol == 0 || p->max_items < 2048) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_multichannel_receiver;
    if (!h) return GR4PM_ERR_NOMEM;
    h->p = *p;
    h->p.tags_cap = std::max<size_t>(p->tags_cap, 64);
    auto bail = [&](gr4pm_status st) {
        gr4pm_multichannel_receiver_destroy(h);
        return st;
    };
    const unsigned n_workers = static_cast<unsigned>(std::min<size_t>(std::max(p->workers, 1), p->n_channels));
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return bail(GR4PM_ERR_HIP);
    if (hipStreamCreateWithPriority(&h->sd_stream, hipStreamNonBlocking, greatest) != hipSuccess) return bail(GR4PM_ERR_HIP);
    h->streams.assign(n_workers, nullptr);
    for (unsigned w = 0; w < n_workers; ++w) {
        // streams of one priority share four hardware queues: spread the workers over the priorities
        const int prio = w % 3 == 0 ? 0 : (w % 3 == 1 ? greatest : least);
        if (hipStreamCreateWithPriority(&h->streams[w], hipStreamNonBlocking, prio) != hipSuccess) return bail(GR4PM_ERR_HIP);
    }
    const size_t sps = p->samples_per_symbol;
    // the constants of packet_receiver.hpp:37-122, as in gr4pm_packet_receiver_create
    std::vector<float> rrc(((sps * 11) | 1));
    const size_t n_rrc = gr4pm_firdes_root_raised_cosine(1.0, static_cast<double>(sps), 1.0, 0.35, sps * 11, rrc.data());
    if (n_rrc == 0) return bail(GR4PM_ERR_NOMEM); // the design ran out of host memory (its entry point reports that as 0 taps)
    rrc.resize(n_rrc);
    float norm = 0.0f;
    for (float v : rrc) norm += v * v;
    norm = std::sqrt(norm);
    for (float& v : rrc) v /= norm;
    static const uint8_t syncword[64] = { // 0x034776C7272895B0
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
    sp.n_channels = p->n_channels;
    sp.max_items = p->max_items;
    sp.stream = h->sd_stream;
    gr4pm_status st = gr4pm_syncword_detection_create(&sp, &h->sd);
    if (st != GR4PM_OK) return bail(st);
    gr4pm::sd_set_coresident(h->sd, true);
    const size_t arms = 32;
    std::vector<float> pfb(((arms * sps * 11) | 1));
    const size_t n_pfb = gr4pm_firdes_root_raised_cosine(static_cast<double>(arms) / static_cast<double>(norm),
                                                         static_cast<double>(arms * sps), 1.0, 0.35, arms * sps * 11,
                                                         pfb.data());
    if (n_pfb == 0) return bail(GR4PM_ERR_NOMEM);
    pfb.resize(n_pfb - 1);
    float bipolar[64];
    for (int i = 0; i < 64; ++i) bipolar[i] = syncword[i] ? -1.0f : 1.0f;
    h->chains.reset(new (std::nothrow) gr4pm_multichannel_receiver::Chain[p->n_channels]);
    if (!h->chains) return bail(GR4PM_ERR_NOMEM);
    if (hipStreamCreateWithPriority(&h->batch_stream, hipStreamNonBlocking, greatest) != hipSuccess) return bail(GR4PM_ERR_HIP);
    gr4pm_rotator_params rp{ 1, 0.0f, (rrc.size() - 1) / 2 + sps, p->n_channels, h->batch_stream };
    if ((st = gr4pm_rotator_create(&rp, &h->cfc)) != GR4PM_OK) return bail(st);
    if (hipStreamCreateWithPriority(&h->costas_stream, hipStreamNonBlocking, least) != hipSuccess) return bail(GR4PM_ERR_HIP);
    gr4pm_costas_loop_params cp{ 0.01, p->costas_constellation, p->n_channels, h->costas_stream };
    if ((st = gr4pm_costas_loop_create(&cp, &h->costas)) != GR4PM_OK) return bail(st);
    {
        // the 32-VGPR PLL form that pays for gr4pm_packet_receiver does not pay here (64 channels x 2^22 samples: 57.0
        // against 58.5 Gsps, latency 10.5 against 8.5 ms; smaller batches wait for the PLL's life time): the fast form
        // unless GR4PM_COSTAS_SMALL asks (A/B)
        static const char* small = gr4pm::experiment_env("GR4PM_COSTAS_SMALL", false);
        if (small) (void)gr4pm_costas_loop_set_small_footprint(h->costas, atoi(small));
    }
    for (size_t c = 0; c < p->n_channels; ++c) {
        auto& ch = h->chains[c];
        hipStream_t s = h->streams[c % n_workers];
        gr4pm_syncword_detection_filter_params fp{ sps, 64, 128, s };
        if ((st = gr4pm_syncword_detection_filter_create(&fp, &ch.sdf)) != GR4PM_OK) return bail(st);
        gr4pm_symbol_filter_params fsp{ sps, pfb.data(), pfb.size(), arms, rrc.size() - 1, 0, s };
        if ((st = gr4pm_symbol_filter_create(&fsp, &ch.symf)) != GR4PM_OK) return bail(st);
        gr4pm_syncword_wipeoff_params wp{ bipolar, 64, s };
        if ((st = gr4pm_syncword_wipeoff_create(&wp, &ch.wipe)) != GR4PM_OK) return bail(st);
    }
    {
        const char* e = getenv("GR4PM_MC_PER_CHANNEL");
        h->per_channel_launches = e && e[0] == '1';
        const size_t C = p->n_channels;
        h->v_symf.resize(C);
        h->v_wipe.resize(C);
        for (size_t c = 0; c < C; ++c) {
            h->v_symf[c] = h->chains[c].symf;
            h->v_wipe[c] = h->chains[c].wipe;
        }
        h->v_tags_in.assign(C, nullptr);
        h->v_tags_out.assign(C, nullptr);
        h->v_n_tags_in.assign(C, 0);
        h->v_n_tags_out.assign(C, 0);
        h->v_produced.assign(C, 0);
    }
    h->hist = 2 * 768 + 1; // SyncwordDetection's delay: 2 * time_threshold + 1 (syncword_detection.hpp:318-319)
    h->sd_fft = sp.fft_size;
    h->sd_stride = sp.fft_size - (63 * sps + rrc.size()) + 1;
    for (auto& t : h->tails) {
        if ((st = t.alloc(h->hist * p->n_channels)) != GR4PM_OK) return bail(st);
        if ((st = t.zero(h->sd_stream)) != GR4PM_OK) return bail(st);
    }
    if (hipStreamSynchronize(h->sd_stream) != hipSuccess) return bail(GR4PM_ERR_HIP);
    h->y_stride = (p->max_items + 63) & ~size_t{ 63 };
    for (auto& sl : h->slots) {
        if ((st = sl.y.alloc(h->y_stride * p->n_channels)) != GR4PM_OK) return bail(st);
        sl.det_tags.resize(p->n_channels * h->p.tags_cap);
        sl.n_det.assign(p->n_channels, 0);
        sl.produced.assign(p->n_channels, 0);
        sl.ch.resize(p->n_channels);
        for (auto& cb : sl.ch) {
            cb.tags.resize(h->p.tags_cap);
            cb.sym_tags.resize(h->p.tags_cap + 64);
        }
    }
    // HIP's current device is per thread and starts at 0: every thread of the handle works on the creator's device
    if (hipGetDevice(&h->device) != hipSuccess) return bail(GR4PM_ERR_HIP);
    try {
        h->workers.reserve(n_workers);
        for (unsigned w = 0; w < n_workers; ++w) h->workers.emplace_back([h, w] { h->worker(w); });
        h->t_stage1 = std::thread([h] { h->stage_loop(1); });
        h->t_stage2 = std::thread([h] { h->stage_loop(2); });
        h->t_stage3 = std::thread([h] { h->stage_loop(3); });
    } catch (...) { // a thread could not start: wind down the ones that did
        return bail(exception_status("gr4pm_multichannel_receiver_create (threads)"));
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

void gr4pm_multichannel_receiver_destroy(gr4pm_multichannel_receiver* h)
try {
    if (!h) return;
    {
        std::lock_guard<std::mutex> l(h->m);
        h->quit = true;
    }
    h->cv_go.notify_all();
    h->to_stage1.stop();
    h->to_stage2.stop();
    h->to_stage3.stop();
    h->done.stop();
    for (std::thread* t : { &h->t_stage1, &h->t_stage2, &h->t_stage3 })
        if (t->joinable()) t->join();
    for (auto& t : h->workers)
        if (t.joinable()) t.join();
    gr4pm_syncword_detection_destroy(h->sd);
    for (size_t c = 0; h->chains && c < h->p.n_channels; ++c) {
        auto& ch = h->chains[c];
        gr4pm_syncword_detection_filter_destroy(ch.sdf);
        gr4pm_symbol_filter_destroy(ch.symf);
        gr4pm_syncword_wipeoff_destroy(ch.wipe);
    }
    gr4pm_rotator_destroy(h->cfc);
    gr4pm_costas_loop_destroy(h->costas);
    if (h->batch_stream) (void)hipStreamDestroy(h->batch_stream);
    if (h->costas_stream) (void)hipStreamDestroy(h->costas_stream);
    for (auto s : h->streams)
        if (s) (void)hipStreamDestroy(s);
    if (h->sd_stream) (void)hipStreamDestroy(h->sd_stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID

gr4pm_status gr4pm_multichannel_receiver_announce(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                  size_t in_stride, size_t n_in)
try {
    if (!h || !in) return GR4PM_ERR_INVALID;
    return gr4pm_syncword_detection_announce(h->sd, in, in_stride, n_in);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_multichannel_receiver_set_input_in_place(gr4pm_multichannel_receiver* h, int on)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (h->inflight != 0 || gr4pm_syncword_detection_items_consumed(h->sd) != 0) {
        set_error("set_input_in_place: before the first batch");
        return GR4PM_ERR_INVALID;
    }
    h->in_place = on != 0;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_multichannel_receiver_submit(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                size_t in_stride, size_t n_in, uint64_t packet_length,
                                                gr4pm_c64* out_symbols, size_t out_stride, size_t* consumed)
try {
    if (!h || !in || !out_symbols || !consumed) return GR4PM_ERR_INVALID;
    *consumed = 0;
    // checked BEFORE the detector consumes the batch: an error here must not lose items (the detector
    // consumes at most n_in items per channel)
    if (out_stride < n_in / h->p.samples_per_symbol + h->p.tags_cap + 2) {
        set_error("out_stride %zu too small for %zu items per channel", out_stride, n_in);
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    if (h->inflight >= kMcSlots) {
        set_error("%d batches in flight: collect one first", h->inflight);
        return GR4PM_ERR_INVALID;
    }
    const bool in_place = h->in_place && !h->per_channel_launches;
    if (in_place && n_in >= h->sd_fft) {
        // what the detector will consume is known up front (hpp:238): a batch too short to leave a whole tail is
        // refused HERE, while the detector's state has not moved
        const size_t will_consume = ((n_in - h->sd_fft) / h->sd_stride + 1) * h->sd_stride;
        if (will_consume < h->hist) {
            set_error("in-place input needs batches of at least %zu consumed items (this one: %zu)", h->hist, will_consume);
            return GR4PM_ERR_INVALID;
        }
    }
    auto& s = h->slots[h->next_slot];
    const size_t C = h->p.n_channels;
    if (s.symall.n < C * out_stride) GR4PM_TRY(s.symall.alloc(C * out_stride));
    s.t_submit = std::chrono::steady_clock::now();
    size_t n_done = 0;
    // stage 0, in the caller's thread: the batched detector (its look-ahead runs the correlator of the announced
    // batches behind this call's own kernels)
    const gr4pm_status st = gr4pm_syncword_detection_process(h->sd, in, in_stride, n_in, in_place ? nullptr : s.y.p,
                                                             h->y_stride, &n_done, s.det_tags.data(), h->p.tags_cap,
                                                             s.n_det.data());
    if (st != GR4PM_OK) return st;
    s.in = nullptr;
    if (in_place) {
        // this batch reads the tail the batch before it saved; its own tail (the last hist items it consumed) goes
        // to the next buffer of the ring, for the batch after it
        constexpr int kTails = kMcSlots + 1;
        s.in = in;
        s.in_stride = in_stride;
        s.head = h->tails[(h->tail_next + kTails - 1) % kTails].p;
        if (n_done >= h->hist) {
            GR4PM_HIP_TRY(hipMemcpy2DAsync(h->tails[h->tail_next].p, h->hist * sizeof(gr4pm_c64),
                                           in + (n_done - h->hist), in_stride * sizeof(gr4pm_c64),
                                           h->hist * sizeof(gr4pm_c64), C, hipMemcpyDeviceToDevice, h->sd_stream));
            GR4PM_HIP_TRY(hipStreamSynchronize(h->sd_stream));
            h->tail_next = (h->tail_next + 1) % kTails;
        } else if (n_done != 0) {
            set_error("in-place input needs batches of at least %zu consumed items", h->hist);
            return GR4PM_ERR_INVALID;
        }
    }
    if (getenv("GR4PM_MC_TIMING"))
        fprintf(stderr, "[gr4pm multichannel] stage 0: %.0f us\n",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s.t_submit).count());
    *consumed = n_done;
    s.consumed = n_done;
    s.base = gr4pm_syncword_detection_items_consumed(h->sd) - n_done;
    s.packet_length = packet_length;
    s.out_symbols = out_symbols;
    s.out_stride = out_stride;
    s.status = GR4PM_OK;
    const int i = h->next_slot;
    h->next_slot = (h->next_slot + 1) % kMcSlots;
    ++h->inflight;
    h->to_stage1.push(i);
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_multichannel_receiver_collect(gr4pm_multichannel_receiver* h, size_t* consumed, size_t* n_symbols,
                                                 gr4pm_tag* tags, size_t* n_tags, gr4pm_tag* detector_tags,
                                                 size_t* n_detector_tags)
try {
    if (!h || !n_symbols) return GR4PM_ERR_INVALID;
    if (h->inflight == 0) {
        set_error("nothing in flight");
        return GR4PM_ERR_INVALID;
    }
    const int i = h->done.pop();
    if (i < 0) return GR4PM_ERR_INVALID;
    --h->inflight;
    auto& s = h->slots[i];
    const size_t C = h->p.n_channels;
    if (consumed) *consumed = s.consumed;
    if (s.status != GR4PM_OK) {
        for (size_t c = 0; c < C; ++c) n_symbols[c] = 0;
        set_error("%s", s.error);
        return s.status;
    }
    for (size_t c = 0; c < C; ++c) {
        const auto& cb = s.ch[c];
        n_symbols[c] = cb.produced;
        if (n_tags) n_tags[c] = cb.n_sym_tags;
        if (tags) std::memcpy(tags + c * h->p.tags_cap, cb.sym_tags.data(), std::min(cb.n_sym_tags, h->p.tags_cap) * sizeof(gr4pm_tag));
        if (n_detector_tags) n_detector_tags[c] = s.n_det[c];
    }
    if (detector_tags) std::memcpy(detector_tags, s.det_tags.data(), s.det_tags.size() * sizeof(gr4pm_tag));
    static const bool timing = getenv("GR4PM_MC_TIMING") != nullptr;
    if (timing)
        fprintf(stderr, "[gr4pm multichannel] batch latency %.0f us\n",
                std::chrono::duration<double, std::micro>(s.t_done - s.t_submit).count());
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

int gr4pm_multichannel_receiver_in_flight(const gr4pm_multichannel_receiver* h) { return h ? h->inflight : 0; }

gr4pm_status gr4pm_multichannel_receiver_process(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                 size_t in_stride, size_t n_in, uint64_t packet_length,
                                                 gr4pm_c64* out_symbols, size_t out_stride, size_t* consumed,
                                                 size_t* n_symbols, gr4pm_tag* tags, size_t* n_tags,
                                                 gr4pm_tag* detector_tags, size_t* n_detector_tags)
try {
    if (!h || !in || !out_symbols || !consumed || !n_symbols) return GR4PM_ERR_INVALID;
    if (h->inflight != 0) {
        set_error("process() with batches in flight: collect them first");
        return GR4PM_ERR_INVALID;
    }
    for (size_t c = 0; c < h->p.n_channels; ++c) n_symbols[c] = 0;
    GR4PM_TRY(gr4pm_multichannel_receiver_submit(h, in, in_stride, n_in, packet_length, out_symbols, out_stride, consumed));
    return gr4pm_multichannel_receiver_collect(h, nullptr, n_symbols, tags, n_tags, detector_tags, n_detector_tags);
}
GR4PM_ABI_CATCH

} // extern "C"
