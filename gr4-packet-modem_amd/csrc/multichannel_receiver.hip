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
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

using namespace gr4pm;

struct gr4pm_multichannel_receiver {
    gr4pm_multichannel_receiver_params p{};
    gr4pm_syncword_detection* sd = nullptr;
    hipStream_t sd_stream = nullptr;
    struct Chain {
        gr4pm_syncword_detection_filter* sdf = nullptr;
        gr4pm_symbol_filter* symf = nullptr;
        gr4pm_syncword_wipeoff* wipe = nullptr;
        std::vector<gr4pm_tag> tags, sym_tags;
        size_t n_acc = 0, n_sym_tags = 0, produced = 0;
        std::vector<uint64_t> idx;
        std::vector<gr4pm_header_msg> msgs;
        std::vector<uint8_t> accepted;
    };
    std::unique_ptr<Chain[]> chains; // [n_channels] (DevBuf members: neither copied nor moved)
    gr4pm_rotator* cfc = nullptr;      // n_channels channels
    gr4pm_costas_loop* costas = nullptr; // n_channels channels
    hipStream_t batch_stream = nullptr;  // the two batched handles
    int plan = -1;
    DevBuf<gr4pm_c64> symall;            // symbol filter outputs, [n_channels][out_stride]
    std::vector<gr4pm_tag> all_tags;     // accepted tags / symbol tags of all channels, concatenated
    std::vector<uint32_t> all_channel;
    std::vector<size_t> produced;
    std::vector<hipStream_t> streams; // one per worker
    DevBuf<gr4pm_c64> y;              // SyncwordDetection's delayed output, [n_channels][y_stride]
    size_t y_stride = 0;
    std::vector<gr4pm_tag> det_tags;  // [n_channels][tags_cap]
    std::vector<size_t> n_det;
    // one job = one process() call, fanned out to the workers
    struct Job {
        size_t consumed = 0;
        uint64_t base = 0, packet_length = 0;
        gr4pm_c64* out_symbols = nullptr;
        size_t out_stride = 0;
        size_t* n_symbols = nullptr;
        gr4pm_tag* tags = nullptr;
        size_t* n_tags = nullptr;
    } job;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;
    unsigned pending = 0;
    bool quit = false;
    gr4pm_status status = GR4PM_OK;
    char error[256] = { 0 };

    gr4pm_status run_channel(size_t c);
    void worker(unsigned w);
};

gr4pm_status gr4pm_multichannel_receiver::run_channel(size_t c)
{
    Chain& ch = chains[c];
    const size_t cap = job.out_stride;
    size_t n_out_tags = 0, consumed = 0, produced_c = 0;
    GR4PM_TRY(gr4pm_cfc_symbol_filter_run_channel(cfc, plan, c, ch.symf, y.p + c * y_stride, job.consumed,
                                                  symall.p + c * job.out_stride, cap, ch.tags.data(), ch.n_acc,
                                                  ch.sym_tags.data(), ch.sym_tags.size(), &n_out_tags, &consumed,
                                                  &produced_c));
    GR4PM_TRY(gr4pm_syncword_wipeoff_process(ch.wipe, symall.p + c * job.out_stride, produced_c,
                                             symall.p + c * job.out_stride, ch.sym_tags.data(), n_out_tags));
    ch.n_sym_tags = n_out_tags;
    ch.produced = produced_c;
    return GR4PM_OK;
}

void gr4pm_multichannel_receiver::worker(unsigned w)
{
    gr4pm_set_deferred_sync(1);
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> l(m);
            cv_go.wait(l, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
        }
        gr4pm_status st = GR4PM_OK;
        for (size_t c = w; c < p.n_channels && st == GR4PM_OK; c += streams.size()) st = run_channel(c);
        if (hipStreamSynchronize(streams[w]) != hipSuccess && st == GR4PM_OK) st = GR4PM_ERR_HIP;
        {
            std::lock_guard<std::mutex> l(m);
            if (st != GR4PM_OK && status == GR4PM_OK) {
                status = st;
                std::strncpy(error, gr4pm_last_error(), sizeof(error) - 1);
            }
            if (--pending == 0) cv_done.notify_all();
        }
    }
}

extern "C" {

gr4pm_status gr4pm_multichannel_receiver_create(const gr4pm_multichannel_receiver_params* p,
                                                gr4pm_multichannel_receiver** out)
{
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
    const size_t arms = 32;
    std::vector<float> pfb(((arms * sps * 11) | 1));
    const size_t n_pfb = gr4pm_firdes_root_raised_cosine(static_cast<double>(arms) / static_cast<double>(norm),
                                                         static_cast<double>(arms * sps), 1.0, 0.35, arms * sps * 11,
                                                         pfb.data());
    pfb.resize(n_pfb - 1);
    float bipolar[64];
    for (int i = 0; i < 64; ++i) bipolar[i] = syncword[i] ? -1.0f : 1.0f;
    h->chains.reset(new (std::nothrow) gr4pm_multichannel_receiver::Chain[p->n_channels]);
    if (!h->chains) return bail(GR4PM_ERR_NOMEM);
    if (hipStreamCreateWithPriority(&h->batch_stream, hipStreamNonBlocking, greatest) != hipSuccess) return bail(GR4PM_ERR_HIP);
    gr4pm_rotator_params rp{ 1, 0.0f, (rrc.size() - 1) / 2 + sps, p->n_channels, h->batch_stream };
    if ((st = gr4pm_rotator_create(&rp, &h->cfc)) != GR4PM_OK) return bail(st);
    gr4pm_costas_loop_params cp{ 0.01, p->costas_constellation, p->n_channels, h->batch_stream };
    if ((st = gr4pm_costas_loop_create(&cp, &h->costas)) != GR4PM_OK) return bail(st);
    for (size_t c = 0; c < p->n_channels; ++c) {
        auto& ch = h->chains[c];
        hipStream_t s = h->streams[c % n_workers];
        gr4pm_syncword_detection_filter_params fp{ sps, 64, 128, s };
        if ((st = gr4pm_syncword_detection_filter_create(&fp, &ch.sdf)) != GR4PM_OK) return bail(st);
        gr4pm_symbol_filter_params fsp{ sps, pfb.data(), pfb.size(), arms, rrc.size() - 1, 0, s };
        if ((st = gr4pm_symbol_filter_create(&fsp, &ch.symf)) != GR4PM_OK) return bail(st);
        gr4pm_syncword_wipeoff_params wp{ bipolar, 64, s };
        if ((st = gr4pm_syncword_wipeoff_create(&wp, &ch.wipe)) != GR4PM_OK) return bail(st);
        ch.tags.resize(h->p.tags_cap);
        ch.sym_tags.resize(h->p.tags_cap + 64);
    }
    h->produced.assign(p->n_channels, 0);
    h->y_stride = (p->max_items + 63) & ~size_t{ 63 };
    if ((st = h->y.alloc(h->y_stride * p->n_channels)) != GR4PM_OK) return bail(st);
    h->det_tags.resize(p->n_channels * h->p.tags_cap);
    h->n_det.assign(p->n_channels, 0);
    for (unsigned w = 0; w < n_workers; ++w) h->workers.emplace_back([h, w] { h->worker(w); });
    *out = h;
    return GR4PM_OK;
}

void gr4pm_multichannel_receiver_destroy(gr4pm_multichannel_receiver* h)
{
    if (!h) return;
    {
        std::lock_guard<std::mutex> l(h->m);
        h->quit = true;
    }
    h->cv_go.notify_all();
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
    for (auto s : h->streams)
        if (s) (void)hipStreamDestroy(s);
    if (h->sd_stream) (void)hipStreamDestroy(h->sd_stream);
    delete h;
}

gr4pm_status gr4pm_multichannel_receiver_announce(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                  size_t in_stride, size_t n_in)
{
    if (!h || !in) return GR4PM_ERR_INVALID;
    return gr4pm_syncword_detection_announce(h->sd, in, in_stride, n_in);
}

gr4pm_status gr4pm_multichannel_receiver_process(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                 size_t in_stride, size_t n_in, uint64_t packet_length,
                                                 gr4pm_c64* out_symbols, size_t out_stride, size_t* consumed,
                                                 size_t* n_symbols, gr4pm_tag* tags, size_t* n_tags,
                                                 gr4pm_tag* detector_tags, size_t* n_detector_tags)
{
    if (!h || !in || !out_symbols || !consumed || !n_symbols) return GR4PM_ERR_INVALID;
    *consumed = 0;
    for (size_t c = 0; c < h->p.n_channels; ++c) n_symbols[c] = 0;
    static const bool timing = getenv("GR4PM_MC_TIMING") != nullptr; // wall time of the four phases, to stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t0 = now();
    // checked BEFORE the detector consumes the batch: an error here must not lose items (the detector
    // consumes at most n_in items per channel)
    if (out_stride < n_in / h->p.samples_per_symbol + h->p.tags_cap + 2) {
        set_error("out_stride %zu too small for %zu items per channel", out_stride, n_in);
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    size_t n_done = 0;
    const gr4pm_status st = gr4pm_syncword_detection_process(h->sd, in, in_stride, n_in, h->y.p, h->y_stride, &n_done,
                                                             h->det_tags.data(), h->p.tags_cap, h->n_det.data());
    if (st != GR4PM_OK) return st;
    *consumed = n_done;
    if (detector_tags) std::memcpy(detector_tags, h->det_tags.data(), h->det_tags.size() * sizeof(gr4pm_tag));
    if (n_detector_tags)
        for (size_t c = 0; c < h->p.n_channels; ++c) n_detector_tags[c] = h->n_det[c];
    const auto t1 = now();
    h->job.consumed = n_done;
    h->job.base = gr4pm_syncword_detection_items_consumed(h->sd) - n_done;
    h->job.packet_length = packet_length;
    h->job.out_symbols = out_symbols;
    h->job.out_stride = out_stride;
    h->job.n_symbols = n_symbols;
    h->job.tags = tags;
    h->job.n_tags = n_tags;
    const size_t C = h->p.n_channels;
    if (h->symall.n < C * out_stride) GR4PM_TRY(h->symall.alloc(C * out_stride));
    // SyncwordDetectionFilter of every channel (host only): the samples pass unchanged, the tags are gated
    h->all_tags.clear();
    h->all_channel.clear();
    for (size_t c = 0; c < C; ++c) {
        auto& ch = h->chains[c];
        const gr4pm_tag* dt = h->det_tags.data() + c * h->p.tags_cap;
        const size_t nd = h->n_det[c];
        ch.idx.resize(nd);
        ch.msgs.assign(std::max<size_t>(nd, 1), gr4pm_header_msg{ packet_length, packet_length == 0 ? 1 : 0 });
        for (size_t i = 0; i < nd; ++i) ch.idx[i] = h->job.base + dt[i].index;
        ch.accepted.assign(std::max<size_t>(nd, 1), 0);
        size_t used = 0;
        GR4PM_TRY(gr4pm_syncword_detection_filter_gate(ch.sdf, ch.idx.data(), nd, ch.msgs.data(), nd, 1,
                                                       ch.accepted.data(), &used));
        ch.n_acc = 0;
        for (size_t i = 0; i < nd; ++i)
            if (ch.accepted[i]) {
                ch.tags[ch.n_acc++] = dt[i];
                h->all_tags.push_back(dt[i]);
                h->all_channel.push_back(static_cast<uint32_t>(c));
            }
    }
    // CoarseFrequencyCorrection of all channels: one plan, one launch of the serial checkpoints
    GR4PM_TRY(gr4pm_cfc_symbol_filter_plan_channels(h->cfc, n_done, h->all_tags.data(), h->all_channel.data(),
                                                    h->all_tags.size(), &h->plan));
    const auto t2 = now();
    // every channel's SymbolFilter + SyncwordWipeoff on the workers
    {
        std::unique_lock<std::mutex> l(h->m);
        h->status = GR4PM_OK;
        h->pending = static_cast<unsigned>(h->workers.size());
        ++h->generation;
        h->cv_go.notify_all();
        h->cv_done.wait(l, [&] { return h->pending == 0; });
        if (h->status != GR4PM_OK) {
            set_error("%s", h->error);
            return h->status;
        }
    }
    const auto t3 = now();
    // CostasLoop of all channels: one launch
    h->all_tags.clear();
    h->all_channel.clear();
    for (size_t c = 0; c < C; ++c) {
        auto& ch = h->chains[c];
        h->produced[c] = ch.produced;
        n_symbols[c] = ch.produced;
        if (n_tags) n_tags[c] = ch.n_sym_tags;
        if (tags) std::memcpy(tags + c * h->p.tags_cap, ch.sym_tags.data(), std::min(ch.n_sym_tags, h->p.tags_cap) * sizeof(gr4pm_tag));
        for (size_t i = 0; i < ch.n_sym_tags; ++i) {
            h->all_tags.push_back(ch.sym_tags[i]);
            h->all_channel.push_back(static_cast<uint32_t>(c));
        }
    }
    const gr4pm_status cst = gr4pm_costas_loop_process_ragged(h->costas, h->symall.p, out_stride, h->produced.data(),
                                                              out_symbols, h->all_tags.data(), h->all_channel.data(),
                                                              h->all_tags.size());
    if (timing) {
        static auto last_return = t0;
        const auto t4 = now();
        fprintf(stderr, "[gr4pm multichannel] caller %.0f us | detector %.0f us, gate + CFC plan %.0f us, symbol filters %.0f us, Costas %.0f us\n",
                us(last_return, t0), us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
        last_return = t4;
    }
    return cst;
}

} // extern "C"
