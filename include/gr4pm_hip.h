/*
 * gr4pm_hip.h -- C ABI of the MI355X-native gr4-packet-modem RX hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The
 * gr::Block<T> wrappers in gr4-packet-modem_amd/host/ (and the ctypes binding used by the
 * tests) call exactly these entry points.  Each entry point cites the reference interface
 * it replaces; paths are relative to
 *   /root/reference/blocks/include/gnuradio-4.0/packet-modem/
 *
 * Conventions
 *  - Sample pointers (`in`, `out`) are DEVICE pointers (HBM resident) unless a function
 *    says otherwise; complex samples are interleaved float32 pairs == std::complex<float>.
 *  - Tag arrays, message arrays, settings and counters are HOST pointers.
 *  - Tags are passed with explicit item indices relative to in[0] of the call (the GR4
 *    runtime presents a tag only at the head of a chunk; the wrapper passes index 0).
 *  - One handle == one HIP stream == single-threaded use, like one gr::Block instance.
 *  - No exceptions cross the ABI.  Negative status == what the reference block throws as
 *    gr::exception; positive status == gr::work::Status other than OK.
 *  - Batched handles (n_channels > 1) process independent channels laid out as
 *    [channel][stride] in one launch; every channel has its own state.
 */
#ifndef GR4PM_HIP_H
#define GR4PM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } gr4pm_c64;

typedef enum {
    GR4PM_OK = 0,
    GR4PM_INSUFFICIENT_INPUT_ITEMS = 1,  /* gr::work::Status::INSUFFICIENT_INPUT_ITEMS */
    GR4PM_INSUFFICIENT_OUTPUT_ITEMS = 2, /* gr::work::Status::INSUFFICIENT_OUTPUT_ITEMS */
    GR4PM_ERR_INVALID = -1,              /* invalid settings (reference: throw gr::exception) */
    GR4PM_ERR_UNSUPPORTED = -2,          /* valid in the reference, not built here yet */
    GR4PM_ERR_HIP = -3,                  /* HIP runtime failure, see gr4pm_last_error() */
    GR4PM_ERR_NOMEM = -4,
    GR4PM_ERR_OVERFLOW = -5,             /* a caller-provided tag/record array was too small */
    GR4PM_ERR_NO_DEVICE = -6,            /* no HIP device: the product has no CPU fallback */
    GR4PM_ERR_INTERNAL = -7              /* a C++ exception other than std::bad_alloc (which is GR4PM_ERR_NOMEM) was caught
                                            at the ABI or in one of the handle's threads: gr4pm_last_error() has what() */
} gr4pm_status;

/* Human readable text of the last failure on the calling thread. */
const char* gr4pm_last_error(void);
/* Library version / build info; also proves the HIP code objects are loaded. */
const char* gr4pm_version(void);
/* Number of visible HIP devices (0 when none; does not create a context). */
int gr4pm_device_count(void);
/* For native callers that chain several handles on ONE HIP stream without reading anything back in
 * between: while on (per calling thread), the process() calls that return nothing device-produced
 * to the host (symbol filter, wipe-off, Costas loop, PayloadMetadataInsert, SyncwordRemove, LLR
 * decoder, scrambler, HeaderPayloadSplit, slicer / packer) do not wait for their stream; the
 * caller synchronises the stream before it reads their outputs, hands them to another stream, or
 * calls the same handle again. */
void gr4pm_set_deferred_sync(int on);

/* The `syncword_*` tag set published by SyncwordDetection (syncword_detection.hpp:106-114)
 * and consumed downstream (symbol_filter.hpp:130-156, coarse_frequency_correction.hpp:78-80,
 * costas_loop.hpp:101-106, syncword_wipeoff.hpp:53-62). */
typedef struct {
    uint64_t index;    /* item index the tag is attached to */
    float amplitude;   /* "syncword_amplitude" */
    float phase;       /* "syncword_phase" */
    double freq;       /* "syncword_freq" (double in the reference) */
    int32_t freq_bin;  /* "syncword_freq_bin" */
    float noise_power; /* "syncword_noise_power" */
    float esn0_db;     /* "syncword_esn0_db" */
    float time_est;    /* "syncword_time_est" */
    int32_t flags;     /* GR4PM_TAG_* */
    int32_t user;      /* the caller's own cookie: never read by the library, copied with the tag by every block that
                          re-emits it (SymbolFilter's re-timed tags, :218-228) -- a binding keeps the full property map
                          of a tag under this handle; SyncwordDetection writes 0.  (Round 6: the GR4 wrapper used to
                          carry its handle in freq_bin; the field takes the struct's former tail padding, size 48.) */
} gr4pm_tag;
#if defined(__cplusplus)
static_assert(sizeof(gr4pm_tag) == 48, "gr4pm_tag is 48 bytes (tests and bindings mirror it)");
#endif
#define GR4PM_TAG_SYNCWORD 1 /* carries the syncword_* keys */
#define GR4PM_TAG_OTHER 2    /* carries other (opaque, wrapper-held) keys */

/* ------------------------------------------------------------------------------------
 * SyncwordDetection -- syncword_detection.hpp:32-357
 *   settings  :133-141      start() :143-202      processBulk() :204-356
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_syncword_detection gr4pm_syncword_detection;
typedef struct {
    size_t fft_size;           /* :133, default 2048 (tuned one-wave path); other powers of two
                                  256..8192 run the generic workgroup-per-block path */
    size_t samples_per_symbol; /* :134 */
    const float* rrc_taps;     /* :135 host */
    size_t n_rrc_taps;
    const uint8_t* syncword;   /* :136 host */
    size_t n_syncword;
    const gr4pm_c64* constellation; /* :137 host */
    size_t n_constellation;
    int min_freq_bin;          /* :138 */
    int max_freq_bin;          /* :139 */
    uint64_t time_threshold;   /* :140 */
    float power_threshold;     /* :141 */
    size_t n_channels;         /* independent channels per call (>= 1) */
    size_t max_items;          /* largest n_in per channel per call (workspace sizing) */
    void* stream;              /* hipStream_t, NULL = default stream */
} gr4pm_syncword_detection_params;

gr4pm_status gr4pm_syncword_detection_create(const gr4pm_syncword_detection_params* params,
                                             gr4pm_syncword_detection** out);
void gr4pm_syncword_detection_destroy(gr4pm_syncword_detection* h);
/* == start(): clears _best/_best_idx/_items_consumed/_history (:191-199) */
gr4pm_status gr4pm_syncword_detection_reset(gr4pm_syncword_detection* h);
/* public state the reference exposes and its tests read (test/qa_syncword_detection.cpp:101-133) */
size_t gr4pm_syncword_detection_syncword_samples_size(const gr4pm_syncword_detection* h);
float gr4pm_syncword_detection_self_corr(const gr4pm_syncword_detection* h);
uint64_t gr4pm_syncword_detection_items_consumed(const gr4pm_syncword_detection* h);
/* Diagnostics of the last process() call (no reference counterpart): how many candidates the scan visited in the channel
 * (the items whose history the reference tests, :268-279) and how many of those were tested by the separate pass over
 * the powers instead of on the candidate kernel's registers (all of them for a time_threshold other than 768). */
void gr4pm_syncword_detection_scan_counts(const gr4pm_syncword_detection* h, size_t channel, uint64_t* visited,
                                          uint64_t* tested_from_memory);
/* == processBulk().  in: [n_channels][in_stride] items, n_in valid per channel.
 * out: [n_channels][out_stride] or NULL (skip the delayed pass-through copy when the
 * consumer reads the input ring itself).  *n_done = items consumed == published per channel
 * (whole strides only, :238,346-350).  tags: host [n_channels][tags_cap], n_tags: host
 * [n_channels]; tag.index is relative to out[0] of this call.  Returns
 * GR4PM_INSUFFICIENT_INPUT_ITEMS when n_in < fft_size (:215-227). */
gr4pm_status gr4pm_syncword_detection_process(gr4pm_syncword_detection* h, const gr4pm_c64* in,
                                              size_t in_stride, size_t n_in, gr4pm_c64* out,
                                              size_t out_stride, size_t* n_done, gr4pm_tag* tags,
                                              size_t tags_cap, size_t* n_tags);
/* Optional look-ahead for callers that own a device ring (no reference counterpart: the
 * reference's scheduler overlaps blocks across worker threads instead).  _announce names the
 * input of a future call: the one after the next process() and after the calls already
 * announced (at most GR4PM_SD_LOOKAHEAD = 2 calls ahead are kept; further announcements are
 * ignored).  The next process() then also launches, on two more streams, everything of the
 * announced calls that does not depend on the scan state of the calls before them (z carry,
 * correlator, candidate bitmap, tile / group tables), where it overlaps this call's scan, tag
 * kernels and read-back; the later process(in, in_stride, n_in) with exactly the announced
 * arguments finds that work done.  The items must not change between the announcement and
 * their call, and whatever produces them must have been queued on the handle's stream (or have
 * finished) when they are announced: the look-ahead streams wait for an event recorded on the
 * handle's stream at announcement time before they read an announced buffer.  A call with other arguments drops the look-ahead and recomputes; results are
 * identical either way.  _hint_next is the one-call form: it replaces every announcement not
 * yet launched (in_next == NULL: just clears them). */
#define GR4PM_SD_LOOKAHEAD 2
gr4pm_status gr4pm_syncword_detection_announce(gr4pm_syncword_detection* h, const gr4pm_c64* in,
                                               size_t in_stride, size_t n_in);
gr4pm_status gr4pm_syncword_detection_hint_next(gr4pm_syncword_detection* h,
                                                const gr4pm_c64* in_next, size_t in_stride,
                                                size_t n_next);
/* debug / measurement: copies the per-sample best-bin correlation power of the last call
 * (device [n_channels][n_done] floats, items_consumed-relative) into `zpow` (device). */
gr4pm_status gr4pm_syncword_detection_last_zpow(gr4pm_syncword_detection* h, float* zpow,
                                                size_t stride);
/* measurement hook: runs ONLY the overlap-save correlator kernel on [n_channels][in_stride]
 * input (no detector, no copy); used by bench.py to time the dominant kernel in isolation. */
gr4pm_status gr4pm_syncword_detection_correlate_only(gr4pm_syncword_detection* h,
                                                     const gr4pm_c64* in, size_t in_stride,
                                                     size_t n_in);

/* ------------------------------------------------------------------------------------
 * SyncwordDetectionFilter<T = c64> -- syncword_detection_filter.hpp:10-211
 * One call == one processBulk() (:54-210) with the messages pending on the two async
 * ports.  Samples are copied device-to-device; the tag gate runs on the host.
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_syncword_detection_filter gr4pm_syncword_detection_filter;
typedef struct {
    size_t samples_per_symbol; /* :44 */
    size_t syncword_size;      /* :45 */
    size_t header_size;        /* :46 */
    void* stream;
} gr4pm_syncword_detection_filter_params;
typedef struct {
    uint64_t packet_length; /* "packet_length" of a parsed_header message */
    int32_t invalid_header; /* 1: message carries "invalid_header"; 2 (headers_per_tag modes only):
                               no message yet, see ..._gate_resolve / ..._insert_resolve */
} gr4pm_header_msg;
gr4pm_status gr4pm_syncword_detection_filter_create(
    const gr4pm_syncword_detection_filter_params* params, gr4pm_syncword_detection_filter** out);
void gr4pm_syncword_detection_filter_destroy(gr4pm_syncword_detection_filter* h);
gr4pm_status gr4pm_syncword_detection_filter_reset(gr4pm_syncword_detection_filter* h);
/* head_tag_flags: GR4PM_TAG_* of the tag at in[0], 0 if none.  *tag_out_flags: which key
 * classes are published at out[0] (:82-104). */
gr4pm_status gr4pm_syncword_detection_filter_process(
    gr4pm_syncword_detection_filter* h, const gr4pm_c64* in, size_t n_in, gr4pm_c64* out,
    size_t out_cap, int head_tag_flags, const gr4pm_header_msg* headers, size_t n_headers,
    size_t n_ignored, size_t* consumed, size_t* headers_consumed, size_t* ignored_consumed,
    int* tag_out_flags);

/* Tag gate only (no sample copy), for device-resident chains where the filter's copy is folded
 * into its neighbours: replays :75-105,134-185 over a sorted list of syncword tag indices
 * (absolute item indices of the stream).  headers_per_tag == 0: headers[k] is the parsed_header
 * message that answers the k-th ACCEPTED tag; != 0: headers[i] answers tag i if it is accepted
 * (n_headers == n_tags).  The reference blocks the stream until that message arrives
 * (:164-185), so the outcome does not depend on message timing.  accepted[i] = 1 when tag i
 * passes.  State (in-packet span) carries across calls.  *headers_used = messages consumed. */
gr4pm_status gr4pm_syncword_detection_filter_gate(gr4pm_syncword_detection_filter* h,
                                                  const uint64_t* tag_index, size_t n_tags,
                                                  const gr4pm_header_msg* headers, size_t n_headers,
                                                  int headers_per_tag, uint8_t* accepted,
                                                  size_t* headers_used);
/* headers_per_tag mode, header decoded on the device: the message of the LAST accepted tag of a
 * call may not exist yet when its header symbols continue in the next batch.  Mark it
 * invalid_header = 2 ("pending") and deliver it with this call before the next gate(). */
gr4pm_status gr4pm_syncword_detection_filter_gate_resolve(gr4pm_syncword_detection_filter* h,
                                                          const gr4pm_header_msg* msg);

/* ------------------------------------------------------------------------------------
 * CoarseFrequencyCorrection<float> -- coarse_frequency_correction.hpp:20-99
 * Rotator<float>                   -- rotator.hpp:20-65
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_rotator gr4pm_rotator;
typedef struct {
    int mode;          /* 0: Rotator (phase_incr), 1: CoarseFrequencyCorrection (delay) */
    float phase_incr;  /* rotator.hpp:42 */
    size_t delay;      /* coarse_frequency_correction.hpp:43 */
    size_t n_channels; /* independent channels, state per channel */
    void* stream;
} gr4pm_rotator_params;
gr4pm_status gr4pm_rotator_create(const gr4pm_rotator_params* params, gr4pm_rotator** out);
void gr4pm_rotator_destroy(gr4pm_rotator* h);
gr4pm_status gr4pm_rotator_reset(gr4pm_rotator* h);
/* Rotator: tags ignored.  CFC: every tag with GR4PM_TAG_SYNCWORD re-loads the frequency
 * `delay` items later (:50-59,76-96).  tags: host, sorted by index, channel 0 only when
 * n_channels == 1; for batches use tag_channel[] to address channels. */
gr4pm_status gr4pm_rotator_process(gr4pm_rotator* h, const gr4pm_c64* in, size_t stride,
                                   size_t n, gr4pm_c64* out, const gr4pm_tag* tags,
                                   const uint32_t* tag_channel, size_t n_tags);

/* ------------------------------------------------------------------------------------
 * CostasLoop<float,float> -- costas_loop.hpp:15-149
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_costas_loop gr4pm_costas_loop;
typedef struct {
    double loop_bandwidth; /* :48 */
    int constellation;     /* constellation.hpp: 0 PILOT, 1 BPSK, 2 QPSK */
    size_t n_channels;
    void* stream;
} gr4pm_costas_loop_params;
gr4pm_status gr4pm_costas_loop_create(const gr4pm_costas_loop_params* params,
                                      gr4pm_costas_loop** out);
void gr4pm_costas_loop_destroy(gr4pm_costas_loop* h);
gr4pm_status gr4pm_costas_loop_reset(gr4pm_costas_loop* h);
void gr4pm_costas_loop_coeffs(const gr4pm_costas_loop* h, float* k1, float* k2);
/* settingsChanged() (:52-88): new constellation / loop bandwidth from tags or messages */
gr4pm_status gr4pm_costas_loop_set(gr4pm_costas_loop* h, double loop_bandwidth,
                                   int constellation);
/* same results either way.  on = 1: the kernel of _process / _process_ragged keeps to 62 VGPRs, on = 2: to 32 VGPRs
 * (what a SIMD has left beside two waves of the syncword correlator: the PLL's waves then run BESIDE a correlator
 * workgroup instead of keeping a compute unit from it) at the price of being slower by itself -- for callers that run
 * it next to a SyncwordDetection: the pipelined gr4pm_packet_receiver asks for 2.  Form 2 is only taken for calls of
 * 2^25 symbols and more: a PLL wave lives for one packet's chain however small the call, so below that the receiver
 * waits for the kernel's own speed (gr4pm_multichannel_receiver keeps form 0: measured) */
gr4pm_status gr4pm_costas_loop_set_small_footprint(gr4pm_costas_loop* h, int on);
gr4pm_status gr4pm_costas_loop_process(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t stride,
                                       size_t n, gr4pm_c64* out, const gr4pm_tag* tags,
                                       const uint32_t* tag_channel, size_t n_tags);
/* the same with a different item count per channel (n_per_channel: host [n_channels]) */
gr4pm_status gr4pm_costas_loop_process_ragged(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t stride,
                                              const size_t* n_per_channel, gr4pm_c64* out,
                                              const gr4pm_tag* tags, const uint32_t* tag_channel,
                                              size_t n_tags);

/* ------------------------------------------------------------------------------------
 * SyncwordWipeoff<c64,float> -- syncword_wipeoff.hpp:12-91
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_syncword_wipeoff gr4pm_syncword_wipeoff;
typedef struct {
    const float* syncword; /* :37 host */
    size_t n_syncword;
    void* stream;
} gr4pm_syncword_wipeoff_params;
gr4pm_status gr4pm_syncword_wipeoff_create(const gr4pm_syncword_wipeoff_params* params,
                                           gr4pm_syncword_wipeoff** out);
void gr4pm_syncword_wipeoff_destroy(gr4pm_syncword_wipeoff* h);
gr4pm_status gr4pm_syncword_wipeoff_reset(gr4pm_syncword_wipeoff* h);
/* `out` may be `in` (in place: only the syncword items are touched, no copy) */
gr4pm_status gr4pm_syncword_wipeoff_process(gr4pm_syncword_wipeoff* h, const gr4pm_c64* in,
                                            size_t n, gr4pm_c64* out, const gr4pm_tag* tags,
                                            size_t n_tags);
/* many channels, in place, one launch (per channel SyncwordWipeoff::processBulk, syncword_wipeoff.hpp:38-90):
 * h[c] is channel c's block (one syncword for all), its items are
 * buf[c * stride .. c * stride + n[c]), its tags tags[c][0 .. n_tags[c]).  Runs on h[0]'s stream. */
gr4pm_status gr4pm_syncword_wipeoff_process_channels(gr4pm_syncword_wipeoff* const* h, size_t n_channels,
                                                     gr4pm_c64* buf, size_t stride, const size_t* n,
                                                     const gr4pm_tag* const* tags, const size_t* n_tags);

/* ------------------------------------------------------------------------------------
 * InterpolatingFirFilter<TIn,TOut,float> -- interpolating_fir_filter.hpp:14-103
 *   item_kind: 0 = std::complex<float>, 1 = float
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_interp_fir gr4pm_interp_fir;
typedef struct {
    size_t interpolation; /* :39 */
    const float* taps;    /* :40 host */
    size_t n_taps;
    int item_kind;
    void* stream;
} gr4pm_interp_fir_params;
gr4pm_status gr4pm_interp_fir_create(const gr4pm_interp_fir_params* params,
                                     gr4pm_interp_fir** out);
void gr4pm_interp_fir_destroy(gr4pm_interp_fir* h);
gr4pm_status gr4pm_interp_fir_reset(gr4pm_interp_fir* h);
/* consumes n_in items, produces n_in * interpolation items (:91-99) */
gr4pm_status gr4pm_interp_fir_process(gr4pm_interp_fir* h, const void* in, size_t n_in,
                                      void* out);

/* ------------------------------------------------------------------------------------
 * SymbolFilter<TIn,TOut,float> -- symbol_filter.hpp:13-253
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_symbol_filter gr4pm_symbol_filter;
typedef struct {
    size_t samples_per_symbol; /* :55 */
    const float* taps;         /* :56 host, prototype PFB taps */
    size_t n_taps;
    size_t num_arms;           /* :58 */
    size_t delay;              /* :59 */
    int item_kind;             /* 0 complex, 1 float */
    void* stream;
} gr4pm_symbol_filter_params;
gr4pm_status gr4pm_symbol_filter_create(const gr4pm_symbol_filter_params* params,
                                        gr4pm_symbol_filter** out);
void gr4pm_symbol_filter_destroy(gr4pm_symbol_filter* h);
gr4pm_status gr4pm_symbol_filter_reset(gr4pm_symbol_filter* h);
/* tags_in: host, sorted, indices relative to in[0]; tags_out: host, indices relative to
 * out[0] (re-timed to symbols, :204-228; syncword_phase adjusted when time_est < 0,
 * :148-156).  Stops when out_cap symbols were produced (:208). */
gr4pm_status gr4pm_symbol_filter_process(gr4pm_symbol_filter* h, const void* in, size_t n_in,
                                         void* out, size_t out_cap, const gr4pm_tag* tags_in,
                                         size_t n_tags_in, gr4pm_tag* tags_out, size_t tags_cap,
                                         size_t* n_tags_out, size_t* consumed, size_t* produced);

/* Fused CoarseFrequencyCorrection -> SymbolFilter (packet_receiver.hpp:221-224 connects them
 * back to back): same results and same state updates as gr4pm_rotator_process (mode 1) followed
 * by gr4pm_symbol_filter_process with the same tags, but the rotated stream is never written
 * to memory.  Both handles must be single-channel / complex and share one stream; out_cap must
 * cover all n_in items (n_in / samples_per_symbol + n_tags_in + 2). */
gr4pm_status gr4pm_cfc_symbol_filter_process(gr4pm_rotator* cfc, gr4pm_symbol_filter* sf,
                                             const gr4pm_c64* in, size_t n_in, gr4pm_c64* out,
                                             size_t out_cap, const gr4pm_tag* tags_in, size_t n_tags_in,
                                             gr4pm_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                                             size_t* consumed, size_t* produced);
/* The same call in two halves, for callers that pipeline them (the native PacketReceiver runs
 * them as two stages): _plan replays the CFC's tag handling and computes the phasor checkpoints
 * of the call on the CFC's stream (the serial part), _run is the filter itself on the
 * SymbolFilter's stream.  The caller orders them (a _run after its _plan has completed; plans
 * and runs each in stream order).  GR4PM_CFC_PLANS plans exist (a ring): a plan stays valid until
 * GR4PM_CFC_PLANS - 1 further plans have been made, so that many calls may sit between the two
 * halves.  `plan` is the value _plan returned.
 * Round 6: where chains are long (>= 2^17 items between two syncword_freq events) and the process has at least eight
 * hardware queues (GPU_MAX_HW_QUEUES), _plan leaves the chains running on streams of the CFC handle's own when it
 * returns -- those of consecutive plans side by side -- and _run makes the SymbolFilter's stream wait for them
 * (events); the caller's ordering rule is unchanged, what it gains is that the stage which makes the plans does not
 * wait for a chain.  GR4PM_ROT_SERIAL=1 (read when the handle is created): never. */
#define GR4PM_CFC_PLANS 12
gr4pm_status gr4pm_cfc_symbol_filter_plan(gr4pm_rotator* cfc, size_t n_in, const gr4pm_tag* tags_in,
                                          size_t n_tags_in, int* plan);
/* many channels: ONE CoarseFrequencyCorrection handle with n_channels channels plans all of them
 * in one launch (tags of all channels, tag_channel[i] = channel of tags_in[i], indices relative
 * to the channel's first item); every channel's own SymbolFilter then runs with
 * _run_channel(plan, channel, ...), `in` pointing at that channel's items */
gr4pm_status gr4pm_cfc_symbol_filter_plan_channels(gr4pm_rotator* cfc, size_t n_in, const gr4pm_tag* tags_in,
                                                   const uint32_t* tag_channel, size_t n_tags_in, int* plan);
gr4pm_status gr4pm_cfc_symbol_filter_run_channel(gr4pm_rotator* cfc, int plan, size_t channel,
                                                 gr4pm_symbol_filter* sf, const gr4pm_c64* in, size_t n_in,
                                                 gr4pm_c64* out, size_t out_cap, const gr4pm_tag* tags_in,
                                                 size_t n_tags_in, gr4pm_tag* tags_out, size_t tags_cap,
                                                 size_t* n_tags_out, size_t* consumed, size_t* produced);
/* ... or ALL channels in ONE launch of the filter kernel (64 channels of 2^22 items are 64 small launches
 * otherwise; per channel it is SymbolFilter::processBulk, symbol_filter.hpp:112-252, fed by
 * CoarseFrequencyCorrection::processBulk, coarse_frequency_correction.hpp:67-98): sf[c] is channel c's SymbolFilter (one design for all; its tag-driven state is replayed on the
 * host exactly as by _run_channel), in: [n_channels][in_stride], out: [n_channels][out_stride] (out_stride >=
 * n_in / samples_per_symbol + tags + 2), tags_in[c] / n_tags_in[c] and tags_out[c] (tags_cap each) /
 * n_tags_out[c] / produced[c] per channel.  Runs on sf[0]'s stream.
 * Two-piece input (n_head > 0): the first n_head items of channel c's call are head[c * head_stride + i], item
 * i >= n_head is in[c * in_stride + i - n_head] -- SyncwordDetection's delayed stream read in place: the last
 * 2 * time_threshold + 1 items of the batch before, then this batch's own input (no delayed copy, 16 B/sample). */
gr4pm_status gr4pm_cfc_symbol_filter_run_channels(gr4pm_rotator* cfc, int plan, gr4pm_symbol_filter* const* sf,
                                                  size_t n_channels, const gr4pm_c64* in, size_t in_stride,
                                                  size_t n_in, gr4pm_c64* out, size_t out_stride,
                                                  const gr4pm_tag* const* tags_in, const size_t* n_tags_in,
                                                  gr4pm_tag* const* tags_out, size_t tags_cap, size_t* n_tags_out,
                                                  size_t* produced, const gr4pm_c64* head, size_t head_stride,
                                                  size_t n_head);
gr4pm_status gr4pm_cfc_symbol_filter_run(gr4pm_rotator* cfc, int plan, gr4pm_symbol_filter* sf,
                                         const gr4pm_c64* in, size_t n_in, gr4pm_c64* out,
                                         size_t out_cap, const gr4pm_tag* tags_in, size_t n_tags_in,
                                         gr4pm_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                                         size_t* consumed, size_t* produced);

/* ------------------------------------------------------------------------------------
 * PfbArbResampler<c64,c64,float,TRate> -- pfb_arb_resampler.hpp:23-183
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_pfb_arb_resampler gr4pm_pfb_arb_resampler;
typedef struct {
    double rate;        /* :63 */
    int rate_is_double; /* TRate: 0 float (default), 1 double */
    const float* taps;  /* :64 host; NULL = the built-in default prototype */
    size_t n_taps;
    size_t filter_size; /* :65 */
    void* stream;
} gr4pm_pfb_arb_resampler_params;
gr4pm_status gr4pm_pfb_arb_resampler_create(const gr4pm_pfb_arb_resampler_params* params,
                                            gr4pm_pfb_arb_resampler** out);
void gr4pm_pfb_arb_resampler_destroy(gr4pm_pfb_arb_resampler* h);
gr4pm_status gr4pm_pfb_arb_resampler_reset(gr4pm_pfb_arb_resampler* h);
gr4pm_status gr4pm_pfb_arb_resampler_process(gr4pm_pfb_arb_resampler* h, const gr4pm_c64* in,
                                             size_t n_in, gr4pm_c64* out, size_t out_cap,
                                             size_t* consumed, size_t* produced);

/* ====================================================================================
 * Symbol-rate control blocks behind SyncwordWipeoff: the immediate
 * consumers of the path's tags, closing the chain to soft bits on the device.
 * ================================================================================== */

/* Tags of this part of the chain (payload_metadata_insert.hpp:44-51).  Keys that name a
 * setting of a downstream block ("constellation", "loop_bandwidth") change that setting from
 * the tagged item on, like the reference's tag-driven settings. */
typedef struct {
    uint64_t index;           /* item index the tag is attached to */
    int32_t kind;             /* GR4PM_PKT_* */
    int32_t constellation;    /* "constellation": 0 PILOT, 1 BPSK, 2 QPSK; < 0: key absent */
    double loop_bandwidth;    /* "loop_bandwidth"; < 0: key absent */
    uint64_t packet_length;   /* GR4PM_PKT_PAYLOAD: "packet_length" of the parsed header */
    uint64_t payload_symbols; /* GR4PM_PKT_PAYLOAD: "payload_symbols" */
    uint64_t payload_bits;    /* GR4PM_PKT_PAYLOAD: "payload_bits" */
    gr4pm_tag syncword;       /* GR4PM_PKT_SYNCWORD: the syncword_* keys travelling along */
} gr4pm_packet_tag;
#define GR4PM_PKT_SYNCWORD 1     /* start of the (wiped-off) syncword, :104-112 */
#define GR4PM_PKT_HEADER_START 2 /* "header_start", :186-194 */
#define GR4PM_PKT_PAYLOAD 3      /* parsed header + payload sizes, :222-234 */

/* ------------------------------------------------------------------------------------
 * PayloadMetadataInsert<c64> -- payload_metadata_insert.hpp:12-324
 * One call == the processBulk() calls (:77-307) the runtime would make over n_in items, cut at
 * the syncword tags, with `headers` = the parsed_header messages pending, oldest first.  The
 * packet symbols are gathered device-to-device; everything between packets is dropped.
 * Where the reference returns to wait for a header (:243-247) the call stops: *consumed < n_in,
 * and the caller presents the rest again when the message is there.
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_payload_metadata_insert gr4pm_payload_metadata_insert;
typedef struct {
    size_t syncword_size;                  /* :60 */
    size_t header_size;                    /* :61 */
    double syncword_costas_loop_bandwidth; /* :62 */
    double header_costas_loop_bandwidth;   /* :63 */
    double payload_costas_loop_bandwidth;  /* :64 */
    void* stream;
} gr4pm_payload_metadata_insert_params;
gr4pm_status gr4pm_payload_metadata_insert_create(const gr4pm_payload_metadata_insert_params* params,
                                                  gr4pm_payload_metadata_insert** out);
void gr4pm_payload_metadata_insert_destroy(gr4pm_payload_metadata_insert* h);
gr4pm_status gr4pm_payload_metadata_insert_reset(gr4pm_payload_metadata_insert* h); /* start(), :71-75 */
/* tags_in: host, sorted, index relative to in[0]; only GR4PM_TAG_SYNCWORD tags matter.
 * headers_per_tag == 0: `headers` is the message queue, oldest first, one message per packet the
 * block opens.  != 0 (n_headers == n_tags_in): headers[i] answers tags_in[i] if that syncword
 * opens a packet, for callers that know every detection's header up front (same convention as
 * gr4pm_syncword_detection_filter_gate); the message of a packet still open at the end of the
 * call is kept.  tags_out: host, index relative to out[0].  *headers_used = messages consumed.
 * *ignored_syncwords: syncwords seen inside a packet (:126-147, the ignored_syncword messages). */
gr4pm_status gr4pm_payload_metadata_insert_process(
    gr4pm_payload_metadata_insert* h, const gr4pm_c64* in, size_t n_in, gr4pm_c64* out, size_t out_cap,
    const gr4pm_tag* tags_in, size_t n_tags_in, const gr4pm_header_msg* headers, size_t n_headers,
    int headers_per_tag, gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* consumed,
    size_t* produced, size_t* headers_used, size_t* ignored_syncwords);

/* headers_per_tag mode: the message of a packet that was opened with a pending header
 * (invalid_header = 2) in an earlier call, delivered before the call that reaches its payload. */
gr4pm_status gr4pm_payload_metadata_insert_resolve(gr4pm_payload_metadata_insert* h,
                                                   const gr4pm_header_msg* msg);

/* CostasLoop fed by PayloadMetadataInsert (packet_receiver.hpp wiring): "constellation" and
 * "loop_bandwidth" keys re-run settingsChanged() (costas_loop.hpp:52-88) from the tagged item
 * on, GR4PM_PKT_SYNCWORD tags with the syncword_* keys set the phase (:101-106).  The new
 * settings stay in the handle.  Single channel. */
gr4pm_status gr4pm_costas_loop_process_packets(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t n,
                                               gr4pm_c64* out, const gr4pm_packet_tag* tags,
                                               size_t n_tags);

/* ------------------------------------------------------------------------------------
 * SyncwordRemove<c64> -- syncword_remove.hpp:11-112
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_syncword_remove gr4pm_syncword_remove;
typedef struct {
    size_t syncword_size; /* :34 */
    void* stream;
} gr4pm_syncword_remove_params;
gr4pm_status gr4pm_syncword_remove_create(const gr4pm_syncword_remove_params* params,
                                          gr4pm_syncword_remove** out);
void gr4pm_syncword_remove_destroy(gr4pm_syncword_remove* h);
gr4pm_status gr4pm_syncword_remove_reset(gr4pm_syncword_remove* h);
/* out holds n - (dropped syncword items) <= n items; GR4PM_PKT_SYNCWORD tags start a syncword
 * and are swallowed (:51-58), the others pass, re-indexed (:59-62). */
gr4pm_status gr4pm_syncword_remove_process(gr4pm_syncword_remove* h, const gr4pm_c64* in, size_t n,
                                           gr4pm_c64* out, const gr4pm_packet_tag* tags_in,
                                           size_t n_tags_in, gr4pm_packet_tag* tags_out, size_t tags_cap,
                                           size_t* n_tags_out, size_t* produced);

/* ------------------------------------------------------------------------------------
 * ConstellationLLRDecoder<float> -- constellation_llr_decoder.hpp:13-142
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_constellation_llr_decoder gr4pm_constellation_llr_decoder;
typedef struct {
    float noise_sigma; /* :45 */
    int constellation; /* :46-47: 1 BPSK, 2 QPSK (0 PILOT is rejected like :72-74) */
    void* stream;
} gr4pm_constellation_llr_decoder_params;
gr4pm_status gr4pm_constellation_llr_decoder_create(const gr4pm_constellation_llr_decoder_params* params,
                                                    gr4pm_constellation_llr_decoder** out);
void gr4pm_constellation_llr_decoder_destroy(gr4pm_constellation_llr_decoder* h);
/* out: device floats, out_cap >= 2 n is always enough.  "constellation" keys switch the mapping
 * from the tagged item on; tags leave at the LLR index of their item (:93-99). */
gr4pm_status gr4pm_constellation_llr_decoder_process(gr4pm_constellation_llr_decoder* h, const gr4pm_c64* in,
                                                     size_t n, float* out, size_t out_cap,
                                                     const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                                                     gr4pm_packet_tag* tags_out, size_t tags_cap,
                                                     size_t* n_tags_out, size_t* produced);

/* ====================================================================================
 * Header decode loop (packet_receiver.hpp:131-139): descrambler ->
 * header/payload split -> header FEC decoder -> header parser.  It produces the parsed_header
 * messages SyncwordDetectionFilter and PayloadMetadataInsert wait for.
 * ================================================================================== */

/* ------------------------------------------------------------------------------------
 * AdditiveScrambler<float | uint8_t> -- additive_scrambler.hpp:24-100
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_additive_scrambler gr4pm_additive_scrambler;
typedef struct {
    uint64_t mask;   /* :61 */
    uint64_t seed;   /* :62 */
    uint64_t length; /* :63 */
    uint64_t count;  /* :64, 0 = never reset by count */
    int item_kind;   /* 1: float soft symbols (sign flip), 2: uint8_t hard symbols (XOR) */
    void* stream;
} gr4pm_additive_scrambler_params;
gr4pm_status gr4pm_additive_scrambler_create(const gr4pm_additive_scrambler_params* params,
                                             gr4pm_additive_scrambler** out);
void gr4pm_additive_scrambler_destroy(gr4pm_additive_scrambler* h);
gr4pm_status gr4pm_additive_scrambler_reset(gr4pm_additive_scrambler* h); /* start(), :68 */
/* reset_index: host, sorted item indices that carry the reset_tag_key (:78-83) */
gr4pm_status gr4pm_additive_scrambler_process(gr4pm_additive_scrambler* h, const void* in, size_t n,
                                              void* out, const uint64_t* reset_index, size_t n_resets);

/* ------------------------------------------------------------------------------------
 * HeaderPayloadSplit<float> -- header_payload_split.hpp:9-147
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_header_payload_split gr4pm_header_payload_split;
typedef struct {
    size_t header_size; /* :33 */
    void* stream;
} gr4pm_header_payload_split_params;
gr4pm_status gr4pm_header_payload_split_create(const gr4pm_header_payload_split_params* params,
                                               gr4pm_header_payload_split** out);
void gr4pm_header_payload_split_destroy(gr4pm_header_payload_split* h);
gr4pm_status gr4pm_header_payload_split_reset(gr4pm_header_payload_split* h); /* start(), :41-45 */
/* GR4PM_PKT_PAYLOAD tags carry "payload_bits" (:70-82).  header / payload: device, each with
 * room for n items.  Tags leave on the output their item goes to (:83-87).  A payload tag in
 * the wrong place is the reference's exception (:75-78): GR4PM_ERR_INVALID. */
gr4pm_status gr4pm_header_payload_split_process(gr4pm_header_payload_split* h, const float* in, size_t n,
                                                float* header, size_t* n_header, float* payload,
                                                size_t* n_payload, const gr4pm_packet_tag* tags_in,
                                                size_t n_tags_in, gr4pm_packet_tag* header_tags,
                                                size_t* n_header_tags, gr4pm_packet_tag* payload_tags,
                                                size_t* n_payload_tags, size_t tags_cap);
/* HeaderPayloadSplit<std::complex<float>>: the split of the symbol tap, packet_receiver.hpp:159-162 (header_size 128,
 * the GR4PM_PKT_PAYLOAD tags then carry "payload_symbols" in payload_bits).  Same state machine, complex items. */
gr4pm_status gr4pm_header_payload_split_process_c64(gr4pm_header_payload_split* h, const gr4pm_c64* in, size_t n,
                                                    gr4pm_c64* header, size_t* n_header, gr4pm_c64* payload,
                                                    size_t* n_payload, const gr4pm_packet_tag* tags_in,
                                                    size_t n_tags_in, gr4pm_packet_tag* header_tags,
                                                    size_t* n_header_tags, gr4pm_packet_tag* payload_tags,
                                                    size_t* n_payload_tags, size_t tags_cap);

/* ------------------------------------------------------------------------------------
 * HeaderFecDecoder -- header_fec_decoder.hpp:13-359: 256 LLRs (rate-1/2 repetition of a
 * (128, 32) LDPC codeword) -> 4 header bytes, or "invalid_header".
 * The reference hands the LDPC decoding to its Rust dependency ldpc-toolbox
 * (ldpc_toolbox_decoder_ctor_alist_string(alist, "HLAminstari8", ""), :276, and
 * ldpc_toolbox_decoder_decode_f32, :315-321, 25 iterations).  `alist` is that same string.
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_header_fec_decoder gr4pm_header_fec_decoder;
typedef struct {
    const char* alist;       /* parity-check matrix, alist text (:31-258) */
    uint32_t max_iterations; /* :314, 25 */
    void* stream;
    int arithmetic;          /* message arithmetic of the horizontal-layered A-Min* decoder ("HLAminstari8": HL =
                                horizontal-layered schedule, Aminstar = the A-Min* check-node rule, i8 = 8-bit messages):
                                0 (default) float32 messages; 1 the same schedule and rule with 8-bit messages -- LLRs in
                                eighths, channel LLRs and messages saturating at +-127.  The crate is not part of the
                                reference tree, so neither form is pinned against it bit for bit; form 1 exists to show
                                what 8-bit messages change (tests/test_gpu_parity.py: frame-error rates side by side). */
} gr4pm_header_fec_decoder_params;
gr4pm_status gr4pm_header_fec_decoder_create(const gr4pm_header_fec_decoder_params* params,
                                             gr4pm_header_fec_decoder** out);
void gr4pm_header_fec_decoder_destroy(gr4pm_header_fec_decoder* h);
/* llrs: device, 256 per codeword.  headers: host, 4 bytes per codeword; invalid: host, one
 * flag per codeword (the "invalid_header" tag of :322-326). */
gr4pm_status gr4pm_header_fec_decoder_process(gr4pm_header_fec_decoder* h, const float* llrs,
                                              size_t n_codewords, uint8_t* headers, uint8_t* invalid);

/* HeaderParser -- header_parser.hpp:46-95 (host): 4 bytes (+ the decoder's verdict) -> the
 * parsed_header message; packet_type: 0 USER_DATA, 1 IDLE, -1 when invalid. */
void gr4pm_header_parse(const uint8_t* headers, const uint8_t* invalid, size_t n, gr4pm_header_msg* msgs,
                        int32_t* packet_type);

/* ====================================================================================
 * Payload tail (packet_receiver.hpp:140-147): BinarySlicer<true> -> PackBits<> -> CrcCheck<>.
 * ================================================================================== */

/* BinarySlicer<invert, float, uint8_t> -- binary_slicer.hpp:10-35: out = invert ? in < 0 : in > 0 */
gr4pm_status gr4pm_binary_slicer_process(const float* in, size_t n, uint8_t* out, int invert, void* stream);
/* PackBits<MSB|LSB, uint8_t, uint8_t> -- pack_bits.hpp: n_out outputs, each joins
 * inputs_per_output inputs of bits_per_input bits (inputs_per_output * bits_per_input <= 8) */
gr4pm_status gr4pm_pack_bits_process(const uint8_t* in, size_t n_out, uint8_t* out, size_t inputs_per_output,
                                     unsigned bits_per_input, int msb_first, void* stream);
/* both at once for the receiver's wiring (invert = true, 8 x 1 bit, MSB first): n_out bytes from
 * 8 n_out soft bits */
gr4pm_status gr4pm_slice_pack_process(const float* in, size_t n_out, uint8_t* out, void* stream);

/* CrcCheck<uint64_t> -- crc_check.hpp:22-239 with Crc<uint64_t> -- crc.hpp:31-156 */
typedef struct gr4pm_crc_check gr4pm_crc_check;
typedef struct {
    unsigned num_bits;          /* crc_check.hpp:61, multiple of 8 */
    uint64_t poly;              /* :62 */
    uint64_t initial_value;     /* :63 */
    uint64_t final_xor;         /* :64 */
    int input_reflected;        /* :65 */
    int result_reflected;       /* :66 */
    int swap_endianness;        /* :67 */
    int discard_crc;            /* :68 */
    uint64_t skip_header_bytes; /* :69 */
    void* stream;
} gr4pm_crc_check_params;
gr4pm_status gr4pm_crc_check_create(const gr4pm_crc_check_params* params, gr4pm_crc_check** out);
void gr4pm_crc_check_destroy(gr4pm_crc_check* h);
/* host helper: Crc::compute over host bytes (crc.hpp:119-156) */
uint64_t gr4pm_crc_check_compute(const gr4pm_crc_check* h, const uint8_t* data, size_t n);
/* in: device bytes; packet i occupies [packet_offset[i], packet_offset[i] + packet_len[i]) ("packet_len"
 * tags).  Packets whose CRC matches are copied to `out` (device) back to back, without the CRC
 * when discard_crc; out_len[i] (host) = bytes written for packet i, 0 = dropped (:152-208). */
gr4pm_status gr4pm_crc_check_process(gr4pm_crc_check* h, const uint8_t* in, const uint64_t* packet_offset,
                                     const uint64_t* packet_len, size_t n_packets, uint8_t* out,
                                     uint64_t* out_len, size_t* n_out_bytes);

/* ====================================================================================
 * PacketReceiver -- packet_receiver.hpp:34-147,191-247: the composition itself, native.
 * The reference wires the blocks into a flowgraph and runs them with the multi-threaded
 * scheduler (one worker per block).  This object owns the blocks of the chain up to the Costas
 * loop (soft_bits: up to the LLR decoder; decode_headers: the whole receiver), its HIP streams and
 * worker threads; batches are submitted and collected, up to four in flight:
 *   stage 0 (caller's thread, inside submit): SyncwordDetection (+ look-ahead of the next batch)
 *   stage 1: SyncwordDetectionFilter gate, CoarseFrequencyCorrection + SymbolFilter, SyncwordWipeoff
 *   stage 2: CostasLoop [PayloadMetadataInsert, tag-driven CostasLoop, SyncwordRemove, LLR decoder]
 *   stage 3 (decode_headers): descrambler, HeaderPayloadSplit, HeaderFecDecoder, HeaderParser,
 *            BinarySlicer, PackBits, CrcCheck
 * The parsed_header feedback is a constant packet_length per submit (0 = every header invalid),
 * or with decode_headers the header decode loop on the device.
 * ================================================================================== */
/* The symbol PDU tap of packet_receiver.hpp:159-189 (`zmq_output`): SyncwordRemove's output ->
 * HeaderPayloadSplit<c64>{ header_size 128, payload_length_key "payload_symbols" } -> TaggedStreamToPdu ->
 * ZmqPduPubSink on TCP port 5000 (headers) / :5001 (payloads).  The split (header_payload_split.hpp:46-135) is done by
 * the receiver: every batch's post-SyncwordRemove symbols are cut into header PDUs (128 symbols) and payload PDUs
 * ("payload_symbols" symbols behind a parsed header); a PDU that crosses a batch boundary appears as a piece with
 * `last == 0` and continues in the next batch with `first == 0`.  With a callback registered, collect() copies the
 * batch's symbols to the host once and calls it once per COMPLETE PDU, in stream order, from the collecting thread:
 * the place where a caller publishes them (gr4pm_packet_receiver_publish_symbol_pdus below does: ZMTP 3.0, no libzmq). */
typedef struct {
    uint64_t offset;  /* first symbol of the piece inside pdu_symbols */
    uint64_t length;  /* symbols of the piece in this batch */
    int32_t kind;     /* 0 header, 1 payload */
    int32_t first;    /* the PDU starts with this piece */
    int32_t last;     /* the PDU ends with this piece */
    int32_t pad;
} gr4pm_symbol_pdu;
typedef void (*gr4pm_symbol_pdu_fn)(void* user, int kind, const gr4pm_c64* symbols_host, size_t n_symbols);
typedef struct gr4pm_packet_receiver gr4pm_packet_receiver;
typedef struct {
    size_t samples_per_symbol; /* packet_receiver.hpp:49 */
    int syncword_freq_bins;    /* :54 */
    float syncword_threshold;  /* :55 */
    int costas_constellation;  /* CostasLoop setting when soft_bits == 0 (0 PILOT 1 BPSK 2 QPSK) */
    size_t max_items;          /* largest batch */
    size_t tags_cap;           /* most detections per batch */
    int pipelined;             /* 0: submit() runs the three stages itself */
    int soft_bits;             /* continue to the LLR decoder (:123-131) */
    int decode_headers;        /* (needs soft_bits) close the header feedback loop on the device and run the
                                  payload tail: descrambler, HeaderPayloadSplit, HeaderFecDecoder, HeaderParser
                                  (:131-139) answer SyncwordDetectionFilter and PayloadMetadataInsert instead of
                                  packet_length; BinarySlicer, PackBits, CrcCheck (:140-147) deliver the packets.
                                  Two passes per batch, see HISTORY.md 5c. */
    const char* header_alist;  /* decode_headers: the header code, header_fec_decoder.hpp:31-258 */
    int packets_only;          /* (decode_headers) round 6: what benchmarks/benchmark_packet_receiver.cpp measures -- IQ in,
                                  CRC-checked packets out -- with the stream between the Costas loop and the packer never
                                  written to memory: SyncwordRemove, the LLR decoder, the descrambler, HeaderPayloadSplit,
                                  the slicer and the packer advance their state on the host as always, and ONE kernel
                                  (k_tail_fused) reads every Costas-loop output symbol once, writes header LLRs for the
                                  header decoder and packed payload bytes for CrcCheck.  The same packets, header messages
                                  and tags as the full form, bit for bit; result.llr, .payload_llr and .pdu_symbols are NULL
                                  (their counts and tags are still reported), out_llr may be NULL, no symbol PDU tap. */
} gr4pm_packet_receiver_params;
typedef struct {
    size_t consumed;                    /* items of the batch SyncwordDetection consumed */
    const gr4pm_c64* symbols;           /* == out_symbols of the submit */
    size_t n_symbols;
    const float* llr;                   /* == out_llr of the submit (soft_bits) */
    size_t n_llr;
    const gr4pm_tag* detector_tags;     /* host; valid until the next collect() */
    size_t n_detector_tags;
    const uint8_t* accepted;            /* per detector tag: passed the filter */
    const gr4pm_tag* tags;              /* symbol-rate tags of the accepted detections */
    size_t n_tags;
    const gr4pm_packet_tag* packet_tags; /* soft_bits: PayloadMetadataInsert's tags (symbol index) */
    size_t n_packet_tags;
    const gr4pm_packet_tag* llr_tags;   /* soft_bits: tags at LLR positions */
    size_t n_llr_tags;
    size_t ignored_syncwords;
    /* decode_headers */
    const gr4pm_header_msg* header_messages; /* the headers the chain decoded in this batch, in order */
    const int32_t* packet_type;
    size_t n_header_messages;
    size_t header_mismatches;           /* of those, how many differ from the message pass A had supplied */
    const float* payload_llr;           /* device: descrambled payload LLRs of this batch */
    size_t n_payload_llr;
    const gr4pm_packet_tag* payload_tags;
    size_t n_payload_tags;
    const uint8_t* packets;             /* == out_packets: bytes of the packets whose CRC-32 matches */
    size_t n_packet_bytes;
    const uint64_t* packet_lengths;     /* per finished packet: bytes delivered, 0 = CRC failure */
    size_t n_packets;
    /* soft_bits: the symbol PDU tap (below) */
    const gr4pm_c64* pdu_symbols;       /* device: SyncwordRemove's output of this batch (header + payload symbols) */
    size_t n_pdu_symbols;
    const gr4pm_symbol_pdu* symbol_pdus; /* host: the pieces of header / payload PDUs inside pdu_symbols, in order */
    size_t n_symbol_pdus;
    size_t symbol_pdu_resyncs;          /* "payload_symbols" tags that arrived where the tap's HeaderPayloadSplit did not
                                           expect one (the reference block throws, header_payload_split.hpp:75-78; here
                                           the tap starts over at the tag and the batch -- LLRs, packets -- is not failed
                                           because of its optional side output) */
} gr4pm_packet_receiver_result;
gr4pm_status gr4pm_packet_receiver_create(const gr4pm_packet_receiver_params* params,
                                          gr4pm_packet_receiver** out);
void gr4pm_packet_receiver_destroy(gr4pm_packet_receiver* h);
/* in: device, n_in items.  delayed: NULL, or the address of item -(2 time_threshold + 1) of the
 * stream relative to in[0] when `in` is a window of a device ring (the chain then reads the
 * delayed stream in place).  next_in/next_n: the following batch (look-ahead) or NULL.
 * out_symbols (device, out_cap >= n_in / samples_per_symbol + tags + 2) and out_llr (device,
 * soft_bits, 2 floats per symbol) and out_packets (device, decode_headers, n_in / 16 bytes is
 * always enough) stay the caller's; the result points at them.  The receiver works on streams of its
 * own, which are not ordered against the caller's: `in` (and `next_in`, and what announce names) must be
 * complete when the call is made. */
gr4pm_status gr4pm_packet_receiver_submit(gr4pm_packet_receiver* h, const gr4pm_c64* in, size_t n_in,
                                          const gr4pm_c64* delayed, const gr4pm_c64* next_in,
                                          size_t next_n, uint64_t packet_length, gr4pm_c64* out_symbols,
                                          size_t out_cap, float* out_llr, size_t llr_cap, uint8_t* out_packets,
                                          size_t packets_cap);
/* names the input of a later submit (after the ones already announced): forwarded to the
 * detector's look-ahead, gr4pm_syncword_detection_announce; next_in of _submit is the one-call
 * form (gr4pm_syncword_detection_hint_next) */
gr4pm_status gr4pm_packet_receiver_announce(gr4pm_packet_receiver* h, const gr4pm_c64* in, size_t n_in);
/* waits for the oldest batch; returns its status (the error text of a failed stage included) */
gr4pm_status gr4pm_packet_receiver_collect(gr4pm_packet_receiver* h, gr4pm_packet_receiver_result* result);
size_t gr4pm_packet_receiver_inflight(const gr4pm_packet_receiver* h);
/* soft_bits receivers: fn == NULL removes the callback */
gr4pm_status gr4pm_packet_receiver_set_symbol_pdu_callback(gr4pm_packet_receiver* h, gr4pm_symbol_pdu_fn fn, void* user);

/* ------------------------------------------------------------------------------------
 * ZmqPduPubSink<T> -- zmq_pdu_pub_sink.hpp:11-44: a ZeroMQ PUB socket, one message per PDU holding its raw items.
 * The library speaks ZMTP 3.0 (NULL mechanism) itself -- greeting, READY, the subscriptions a SUB peer sends, one
 * single-frame message per send, PUB drop semantics (no subscriber / a peer whose queue holds 1000 messages) -- so a
 * zmq.SUB socket (scripts/plot_symbols.py:10-17) connects to it as to the reference's; libzmq is not needed.
 * Host only: works without a HIP device.
 * ---------------------------------------------------------------------------------- */
typedef struct gr4pm_zmq_pub gr4pm_zmq_pub;
/* endpoint (:26): tcp://HOST:PORT as ZeroMQ spells it -- HOST an IPv4 address or a star (every interface), PORT a
 * number, or a star / 0 for an ephemeral port (gr4pm_zmq_pub_port); the reference's default is port 5555 on every interface.
 * = start(): socket.bind(endpoint) (:29) */
gr4pm_status gr4pm_zmq_pub_create(const char* endpoint, gr4pm_zmq_pub** out);
void gr4pm_zmq_pub_destroy(gr4pm_zmq_pub* h); /* queued messages get 200 ms to leave */
/* processOne() (:31-41): data = a PDU's items (host), one message; never blocks on a peer */
gr4pm_status gr4pm_zmq_pub_send(gr4pm_zmq_pub* h, const void* data, size_t bytes);
int gr4pm_zmq_pub_port(const gr4pm_zmq_pub* h);           /* the bound TCP port */
size_t gr4pm_zmq_pub_subscribers(const gr4pm_zmq_pub* h); /* connected peers holding a subscription */
uint64_t gr4pm_zmq_pub_dropped(const gr4pm_zmq_pub* h);   /* messages a subscribed peer did not get: its queue was full */
/* packet_receiver.hpp:159-189 (`zmq_output`) in one call: header PDUs to header_endpoint (port 5000 on every interface, :166),
 * payload PDUs to payload_endpoint (port 5001, :168), published by collect() (it takes the place of a callback set
 * with gr4pm_packet_receiver_set_symbol_pdu_callback, and is removed by one).  NULL, NULL stops publishing.  `ports`
 * (may be NULL): the two bound TCP ports. */
gr4pm_status gr4pm_packet_receiver_publish_symbol_pdus(gr4pm_packet_receiver* h, const char* header_endpoint,
                                                       const char* payload_endpoint, int ports[2]);

/* gr4pm_multichannel_receiver: BASELINE configs[2] -- n_channels independent receive chains on one
 * GPU (front-end mode of gr4pm_packet_receiver, channel by channel).  One batched
 * SyncwordDetection handle for all channels, then every channel's own SyncwordDetectionFilter /
 * CoarseFrequencyCorrection / SymbolFilter / SyncwordWipeoff / CostasLoop, spread over `workers`
 * threads with a stream each.  process() is synchronous; submit() / collect() run the same chain as a
 * pipeline: submit returns when the detector has consumed the batch, the stages behind it (tag gates + CFC
 * plan | symbol filters + wipe-off | Costas loop) work on up to GR4PM_MC_SLOTS batches at a time in their own
 * threads, collect() waits for the oldest one (results in submission order, bit-identical to process()). */
#define GR4PM_MC_SLOTS 4
typedef struct gr4pm_multichannel_receiver gr4pm_multichannel_receiver;
typedef struct {
    size_t n_channels;
    size_t samples_per_symbol;
    int syncword_freq_bins;
    float syncword_threshold;
    int costas_constellation; /* 0 PILOT, 1 BPSK, 2 QPSK */
    size_t max_items;         /* per channel and call */
    size_t tags_cap;          /* per channel and call */
    int workers;
} gr4pm_multichannel_receiver_params;
gr4pm_status gr4pm_multichannel_receiver_create(const gr4pm_multichannel_receiver_params* params,
                                                gr4pm_multichannel_receiver** out);
void gr4pm_multichannel_receiver_destroy(gr4pm_multichannel_receiver* h);
/* the detector's look-ahead (gr4pm_syncword_detection_announce) */
gr4pm_status gr4pm_multichannel_receiver_announce(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                  size_t in_stride, size_t n_in);
/* in: device [n_channels][in_stride], n_in items per channel.  packet_length: the parsed_header
 * answer for every packet (0: "invalid_header").  out_symbols: device [n_channels][out_stride]
 * (out_stride >= n_in / samples_per_symbol + tags + 2); n_symbols, n_tags, n_detector_tags: host
 * [n_channels]; tags, detector_tags: host [n_channels][tags_cap] (may be NULL). */
gr4pm_status gr4pm_multichannel_receiver_process(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                 size_t in_stride, size_t n_in, uint64_t packet_length,
                                                 gr4pm_c64* out_symbols, size_t out_stride, size_t* consumed,
                                                 size_t* n_symbols, gr4pm_tag* tags, size_t* n_tags,
                                                 gr4pm_tag* detector_tags, size_t* n_detector_tags);

/* the pipelined form: out_symbols must stay valid until the batch has been collected.  The receiver works on
 * streams of its own, which are not ordered against the caller's: `in` must be complete (its producer finished,
 * or waited for) when submit / announce is called. */
gr4pm_status gr4pm_multichannel_receiver_submit(gr4pm_multichannel_receiver* h, const gr4pm_c64* in,
                                                size_t in_stride, size_t n_in, uint64_t packet_length,
                                                gr4pm_c64* out_symbols, size_t out_stride, size_t* consumed);
gr4pm_status gr4pm_multichannel_receiver_collect(gr4pm_multichannel_receiver* h, size_t* consumed, size_t* n_symbols,
                                                 gr4pm_tag* tags, size_t* n_tags, gr4pm_tag* detector_tags,
                                                 size_t* n_detector_tags);
int gr4pm_multichannel_receiver_in_flight(const gr4pm_multichannel_receiver* h);
/* on: the caller keeps every submitted input valid and unchanged until its batch has been collected (a device
 * ring).  SyncwordDetection's output is its input delayed by 2 * time_threshold + 1 items
 * (syncword_detection.hpp:318-319,342).  The receiver then reads SyncwordDetection's delayed stream in place (the tail of the batch before, kept
 * by the receiver, + the batch's own input) instead of writing a delayed copy of every batch (16 B/sample).
 * Same results.  Call it before the first batch. */
gr4pm_status gr4pm_multichannel_receiver_set_input_in_place(gr4pm_multichannel_receiver* h, int on);

/* ====================================================================================
 * Burst generator pieces (packet_transmitter_pdu.hpp:131-337): with
 * AdditiveScrambler, PackBits, InterpolatingFirFilter, Rotator and PfbArbResampler above they
 * build the test signal on the device.
 * ================================================================================== */
/* Mapper<uint8_t, c64 | float> -- mapper.hpp:13-51: out[i] = map[in[i] & (map_size - 1)];
 * map_size must be a power of two (:37-41); item_kind 0: c64 map/out, 1: float. */
gr4pm_status gr4pm_mapper_process(const uint8_t* in, size_t n, void* out, const void* map_host,
                                  size_t map_size, int item_kind, void* stream);
/* BurstShaper<c64 | float, ., float> -- burst_shaper.hpp:47-126: the first leading_n items of
 * every packet are multiplied by leading[], the last trailing_n by trailing[] (short packets:
 * leading first, :98-124).  Packets: [packet_offset[i], + packet_len[i]) ("packet_len" tags),
 * whole packets per call; items outside packets are copied. */
gr4pm_status gr4pm_burst_shaper_process(const void* in, size_t n, void* out, int item_kind,
                                        const float* leading_host, size_t leading_n,
                                        const float* trailing_host, size_t trailing_n,
                                        const uint64_t* packet_offset, const uint64_t* packet_len,
                                        size_t n_packets, void* stream);

/* The library's cosf / sinf on device arrays: the local oscillator of CostasLoop (costas_loop.hpp:113-115 calls
 * std::cos(float) / std::sin(float), i.e. glibc's cosf / sinf; the kernels restate glibc's double-precision
 * algorithm and are bit-exact with it for |x| < 120).  Exposed so that the parity suite can pin it directly. */
gr4pm_status gr4pm_sincosf(const float* x, size_t n, float* sin_out, float* cos_out);

/* The phase wrap at the end of a CostasLoop iteration on device arrays -- costas_loop.hpp:141-145
 * (`if (phase >= pi) phase -= 2 pi; else if (phase < -pi) phase += 2 pi`, float).  The kernels evaluate it without
 * compares (two fused multiply-adds with the clamp modifier); exposed so that the parity suite can pin it on the
 * boundary values a signal rarely produces. */
gr4pm_status gr4pm_costas_phase_wrap(const float* x, size_t n, float* out);

/* firdes::root_raised_cosine<float> -- firdes.hpp:29-76 (host helper; out: ntaps|1 floats) */
size_t gr4pm_firdes_root_raised_cosine(double gain, double sampling_freq, double symbol_rate,
                                       double alpha, size_t ntaps, float* out);

#ifdef __cplusplus
}
#endif
#endif /* GR4PM_HIP_H */
