// common.hpp -- shared host-side plumbing of libgr4pm_hip.so (status codes, error text,
// HIP call checking, small RAII device buffer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include <vector>

#include "../../include/gr4pm_hip.h"
#include "hostlogic/tail_plan.hpp"

struct gr4pm_additive_scrambler;
struct gr4pm_header_payload_split;
struct gr4pm_syncword_remove;
struct gr4pm_constellation_llr_decoder;
struct gr4pm_costas_loop;
struct gr4pm_payload_metadata_insert;

namespace gr4pm {

void set_error(const char* fmt, ...);

#define GR4PM_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::gr4pm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                               __FILE__, __LINE__);                                          \
            return GR4PM_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

#define GR4PM_TRY(expr)                                                                      \
    do {                                                                                     \
        gr4pm_status _s = (expr);                                                            \
        if (_s != GR4PM_OK) return _s;                                                       \
    } while (0)

// the product has no CPU fallback: every create() goes through this first
gr4pm_status require_device();

// "No exceptions cross the ABI" (include/gr4pm_hip.h): every extern "C" entry is a function-try-block whose handler is
// GR4PM_ABI_CATCH, and every thread a handle starts runs its body under guarded().  The library allocates through
// std::vector / std::deque / std::thread, so std::bad_alloc and std::system_error are possible anywhere; they become
// GR4PM_ERR_NOMEM / GR4PM_ERR_INTERNAL plus gr4pm_last_error() text instead of unwinding into C (or std::terminate in
// a worker).  tools/check_abi_guards.py (run by build()) verifies that no entry point lacks the handler.
gr4pm_status exception_status(const char* where) noexcept; // call inside a catch (...) handler
#define GR4PM_ABI_CATCH                                                                      \
    catch (...) { return ::gr4pm::exception_status(__func__); }
#define GR4PM_ABI_CATCH_RET(value)                                                           \
    catch (...) { (void)::gr4pm::exception_status(__func__); return value; }
#define GR4PM_ABI_CATCH_VOID                                                                 \
    catch (...) { (void)::gr4pm::exception_status(__func__); }
// runs fn() (a stage body); an exception becomes a status, never leaves the thread
template <typename F>
inline gr4pm_status guarded(const char* where, F&& fn) noexcept
{
    try {
        fn();
        return GR4PM_OK;
    } catch (...) {
        return exception_status(where);
    }
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    // pinned staging area of upload_staged(): a copy from pageable memory would be staged (or
    // pinned on the fly) inside the runtime, under locks other host threads' HIP calls wait for
    T* stage = nullptr;
    size_t stage_n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf()
    {
        release();
        if (stage) (void)hipHostFree(stage);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    gr4pm_status alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return GR4PM_ERR_NOMEM;
        }
        n = count;
        return GR4PM_OK;
    }
    gr4pm_status zero(hipStream_t s)
    {
        GR4PM_HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), s));
        return GR4PM_OK;
    }
    gr4pm_status upload(const T* host, size_t count, hipStream_t s)
    {
        GR4PM_HIP_TRY(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
        return GR4PM_OK;
    }
    // per-call tables: through the pinned staging area, truly asynchronous.  The caller
    // synchronises the stream before the next upload_staged() of this buffer.
    // makes room for uploads of up to `count` items ahead of time (buffer rings: first use of every set)
    gr4pm_status reserve_stage(size_t count)
    {
        if (stage_n >= count) return GR4PM_OK;
        if (stage) (void)hipHostFree(stage);
        stage = nullptr;
        stage_n = 0;
        const size_t want = count * 2;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&stage), want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            stage = nullptr;
            set_error("hipHostMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
            return GR4PM_ERR_NOMEM;
        }
        stage_n = want;
        return GR4PM_OK;
    }
    gr4pm_status upload_staged(const T* host, size_t count, hipStream_t s)
    {
        if (count == 0) return GR4PM_OK;
        GR4PM_TRY(reserve_stage(count));
        memcpy(stage, host, count * sizeof(T));
        GR4PM_HIP_TRY(hipMemcpyAsync(p, stage, count * sizeof(T), hipMemcpyHostToDevice, s));
        return GR4PM_OK;
    }
};

template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf()
    {
        if (p) (void)hipHostFree(p);
    }
    gr4pm_status alloc(size_t count)
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        if (count == 0) count = 1;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipHostMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return GR4PM_ERR_NOMEM;
        }
        n = count;
        return GR4PM_OK;
    }
};

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Environment switches of the A/B and timing experiments (HISTORY.md appendix).  Every one is read ONCE per process
// through these helpers; a switch that makes a call return GR4PM_OK with WRONG OUTPUTS (kernels left out or replaced
// by stand-ins: GR4PM_TIMING_SKIP, GR4PM_SYMF_ABL, GR4PM_FAKE ...) says so on stderr the first time it is seen, so a
// variable that leaked into a production environment cannot go unnoticed.
void sd_set_coresident(struct ::gr4pm_syncword_detection* h, bool on); // syncword_detection.hip: see launch_correlate
// header_blocks.hip: BinarySlicer + PackBits over a stream in two pieces (the native receiver's payload tail)
gr4pm_status slice_pack_two(const float* a, size_t na, const float* b, size_t n_out, uint8_t* out, hipStream_t s);
// round 6, the packets_only receiver: the host halves of the blocks behind the Costas loop (state advances, tags and span
// tables as in their process() calls, no kernel) and the one kernel that does their work (header_blocks.hip, k_tail_fused)
// PayloadMetadataInsert's host half (state, tags, span table; nothing is moved) and the Costas loop reading its input stream
// through that table: the block's gather folded into the loop's loads
gr4pm_status payload_metadata_insert_plan(::gr4pm_payload_metadata_insert* h, size_t n_in, size_t out_cap, const gr4pm_tag* tags_in,
                                          size_t n_tags_in, const gr4pm_header_msg* headers, size_t n_headers, int headers_per_tag,
                                          gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* consumed,
                                          size_t* produced, size_t* headers_used, size_t* ignored_syncwords,
                                          std::vector<hostlogic::CopySpan>& spans);
gr4pm_status costas_loop_process_packets_from(::gr4pm_costas_loop* h, const gr4pm_c64* in, const hostlogic::CopySpan* spans,
                                              size_t n_spans, size_t n, gr4pm_c64* out, const gr4pm_packet_tag* tags, size_t n_tags);
gr4pm_status syncword_remove_plan(::gr4pm_syncword_remove* h, size_t n, const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                                  gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* produced,
                                  std::vector<hostlogic::CopySpan>& spans);
// (all_qpsk: every run maps a symbol to two LLRs -- what the composition of hostlogic/tail_plan.hpp assumes)
gr4pm_status llr_decoder_plan(::gr4pm_constellation_llr_decoder* h, size_t n, const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                              gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* produced, bool* all_qpsk,
                              float* scale);
gr4pm_status scrambler_plan(::gr4pm_additive_scrambler* h, size_t n, const uint64_t* reset_index, size_t n_resets,
                            std::vector<hostlogic::ScrambleRun>& runs);
gr4pm_status header_payload_split_plan(::gr4pm_header_payload_split* h, size_t n, const gr4pm_packet_tag* tags_in,
                                       size_t n_tags_in, gr4pm_packet_tag* header_tags, gr4pm_packet_tag* payload_tags,
                                       size_t tags_cap, hostlogic::HpsReplay& rp);
gr4pm_status tail_fused(::gr4pm_additive_scrambler* scr, DevBuf<hostlogic::TailSpan>& table,
                        const std::vector<hostlogic::TailSpan>& spans, const gr4pm_c64* symbols, float scale, float* header_llr,
                        uint8_t* packed, hipStream_t s);
const char* experiment_env(const char* name, bool wrong_results); // nullptr when unset
unsigned experiment_env_wg(const char* name, unsigned fallback, unsigned lo, unsigned hi); // clamped to [lo, hi]

// The last thing most process() calls do is wait for their stream.  A native caller that chains
// several handles on ONE stream and reads nothing back in between (csrc/packet_receiver.hip) turns
// that wait off for its thread (gr4pm_set_deferred_sync) and synchronises once per stage.
bool deferred_sync();
void set_deferred_sync(bool on);
// switches the calling thread to deferred synchronisation for a scope and restores the previous state on
// EVERY exit path (an early error return used to leave it on: later calls of that thread then returned
// before their kernels had finished)
struct DeferredSyncScope {
    bool was;
    DeferredSyncScope() : was(deferred_sync()) { set_deferred_sync(true); }
    ~DeferredSyncScope() { set_deferred_sync(was); }
    DeferredSyncScope(const DeferredSyncScope&) = delete;
    DeferredSyncScope& operator=(const DeferredSyncScope&) = delete;
};
inline hipError_t final_sync(hipStream_t s) { return deferred_sync() ? hipSuccess : hipStreamSynchronize(s); }

} // namespace gr4pm
