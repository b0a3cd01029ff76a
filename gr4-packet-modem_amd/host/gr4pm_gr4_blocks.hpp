// gr4pm_gr4_blocks.hpp -- GNU Radio 4.0 block wrappers over the C ABI (include/gr4pm_hip.h).
//
// Same class templates (parameter lists and defaults), ports, settings and tag keys as the
// reference blocks, so that a flowgraph written against
//   <gnuradio-4.0/packet-modem/syncword_detection.hpp> etc.
// compiles against these instead: put gr4-packet-modem_amd/host in front of the reference's
// blocks/include on the include path -- host/gnuradio-4.0/packet-modem/*.hpp carry the reference's
// header names and pull the classes below into gr::packet_modem.  The classes
// themselves live in gr::packet_modem::hip so that they can coexist with the CPU blocks.
//
// Supported instantiations are the ones the reference's receiver and benchmarks use
// (packet_receiver.hpp:84-125, apps/packet_transceiver.cpp:71-73, python/bindings/register_*.cpp):
// complex<float> items, float taps / phases; everything else is a static_assert, not a silent CPU path.
//
// gnuradio4 is not in this repository's image.  The header is compiled by __graft_entry__.build()
// and driven by tests/gr4_blocks_driver.cpp against tests/gr4_stub/ (a test-only stand-in for the
// GR4 surface used here: gr::Block<D>, PortIn/PortOut, ConsumableSpan/PublishableSpan (size, begin,
// consume, publish), input_tags_present(), mergedInputTag(), publishTag(), gr::exception,
// ENABLE_REFLECTION) -- SURVEY.md 8(b).
//
// Staging.  GR4 port buffers are host memory; the kernels work on device memory.  A wrapped block
// uploads its input span and downloads its output span -- unless the span it is handed was written
// by another wrapped block: every block leaves its device-side output registered in a process-wide
// arena keyed by the HOST address of the span it published (detail::Arena), and a consumer whose
// input span lies inside such a registration reads the device copy (no H2D).  With the setting
// `host_output = false` a block also skips its D2H copy; if a consumer that is not a wrapped block
// (or a partial consumer) shows up later, the unconsumed remainder is written back to the host ring
// before the device buffer is reused.  So a chain of wrapped blocks stages once at its entry and
// once at its exit.  Every HIP call is checked.
#pragma once
#include <gnuradio-4.0/Block.hpp>
#include <gnuradio-4.0/reflection.hpp>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <complex>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <map>
#include <condition_variable>
#include <mutex>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

#include "gr4pm_hip.h"

// Trace prints on entry and exit of every processBulk(), like the reference's blocks (README.md:144-150,
// e.g. syncword_detection.hpp:207-212,351-353): compiled in with -DTRACE (cmake -D CMAKE_CXX_FLAGS=-DTRACE ..), same
// wording ("<name>::processBulk(inSpan.size() = .., outSpan.size = ..)", "<name> consumed = .., published = ..").
#ifdef TRACE
#define GR4PM_TRACE_ENTRY(in_size, out_size)                                                                       \
    std::fprintf(stderr, "%s::processBulk(inSpan.size() = %zu, outSpan.size = %zu)\n", std::string(this->name).c_str(), \
                 static_cast<size_t>(in_size), static_cast<size_t>(out_size))
#define GR4PM_TRACE_EXIT(consumed, published)                                                                      \
    std::fprintf(stderr, "%s consumed = %zu, published = %zu\n", std::string(this->name).c_str(),                  \
                 static_cast<size_t>(consumed), static_cast<size_t>(published))
#else
#define GR4PM_TRACE_ENTRY(in_size, out_size) ((void)0)
#define GR4PM_TRACE_EXIT(consumed, published) ((void)0)
#endif

namespace gr::packet_modem {
// pdu.hpp:15-21 (the reference's own header defines it; the drop-in of zmq_pdu_pub_sink.hpp includes that when it is on
// the include path).  Only named here: ZmqPduPubSink<T>'s port carries Pdu<T>.
template <typename T>
struct Pdu;
} // namespace gr::packet_modem

namespace gr::packet_modem::hip {

namespace detail {
inline void check(gr4pm_status s, const char* what)
{
    if (s < 0) throw gr::exception(std::string(what) + ": " + gr4pm_last_error());
}
inline void check_hip(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw gr::exception(std::string(what) + ": " + hipGetErrorString(e));
}
// largest chunk a wrapper hands to the library in one call (the handles are created for it)
inline size_t max_items()
{
    static const size_t n = [] {
        const char* e = std::getenv("GR4PM_GR4_MAX_ITEMS");
        const size_t v = e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : (size_t{ 1 } << 22);
        return std::max<size_t>(v, 4096);
    }();
    return n;
}

// Device-side shadows of published output spans, keyed by host address.
//
// Double-mapped buffers.  gnuradio4's CircularBuffer maps its storage twice, back to back: the items a producer wrote
// through a span that ran past the end of the first mapping are seen by the consumer, after the wrap, at addresses one
// ring size LOWER.  A raw address comparison would miss them (and with host_output = false there is no host copy to
// fall back to).  An integrator that runs wrapped blocks with host_output = false over such buffers registers every
// buffer once with add_mirrored_ring(base, bytes); addresses inside [base, base + 2 bytes) are then compared modulo the
// ring.  Without a registration addresses are compared as they are (single-mapped buffers, the test stand-in).
//
// Threads (round 4).  The two ends of an edge may run concurrently (the reference's multi-threaded schedulers,
// benchmarks/README.md:8-26): a producer's next processBulk() then starts while the consumer is still reading the
// device copy of the span before.  Hence a producer owns a small POOL of device buffers, a consumer PINS the registration
// it reads (find() ... release()), and a producer never overwrites a buffer that is pinned: it takes another buffer of
// its pool (fully consumed or never published first, then a new one up to kMaxPool, then -- after a grace period for the
// reader -- the oldest unpinned one, whose unconsumed rest is written back to the host span so that the consumer falls
// back to H2D; only if every buffer is pinned it waits for a release).  Everything under one mutex; the copies of a write-back run inside it.
class Arena
{
public:
    static constexpr int kMaxPool = 16; // spans a producer can be ahead of its reader: port buffer size / smallest span
    struct Seg {
        const void* owner;
        int buf;          // index in the owner's pool
        uint64_t id;
        int users;        // consumers reading dev right now
        const char* host; // canonical (first-mapping) address of the first byte
        char* dev;
        size_t bytes, consumed;
        bool on_host;     // the host span already holds the data
        size_t ring;      // size of the mirrored ring the span lives in, 0: none
        const char* raw;  // address as the producer saw it (write-back goes there)
    };
    static Arena& instance()
    {
        static Arena a;
        return a;
    }
    void add_mirrored_ring(const void* base, size_t bytes)
    {
        std::lock_guard<std::mutex> g(_m);
        _rings.push_back({ static_cast<const char*>(base), bytes });
    }
    // consumer side: device address of [host, host + bytes) if a producer left it here; the registration stays pinned
    // (its buffer is not reused) until release(*pin)
    const void* find(const void* host, size_t bytes, uint64_t* pin)
    {
        std::lock_guard<std::mutex> g(_m);
        size_t ring = 0;
        const char* h = canonical(static_cast<const char*>(host), &ring);
        for (auto& s : _segs) {
            const ptrdiff_t off = offset_in(s, h, bytes, ring);
            if (off >= 0) {
                ++s.users;
                ++_hits;
                if (std::find(_read_owners.begin(), _read_owners.end(), s.owner) == _read_owners.end()) _read_owners.push_back(s.owner);
                *pin = s.id;
                return s.dev + off;
            }
        }
        ++_misses;
        *pin = 0;
        return nullptr;
    }
    void release(uint64_t pin)
    {
        if (pin == 0) return;
        {
            std::lock_guard<std::mutex> g(_m);
            for (auto* list : { &_segs, &_zombies })
                for (size_t i = 0; i < list->size(); ++i)
                    if ((*list)[i].id == pin) {
                        --(*list)[i].users;
                        if (list == &_zombies && (*list)[i].users <= 0) list->erase(list->begin() + static_cast<ptrdiff_t>(i));
                        goto done;
                    }
        }
    done:
        _cv.notify_all();
    }
    void consumed(const void* host, size_t bytes)
    {
        std::lock_guard<std::mutex> g(_m);
        size_t ring = 0;
        const char* h = canonical(static_cast<const char*>(host), &ring);
        for (auto& s : _segs) {
            const ptrdiff_t off = offset_in(s, h, bytes, ring);
            if (off >= 0) s.consumed = std::max(s.consumed, static_cast<size_t>(off) + bytes);
        }
        _cv.notify_all();
    }
    // write back what the host does not hold yet of every registration that overlaps [host, host + bytes)
    void flush(const void* host, size_t bytes)
    {
        std::lock_guard<std::mutex> g(_m);
        size_t ring = 0;
        const char* h = canonical(static_cast<const char*>(host), &ring);
        for (auto& s : _segs)
            if (!s.on_host && overlaps(s, h, bytes, ring)) {
                check_hip(hipMemcpy(const_cast<char*>(s.raw), s.dev, s.bytes, hipMemcpyDeviceToHost), "arena flush");
                s.on_host = true;
            }
    }
    // producer side: which buffer of its pool (`n_bufs` of them so far, the last one used was `cur`) the owner may
    // overwrite now; n_bufs = "add a buffer".  The registration of the chosen buffer is retired: what nobody consumed and
    // the host does not hold yet is written back first.
    int acquire(const void* owner, int n_bufs, int cur)
    {
        std::unique_lock<std::mutex> g(_m);
        bool waited = false;
        for (;;) {
            int oldest_unpinned = -1;
            bool any_pinned = false;
            uint64_t oldest_id = ~uint64_t{ 0 };
            for (int k = 1; k <= n_bufs; ++k) {
                const int b = (cur + k) % n_bufs;
                bool pinned = false, pending = false;
                uint64_t id = 0;
                for (const auto* list : { &_segs, &_zombies })
                    for (const auto& s : *list)
                        if (s.owner == owner && s.buf == b) {
                            pinned = pinned || s.users > 0;
                            pending = pending || (list == &_segs && !s.on_host && s.consumed < s.bytes);
                            id = std::max(id, s.id);
                        }
                any_pinned = any_pinned || pinned;
                if (pinned) continue;
                if (!pending) {
                    retire_locked(owner, b);
                    return b;
                }
                if (id < oldest_id) oldest_id = id, oldest_unpinned = b;
            }
            // The pool grows only for a reader that reads from the device (a pinned buffer, or a wrapped block that has
            // read this producer's spans before).  With a downstream that is not a wrapped block nothing is ever
            // reported consumed: its producer keeps ONE buffer in turn and the oldest span goes to the host at once
            // (sixteen device buffers per output and a write-back into already published port memory otherwise).
            const bool has_reader = std::find(_read_owners.begin(), _read_owners.end(), owner) != _read_owners.end();
            if (n_bufs == 0 || (n_bufs < kMaxPool && (any_pinned || has_reader))) return n_bufs;
            // every buffer holds a span that has not been read to its end.  Evicting the oldest would take away exactly
            // what the reader needs next, so give the reader a moment first (release() and consumed() wake this up) --
            // if this producer's spans have ever been read from the device at all: a reader that is not a wrapped block
            // never reports anything, and its producer must not wait for it.  After the grace period the oldest span goes
            // to the host
            if (has_reader && !waited && _cv.wait_for(g, std::chrono::milliseconds(2)) == std::cv_status::no_timeout) continue;
            waited = true;
            if (oldest_unpinned >= 0) {
                retire_locked(owner, oldest_unpinned);
                return oldest_unpinned;
            }
            _cv.wait(g); // every buffer is being read: until a consumer lets go
        }
    }
    // the owner goes away (or frees a buffer): forget its registrations (unconsumed data is written back first)
    void retire(const void* owner, int buf = -1)
    {
        std::unique_lock<std::mutex> g(_m);
        // a buffer that is still being read is not freed under its reader
        _cv.wait(g, [&] {
            for (const auto* list : { &_segs, &_zombies })
                for (const auto& s : *list)
                    if (s.owner == owner && (buf < 0 || s.buf == buf) && s.users > 0) return false;
            return true;
        });
        retire_locked(owner, buf);
        if (buf < 0) _read_owners.erase(std::remove(_read_owners.begin(), _read_owners.end(), owner), _read_owners.end());
    }
    void publish(const void* owner, int buf, const void* host, void* dev, size_t bytes, bool on_host)
    {
        if (bytes == 0) return;
        std::lock_guard<std::mutex> g(_m);
        size_t ring = 0;
        const char* raw = static_cast<const char*>(host);
        const char* h = canonical(raw, &ring);
        // a new span over the same host memory supersedes whatever was registered there (the port buffer hands a region
        // out for writing only after its readers are through with it; a reader that is not keeps its buffer pinned)
        // -- and the part of an older span that the new one does NOT cover may still be unread (a producer a whole ring
        // ahead overwrites the head of a span whose tail the consumer has yet to see): it goes to the host span first
        for (size_t i = 0; i < _segs.size();)
            if (overlaps(_segs[i], h, bytes, ring)) {
                const Seg o = _segs[i];
                if (!o.on_host && o.consumed < o.bytes)
                    check_hip(hipMemcpy(const_cast<char*>(o.raw) + o.consumed, o.dev + o.consumed, o.bytes - o.consumed,
                                        hipMemcpyDeviceToHost),
                              "arena write-back (superseded span)");
                if (o.users > 0) _zombies.push_back(o);
                _segs.erase(_segs.begin() + static_cast<ptrdiff_t>(i));
            } else {
                ++i;
            }
        _segs.push_back({ owner, buf, ++_next_id, 0, h, static_cast<char*>(dev), bytes, 0, on_host, ring, raw });
    }

private:
    struct Ring {
        const char* base;
        size_t bytes;
    };
    void retire_locked(const void* owner, int buf)
    {
        for (size_t i = 0; i < _segs.size();) {
            if (_segs[i].owner == owner && (buf < 0 || _segs[i].buf == buf)) {
                const Seg s = _segs[i];
                if (!s.on_host && s.consumed < s.bytes)
                    check_hip(hipMemcpy(const_cast<char*>(s.raw) + s.consumed, s.dev + s.consumed,
                                        s.bytes - s.consumed, hipMemcpyDeviceToHost),
                              "arena write-back");
                _segs.erase(_segs.begin() + static_cast<ptrdiff_t>(i));
            } else {
                ++i;
            }
        }
    }
    // first-mapping address of h and the size of its ring (0: not inside a registered ring)
    const char* canonical(const char* h, size_t* ring) const
    {
        for (const auto& r : _rings)
            if (h >= r.base && h < r.base + 2 * r.bytes) {
                *ring = r.bytes;
                return r.base + static_cast<size_t>(h - r.base) % r.bytes;
            }
        *ring = 0;
        return h;
    }
    // byte offset of [h, h + bytes) inside the registration, -1 if it is not wholly inside.  In a ring a registration
    // may run past the end of the first mapping: a span that begins after the wrap is then one ring size further on.
    static ptrdiff_t offset_in(const Seg& s, const char* h, size_t bytes, size_t ring)
    {
        if (h >= s.host && h + bytes <= s.host + s.bytes) return h - s.host;
        if (ring != 0 && ring == s.ring && h + ring >= s.host && h + ring + bytes <= s.host + s.bytes) return h + ring - s.host;
        return -1;
    }
    static bool overlaps(const Seg& s, const char* h, size_t bytes, size_t ring)
    {
        if (h < s.host + s.bytes && s.host < h + bytes) return true;
        if (ring != 0 && ring == s.ring) {
            if (h + ring < s.host + s.bytes && s.host < h + ring + bytes) return true; // query after the wrap
            if (h < s.host + ring + s.bytes && s.host + ring < h + bytes) return true; // registration after the wrap
        }
        return false;
    }
    std::mutex _m;
    std::condition_variable _cv;
    std::vector<Seg> _segs, _zombies; // zombies: superseded while a consumer was still reading them
    std::vector<Ring> _rings;
    uint64_t _next_id = 0;
    size_t _hits = 0, _misses = 0;
    std::vector<const void*> _read_owners; // producers whose device copies a wrapped consumer has read

public:
    // how many input spans were found on the device / had to be uploaded (tests, tuning)
    std::pair<size_t, size_t> lookups()
    {
        std::lock_guard<std::mutex> g(_m);
        return { _hits, _misses };
    }
};

// device staging: the input side uploads into one buffer that grows on demand (or reads a producer's device copy);
// the output side rotates through a small pool so that a consumer in another thread can still read the span before
template <typename T>
struct DeviceStage {
    struct Buf {
        T* p = nullptr;
        size_t n = 0;
    };
    std::vector<Buf> bufs{ Buf{} };
    int cur = 0;
    uint64_t pin = 0; // the producer registration this stage is reading (input side)
    DeviceStage() = default;
    DeviceStage(const DeviceStage&) = delete;
    DeviceStage& operator=(const DeviceStage&) = delete;
    ~DeviceStage()
    {
        try {
            Arena::instance().release(pin);
            Arena::instance().retire(this);
        } catch (...) {
        }
        for (auto& b : bufs)
            if (b.p) (void)hipFree(b.p);
    }
    T* get(size_t count) // the current buffer, at least `count` items (its registration has been retired)
    {
        Buf& b = bufs[static_cast<size_t>(cur)];
        if (count > b.n) {
            if (b.p) check_hip(hipFree(b.p), "hipFree");
            b.p = nullptr;
            b.n = 0;
            check_hip(hipMalloc(reinterpret_cast<void**>(&b.p), std::max<size_t>(count, 1) * sizeof(T)), "hipMalloc");
            b.n = count;
        }
        return b.p;
    }
    // input of a block: the producer's device copy if the span was published by a wrapped block, else H2D
    template <typename H>
    const T* in(const H* host, size_t count)
    {
        static_assert(sizeof(H) == sizeof(T));
        Arena::instance().release(pin); // the span of the call before
        pin = 0;
        if (count == 0) return get(1);
        if (const void* d = Arena::instance().find(host, count * sizeof(T), &pin)) {
            if (std::getenv("GR4PM_GR4_DEBUG")) std::fprintf(stderr, "arena hit  %p %zu\n", static_cast<const void*>(host), count);
            return static_cast<const T*>(d);
        }
        if (std::getenv("GR4PM_GR4_DEBUG")) std::fprintf(stderr, "arena miss %p %zu\n", static_cast<const void*>(host), count);
        // a span that only partly lies in a device-only registration: bring the host copy up to date first
        Arena::instance().flush(host, count * sizeof(T));
        T* d = get(count);
        check_hip(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
        return d;
    }
    // buffer for `count` output items: one of the pool that no consumer is reading
    T* out(size_t count)
    {
        const int b = Arena::instance().acquire(this, static_cast<int>(bufs.size()), cur);
        if (b == static_cast<int>(bufs.size())) bufs.push_back(Buf{});
        cur = b;
        return get(count);
    }
    // `count` items of the output buffer become the host span [host, host + count)
    template <typename H>
    void publish(H* host, size_t count, bool host_output)
    {
        static_assert(sizeof(H) == sizeof(T));
        if (count == 0) return;
        T* p = bufs[static_cast<size_t>(cur)].p;
        if (host_output) check_hip(hipMemcpy(host, p, count * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        Arena::instance().publish(this, cur, host, p, count * sizeof(T), host_output);
    }
    // the input span of this call has been used (its device copy may go)
    void done()
    {
        Arena::instance().release(pin);
        pin = 0;
    }
};
inline void consumed(const void* host, size_t bytes) { Arena::instance().consumed(host, bytes); }

inline gr::property_map to_map(const gr4pm_tag& t)
{
    // syncword_detection.hpp:106-114
    return { { "syncword_amplitude", t.amplitude }, { "syncword_phase", t.phase },
             { "syncword_freq", t.freq },           { "syncword_freq_bin", t.freq_bin },
             { "syncword_noise_power", t.noise_power }, { "syncword_esn0_db", t.esn0_db },
             { "syncword_time_est", t.time_est } };
}
inline gr4pm_tag from_map(const gr::property_map& m, uint64_t index)
{
    gr4pm_tag t{};
    t.index = index;
    if (m.contains("syncword_amplitude")) {
        t.flags |= GR4PM_TAG_SYNCWORD;
        t.amplitude = pmtv::cast<float>(m.at("syncword_amplitude"));
        if (m.contains("syncword_phase")) t.phase = pmtv::cast<float>(m.at("syncword_phase"));
        if (m.contains("syncword_freq")) t.freq = pmtv::cast<double>(m.at("syncword_freq"));
        if (m.contains("syncword_time_est")) t.time_est = pmtv::cast<float>(m.at("syncword_time_est"));
    }
    for (const auto& [k, v] : m)
        if (!k.starts_with("syncword_")) t.flags |= GR4PM_TAG_OTHER;
    return t;
}
inline int constellation_id(const std::string& s)
{
    std::string u;
    for (char c : s) u.push_back(static_cast<char>(std::toupper(static_cast<unsigned char>(c))));
    if (u == "PILOT") return 0;
    if (u == "BPSK") return 1;
    if (u == "QPSK") return 2;
    throw gr::exception("unknown constellation " + s); // enum_cast(...).value() throws, costas_loop.hpp:59-61
}
template <typename T>
inline constexpr bool is_c64 = std::is_same_v<T, std::complex<float>>;
// gr::packet_modem::Pdu<T> (pdu.hpp:15-21) by shape: value_type, data = std::vector<value_type>, tags
template <typename P>
concept PduLike = requires(P p) {
    typename P::value_type;
    requires std::is_same_v<std::remove_cvref_t<decltype(p.data)>, std::vector<typename P::value_type>>;
    p.tags.size();
};
} // namespace detail

using c64 = std::complex<float>;

// ---------------------------------------------------------------- SyncwordDetection
// replaces gr::packet_modem::SyncwordDetection (syncword_detection.hpp:32-357)
class SyncwordDetection : public gr::Block<SyncwordDetection>
{
    gr4pm_syncword_detection* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;
    std::vector<gr4pm_tag> _tags;
    size_t _max_items = 0;

public:
    size_t _syncword_samples_size = 0; // read by tests/apps (qa_syncword_detection.cpp:133)
    gr::PortIn<c64> in;
    gr::PortOut<c64> out;
    size_t fft_size = 2048;
    size_t samples_per_symbol = 4;
    std::vector<float> rrc_taps;
    std::vector<uint8_t> syncword;
    std::vector<c64> constellation;
    int min_freq_bin = 0;
    int max_freq_bin = 0;
    uint64_t time_threshold = 768;
    float power_threshold = 9.5;
    bool host_output = true; // false: downstream is a wrapped block, keep the samples on the device

    SyncwordDetection() = default;
    SyncwordDetection(const SyncwordDetection&) = delete;
    ~SyncwordDetection() { gr4pm_syncword_detection_destroy(_h); }

    void start()
    {
        gr4pm_syncword_detection_destroy(_h);
        _h = nullptr;
        gr4pm_syncword_detection_params p{};
        p.fft_size = fft_size;
        p.samples_per_symbol = samples_per_symbol;
        p.rrc_taps = rrc_taps.data();
        p.n_rrc_taps = rrc_taps.size();
        p.syncword = syncword.data();
        p.n_syncword = syncword.size();
        p.constellation = reinterpret_cast<const gr4pm_c64*>(constellation.data());
        p.n_constellation = constellation.size();
        p.min_freq_bin = min_freq_bin;
        p.max_freq_bin = max_freq_bin;
        p.time_threshold = time_threshold;
        p.power_threshold = power_threshold;
        p.n_channels = 1;
        _max_items = std::max(detail::max_items(), 2 * fft_size);
        p.max_items = _max_items;
        detail::check(gr4pm_syncword_detection_create(&p, &_h), "SyncwordDetection::start"); // :146,151 throw
        _syncword_samples_size = gr4pm_syncword_detection_syncword_samples_size(_h);
        in.min_samples = fft_size; // syncword_detection.hpp:200-201
        out.min_samples = fft_size;
        _tags.resize(_max_items / (time_threshold + 1) + 16);
    }

    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (inSpan.size() < fft_size) { // :215-227
            if (!inSpan.consume(0)) throw gr::exception("consume failed");
            outSpan.publish(0);
            return gr::work::Status::INSUFFICIENT_INPUT_ITEMS;
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size(), _max_items });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        size_t n_done = 0, n_tags = 0;
        detail::check(gr4pm_syncword_detection_process(_h, din, n, n, dout, n, &n_done, _tags.data(), _tags.size(),
                                                       &n_tags),
                      "SyncwordDetection::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), n_done, host_output);
        detail::consumed(hin, n_done * sizeof(c64));
        _din.done();
        for (size_t i = 0; i < n_tags; ++i)
            out.publishTag(detail::to_map(_tags[i]), static_cast<ssize_t>(_tags[i].index)); // :321-324
        if (!inSpan.consume(n_done)) throw gr::exception("consume failed"); // :346-348
        outSpan.publish(n_done);
        GR4PM_TRACE_EXIT(n_done, n_done);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- rotators
// replaces gr::packet_modem::Rotator<T = float> (rotator.hpp:20-65)
template <typename T = float>
class Rotator : public gr::Block<Rotator<T>>
{
    static_assert(std::is_same_v<T, float>, "gr4pm: Rotator is built for T = float (complex<float> items)");
    gr4pm_rotator* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<std::complex<T>> in;
    gr::PortOut<std::complex<T>> out;
    T phase_incr = 0;
    bool host_output = true;
    Rotator() = default;
    Rotator(const Rotator&) = delete;
    ~Rotator() { gr4pm_rotator_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&) { start(); } // :44-48
    void start()                                                                         // :50-54
    {
        gr4pm_rotator_destroy(_h);
        _h = nullptr;
        gr4pm_rotator_params p{ 0, phase_incr, 0, 1, nullptr };
        detail::check(gr4pm_rotator_create(&p, &_h), "Rotator::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        detail::check(gr4pm_rotator_process(_h, din, n, n, dout, nullptr, nullptr, 0), "Rotator");
        _dout.publish(std::to_address(outSpan.begin()), n, host_output);
        detail::consumed(hin, n * sizeof(c64));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        GR4PM_TRACE_EXIT(n, n);
        return gr::work::Status::OK;
    }
};

// replaces gr::packet_modem::CoarseFrequencyCorrection<T = float> (coarse_frequency_correction.hpp:20-99)
template <typename T = float>
class CoarseFrequencyCorrection : public gr::Block<CoarseFrequencyCorrection<T>>
{
    static_assert(std::is_same_v<T, float>, "gr4pm: CoarseFrequencyCorrection is built for T = float");
    gr4pm_rotator* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<std::complex<T>> in;
    gr::PortOut<std::complex<T>> out;
    size_t delay = 0;
    bool host_output = true;
    CoarseFrequencyCorrection() = default;
    CoarseFrequencyCorrection(const CoarseFrequencyCorrection&) = delete;
    ~CoarseFrequencyCorrection() { gr4pm_rotator_destroy(_h); }
    void start() // :61-65
    {
        gr4pm_rotator_destroy(_h);
        _h = nullptr;
        gr4pm_rotator_params p{ 1, 0.0f, delay, 1, nullptr };
        detail::check(gr4pm_rotator_create(&p, &_h), "CoarseFrequencyCorrection::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) { // :76-82: the tag refers to inSpan[0]
            tag = detail::from_map(this->mergedInputTag().map, 0);
            if (this->mergedInputTag().map.contains("syncword_freq")) {
                tag.flags |= GR4PM_TAG_SYNCWORD;
                tag.freq = pmtv::cast<double>(this->mergedInputTag().map.at("syncword_freq"));
                n_tags = 1;
            }
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        detail::check(gr4pm_rotator_process(_h, din, n, n, dout, &tag, nullptr, n_tags), "CoarseFrequencyCorrection");
        _dout.publish(std::to_address(outSpan.begin()), n, host_output);
        detail::consumed(hin, n * sizeof(c64));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        GR4PM_TRACE_EXIT(n, n);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SyncwordDetectionFilter
// replaces gr::packet_modem::SyncwordDetectionFilter<T = complex<float>> (syncword_detection_filter.hpp:10-211)
template <typename T = std::complex<float>>
class SyncwordDetectionFilter : public gr::Block<SyncwordDetectionFilter<T>>
{
    static_assert(detail::is_c64<T>, "gr4pm: SyncwordDetectionFilter is built for complex<float> items");
    gr4pm_syncword_detection_filter* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<gr::Message, gr::Async> parsed_header;
    gr::PortIn<gr::Message, gr::Async> ignored_syncword;
    gr::PortIn<T> in;
    gr::PortOut<T> out;
    size_t samples_per_symbol = 4;
    size_t syncword_size = 64;
    size_t header_size = 128;
    bool host_output = true;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    SyncwordDetectionFilter() = default;
    SyncwordDetectionFilter(const SyncwordDetectionFilter&) = delete;
    ~SyncwordDetectionFilter() { gr4pm_syncword_detection_filter_destroy(_h); }
    void start()
    {
        gr4pm_syncword_detection_filter_destroy(_h);
        _h = nullptr;
        gr4pm_syncword_detection_filter_params p{ samples_per_symbol, syncword_size, header_size, nullptr };
        detail::check(gr4pm_syncword_detection_filter_create(&p, &_h), "SyncwordDetectionFilter::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& headerSpan,
                                 const gr::ConsumableSpan auto& ignoredSpan,
                                 const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        int head_flags = 0;
        gr::property_map syncword_keys, other_keys;
        if (this->input_tags_present()) { // :75-93
            for (const auto& [key, val] : this->mergedInputTag().map) {
                if (key.starts_with("syncword_")) {
                    head_flags |= GR4PM_TAG_SYNCWORD;
                    syncword_keys[key] = val;
                } else {
                    head_flags |= GR4PM_TAG_OTHER;
                    other_keys[key] = val;
                }
            }
        }
        std::vector<gr4pm_header_msg> msgs;
        for (size_t i = 0; i < headerSpan.size(); ++i) { // :134-152
            const auto& meta = headerSpan[i].data.value();
            gr4pm_header_msg m{};
            m.invalid_header = meta.contains("invalid_header") ? 1 : 0;
            if (!m.invalid_header) m.packet_length = pmtv::cast<uint64_t>(meta.at("packet_length"));
            msgs.push_back(m);
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        size_t consumed = 0, hc = 0, ic = 0;
        int out_flags = 0;
        detail::check(gr4pm_syncword_detection_filter_process(_h, din, n, dout, n, head_flags, msgs.data(),
                                                              msgs.size(), ignoredSpan.size(), &consumed, &hc,
                                                              &ic, &out_flags),
                      "SyncwordDetectionFilter::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), consumed, host_output);
        detail::consumed(hin, consumed * sizeof(c64));
        _din.done();
        gr::property_map output_tags; // :82-104
        if (out_flags & GR4PM_TAG_SYNCWORD) output_tags.insert(syncword_keys.begin(), syncword_keys.end());
        if (out_flags & GR4PM_TAG_OTHER) output_tags.insert(other_keys.begin(), other_keys.end());
        if (!output_tags.empty()) out.publishTag(output_tags, 0);
        if (!inSpan.consume(consumed)) throw gr::exception("inSpan.consume failed");
        if (!headerSpan.consume(hc)) throw gr::exception("headerSpan.consume failed");
        if (!ignoredSpan.consume(ic)) throw gr::exception("ignoredSpan.consume failed");
        outSpan.publish(consumed);
        this->_mergedInputTag.map.clear(); // :125,202
        GR4PM_TRACE_EXIT(consumed, consumed);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SymbolFilter
// replaces gr::packet_modem::SymbolFilter<TIn, TOut = TIn, TTaps = TIn> (symbol_filter.hpp:13-253)
template <typename TIn, typename TOut = TIn, typename TTaps = TIn>
class SymbolFilter : public gr::Block<SymbolFilter<TIn, TOut, TTaps>, gr::Resampling<>>
{
    // <complex<float>, complex<float>, float> (packet_receiver.hpp:111) and <float, float, float>
    // (test/qa_symbol_filter.cpp:17-63, python/bindings/register_symbol_filter.cpp:9-15): the ABI's item_kind 0 / 1
    static_assert(std::is_same_v<TIn, TOut> && (detail::is_c64<TIn> || std::is_same_v<TIn, float>) && std::is_same_v<TTaps, float>,
                  "gr4pm: SymbolFilter is built for <complex<float>, complex<float>, float> and <float, float, float>");
    using Item = std::conditional_t<detail::is_c64<TIn>, gr4pm_c64, float>;
    gr4pm_symbol_filter* _h = nullptr;
    detail::DeviceStage<Item> _din, _dout;
    // full maps of the tags queued inside the filter (opaque keys travel with them), keyed by the handle the tag carries
    // through the library in its `user` field (the caller's cookie; round 6: no longer freq_bin); an entry leaves when
    // its tag is published (it used to stay for the life of the block: one property_map leaked per packet)
    std::map<int32_t, gr::property_map> _held;
    int32_t _next_handle = 0;
    std::vector<gr4pm_tag> _tags_out;

public:
    gr::PortIn<TIn> in;
    gr::PortOut<TOut> out;
    size_t samples_per_symbol = 4;
    std::vector<TTaps> taps;
    size_t num_arms = 32;
    size_t delay = 0;
    bool host_output = true;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;
    size_t held_tag_maps() const { return _held.size(); } // tests: stays at the number of tags inside the filter

    SymbolFilter() = default;
    SymbolFilter(const SymbolFilter&) = delete;
    ~SymbolFilter() { gr4pm_symbol_filter_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        gr4pm_symbol_filter_destroy(_h);
        _h = nullptr;
        gr4pm_symbol_filter_params p{ samples_per_symbol, taps.data(), taps.size(), num_arms, delay, detail::is_c64<TIn> ? 0 : 1, nullptr };
        detail::check(gr4pm_symbol_filter_create(&p, &_h), "SymbolFilter::settingsChanged"); // :67-73 throw
        // input_chunk_size / output_chunk_size stay 1 : 1 like the reference's (symbol_filter.hpp:74-81 has the
        // samples_per_symbol : 1 ratio commented out: input tags are not aligned to samples_per_symbol blocks, and a
        // scheduler that sizes spans in that ratio would never offer the items in front of such a tag)
        _tags_out.resize(64);
        _held.clear();
    }
    void start()
    {
        if (!_h) settingsChanged({}, {});
        detail::check(gr4pm_symbol_filter_reset(_h), "SymbolFilter::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) settingsChanged({}, {});
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) { // :127-206: the tag refers to inSpan[0]
            tag = detail::from_map(this->mergedInputTag().map, 0);
            tag.user = _next_handle; // handle of the full map
            _held.emplace(_next_handle, this->mergedInputTag().map);
            _next_handle = _next_handle == std::numeric_limits<int32_t>::max() ? 0 : _next_handle + 1;
            n_tags = 1;
        }
        const size_t n = std::min(inSpan.size(), detail::max_items());
        const size_t cap = std::min(outSpan.size(), detail::max_items());
        const TIn* hin = std::to_address(inSpan.begin());
        const Item* din = _din.in(hin, n);
        Item* dout = _dout.out(cap);
        size_t n_out_tags = 0, consumed = 0, produced = 0;
        detail::check(gr4pm_symbol_filter_process(_h, din, n, dout, cap, &tag, n_tags, _tags_out.data(),
                                                  _tags_out.size(), &n_out_tags, &consumed, &produced),
                      "SymbolFilter::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), produced, host_output);
        detail::consumed(hin, consumed * sizeof(TIn));
        _din.done();
        for (size_t i = 0; i < n_out_tags; ++i) { // :218-228 re-timed tags, :152-155 adjusted phase
            auto node = _held.extract(_tags_out[i].user); // published once: the entry goes with it
            if (node.empty()) throw gr::exception("SymbolFilter: a published tag has no queued map");
            auto& map = node.mapped();
            if (_tags_out[i].flags & GR4PM_TAG_SYNCWORD) map["syncword_phase"] = _tags_out[i].phase;
            out.publishTag(map, static_cast<ssize_t>(_tags_out[i].index));
        }
        if (!inSpan.consume(consumed)) throw gr::exception("consume failed"); // :240-243
        outSpan.publish(produced);
        GR4PM_TRACE_EXIT(consumed, produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- CostasLoop
// replaces gr::packet_modem::CostasLoop<T = float, TPhase = float> (costas_loop.hpp:15-149)
template <typename T = float, typename TPhase = float>
class CostasLoop : public gr::Block<CostasLoop<T, TPhase>>
{
    static_assert(std::is_same_v<T, float> && std::is_same_v<TPhase, float>, "gr4pm: CostasLoop is built for <float, float>");
    gr4pm_costas_loop* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<std::complex<T>> in;
    gr::PortOut<std::complex<T>> out;
    double loop_bandwidth = 0.01;
    std::string constellation = "BPSK";
    bool host_output = true;

    CostasLoop() = default;
    CostasLoop(const CostasLoop&) = delete;
    ~CostasLoop() { gr4pm_costas_loop_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&) // :52-88 (also driven by tags)
    {
        if (!_h) {
            gr4pm_costas_loop_params p{ loop_bandwidth, detail::constellation_id(constellation), 1, nullptr };
            detail::check(gr4pm_costas_loop_create(&p, &_h), "CostasLoop::settingsChanged");
        } else {
            detail::check(gr4pm_costas_loop_set(_h, loop_bandwidth, detail::constellation_id(constellation)),
                          "CostasLoop::set");
        }
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) settingsChanged({}, {});
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) {
            const auto& map = this->mergedInputTag().map;
            // the runtime applies tag keys that name a setting before the call (costas_loop.hpp:52-88 via
            // settings auto-update): "constellation" / "loop_bandwidth" from PayloadMetadataInsert
            bool changed = false;
            if (map.contains("constellation")) {
                constellation = pmtv::cast<std::string>(map.at("constellation"));
                changed = true;
            }
            if (map.contains("loop_bandwidth")) {
                loop_bandwidth = pmtv::cast<double>(map.at("loop_bandwidth"));
                changed = true;
            }
            if (changed) settingsChanged({}, {});
            if (map.contains("syncword_phase")) { // :101-106
                tag.index = 0;
                tag.flags = GR4PM_TAG_SYNCWORD;
                tag.phase = pmtv::cast<float>(map.at("syncword_phase"));
                n_tags = 1;
            }
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        detail::check(gr4pm_costas_loop_process(_h, din, n, n, dout, &tag, nullptr, n_tags), "CostasLoop::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), n, host_output);
        detail::consumed(hin, n * sizeof(c64));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        GR4PM_TRACE_EXIT(n, n);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SyncwordWipeoff
// replaces gr::packet_modem::SyncwordWipeoff<T = complex<float>, TSyncword = float> (syncword_wipeoff.hpp:12-91)
template <typename T = std::complex<float>, typename TSyncword = float>
class SyncwordWipeoff : public gr::Block<SyncwordWipeoff<T, TSyncword>>
{
    static_assert(detail::is_c64<T> && std::is_same_v<TSyncword, float>, "gr4pm: SyncwordWipeoff is built for <complex<float>, float>");
    gr4pm_syncword_wipeoff* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<T> in;
    gr::PortOut<T> out;
    std::vector<TSyncword> syncword;
    bool host_output = true;

    SyncwordWipeoff() = default;
    SyncwordWipeoff(const SyncwordWipeoff&) = delete;
    ~SyncwordWipeoff() { gr4pm_syncword_wipeoff_destroy(_h); }
    void start()
    {
        gr4pm_syncword_wipeoff_destroy(_h);
        _h = nullptr;
        gr4pm_syncword_wipeoff_params p{ syncword.data(), syncword.size(), nullptr };
        detail::check(gr4pm_syncword_wipeoff_create(&p, &_h), "SyncwordWipeoff::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present() && this->mergedInputTag().map.contains("syncword_amplitude")) { // :53-62
            tag.index = 0;
            tag.flags = GR4PM_TAG_SYNCWORD;
            n_tags = 1;
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        detail::check(gr4pm_syncword_wipeoff_process(_h, din, n, dout, &tag, n_tags), "SyncwordWipeoff::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), n, host_output);
        detail::consumed(hin, n * sizeof(c64));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        GR4PM_TRACE_EXIT(n, n);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- InterpolatingFirFilter
// replaces gr::packet_modem::InterpolatingFirFilter<TIn, TOut = TIn, TTaps = TIn> (interpolating_fir_filter.hpp:14-103)
template <typename TIn, typename TOut = TIn, typename TTaps = TIn>
class InterpolatingFirFilter : public gr::Block<InterpolatingFirFilter<TIn, TOut, TTaps>, gr::Resampling<>>
{
    // <complex<float>, complex<float>, float> (packet_transmitter_pdu.hpp:343) and <float, float, float> (the ABI's
    // item_kind 0 / 1); Pdu<complex<float>> items: the specialisation below
    static_assert(std::is_same_v<TIn, TOut> && (detail::is_c64<TIn> || std::is_same_v<TIn, float>) && std::is_same_v<TTaps, float>,
                  "gr4pm: InterpolatingFirFilter is built for <complex<float>, complex<float>, float>, <float, float, float> "
                  "and their Pdu forms");
    using Item = std::conditional_t<detail::is_c64<TIn>, gr4pm_c64, float>;
    gr4pm_interp_fir* _h = nullptr;
    detail::DeviceStage<Item> _din, _dout;

public:
    gr::PortIn<TIn> in;
    gr::PortOut<TOut> out;
    size_t interpolation = 1;
    std::vector<TTaps> taps;
    bool host_output = true;

    InterpolatingFirFilter() = default;
    InterpolatingFirFilter(const InterpolatingFirFilter&) = delete;
    ~InterpolatingFirFilter() { gr4pm_interp_fir_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        gr4pm_interp_fir_destroy(_h);
        _h = nullptr;
        gr4pm_interp_fir_params p{ interpolation, taps.data(), taps.size(), detail::is_c64<TIn> ? 0 : 1, nullptr };
        detail::check(gr4pm_interp_fir_create(&p, &_h), "InterpolatingFirFilter::settingsChanged"); // :45-47
        this->input_chunk_size = 1; // :50-51
        this->output_chunk_size = interpolation;
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) settingsChanged({}, {});
        const size_t n = std::min({ inSpan.size(), outSpan.size() / interpolation, detail::max_items() / interpolation }); // :91
        const TIn* hin = std::to_address(inSpan.begin());
        const Item* din = _din.in(hin, n);
        Item* dout = _dout.out(n * interpolation);
        detail::check(gr4pm_interp_fir_process(_h, din, n, dout), "InterpolatingFirFilter::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), n * interpolation, host_output);
        detail::consumed(hin, n * sizeof(TIn));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n * interpolation);
        GR4PM_TRACE_EXIT(n, n * interpolation);
        return gr::work::Status::OK;
    }
};

// Pdu<TIn> -> Pdu<TOut> (interpolating_fir_filter.hpp:104-175; packet_transmitter_pdu.hpp:288): one PDU per
// processOne(), the filter history runs on ACROSS PDUs (:155-165 never clears _history), tags re-indexed by the
// interpolation (:167-171).  Any type with `value_type`, `data` (std::vector<value_type>) and `tags` is a PDU here, so
// that this header needs nothing of the reference's pdu.hpp.
template <detail::PduLike PIn, detail::PduLike POut, typename TTaps>
class InterpolatingFirFilter<PIn, POut, TTaps> : public gr::Block<InterpolatingFirFilter<PIn, POut, TTaps>>
{
    using TIn = typename PIn::value_type;
    using TOut = typename POut::value_type;
    static_assert(std::is_same_v<TIn, TOut> && (detail::is_c64<TIn> || std::is_same_v<TIn, float>) && std::is_same_v<TTaps, float>,
                  "gr4pm: InterpolatingFirFilter<Pdu> is built for Pdu<complex<float>> and Pdu<float> items, float taps");
    using Item = std::conditional_t<detail::is_c64<TIn>, gr4pm_c64, float>;
    gr4pm_interp_fir* _h = nullptr;
    detail::DeviceStage<Item> _din, _dout;

public:
    gr::PortIn<PIn> in;
    gr::PortOut<POut> out;
    size_t interpolation = 1;
    std::vector<TTaps> taps;

    InterpolatingFirFilter() = default;
    InterpolatingFirFilter(const InterpolatingFirFilter&) = delete;
    ~InterpolatingFirFilter() { gr4pm_interp_fir_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        // (the reference moves the old history into the new filter, :138-146; settings change once, before the first PDU)
        gr4pm_interp_fir_destroy(_h);
        _h = nullptr;
        gr4pm_interp_fir_params p{ interpolation, taps.data(), taps.size(), detail::is_c64<TIn> ? 0 : 1, nullptr };
        detail::check(gr4pm_interp_fir_create(&p, &_h), "InterpolatingFirFilter<Pdu>::settingsChanged"); // :124-126
    }
    [[nodiscard]] POut processOne(const PIn& pdu)
    {
        if (!_h) settingsChanged({}, {});
        POut pdu_out;
        const size_t n = pdu.data.size();
        pdu_out.data.resize(n * interpolation);
        if (n) {
            const Item* din = _din.in(pdu.data.data(), n);
            Item* dout = _dout.out(n * interpolation);
            detail::check(gr4pm_interp_fir_process(_h, din, n, dout), "InterpolatingFirFilter<Pdu>::processOne");
            detail::check_hip(hipMemcpy(pdu_out.data.data(), dout, n * interpolation * sizeof(Item), hipMemcpyDeviceToHost),
                              "hipMemcpy D2H");
            _din.done();
        }
        pdu_out.tags.reserve(pdu.tags.size());
        for (auto tag : pdu.tags) { // :167-171
            tag.index *= static_cast<decltype(tag.index)>(interpolation);
            pdu_out.tags.push_back(std::move(tag));
        }
        return pdu_out;
    }
};

// ---------------------------------------------------------------- PfbArbResampler
// replaces gr::packet_modem::PfbArbResampler<TIn, TOut = TIn, TTaps = TIn, TRate = float> (pfb_arb_resampler.hpp:23-183)
template <typename TIn, typename TOut = TIn, typename TTaps = TIn, typename TRate = float>
class PfbArbResampler : public gr::Block<PfbArbResampler<TIn, TOut, TTaps, TRate>>
{
    static_assert(detail::is_c64<TIn> && detail::is_c64<TOut> && std::is_same_v<TTaps, float> &&
                      (std::is_same_v<TRate, float> || std::is_same_v<TRate, double>),
                  "gr4pm: PfbArbResampler is built for <complex<float>, complex<float>, float, float | double>");
    gr4pm_pfb_arb_resampler* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<TIn, gr::Async> in; // :59-62: no rational resampling ratio
    gr::PortOut<TOut, gr::Async> out;
    TRate rate{ 1.0 };
    std::vector<TTaps> taps; // the reference's default (pfb_arb_taps.hpp) ships as data/pfb_arb_taps.f32
    size_t filter_size = 32;
    bool host_output = true;

    PfbArbResampler() = default;
    PfbArbResampler(const PfbArbResampler&) = delete;
    ~PfbArbResampler() { gr4pm_pfb_arb_resampler_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&)
    {
        gr4pm_pfb_arb_resampler_destroy(_h);
        _h = nullptr;
        gr4pm_pfb_arb_resampler_params p{ static_cast<double>(rate), std::is_same_v<TRate, double> ? 1 : 0,
                                          taps.data(), taps.size(), filter_size, nullptr };
        detail::check(gr4pm_pfb_arb_resampler_create(&p, &_h), "PfbArbResampler::settingsChanged"); // :70-72
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) settingsChanged({}, {});
        const size_t n = std::min(inSpan.size(), detail::max_items());
        const size_t cap = std::min(outSpan.size(), detail::max_items());
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(cap);
        size_t consumed = 0, produced = 0;
        detail::check(gr4pm_pfb_arb_resampler_process(_h, din, n, dout, cap, &consumed, &produced),
                      "PfbArbResampler::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), produced, host_output);
        detail::consumed(hin, consumed * sizeof(c64));
        _din.done();
        if (!inSpan.consume(consumed)) throw gr::exception("consume failed"); // :169-172
        outSpan.publish(produced);
        GR4PM_TRACE_EXIT(consumed, produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- PayloadMetadataInsert
// replaces gr::packet_modem::PayloadMetadataInsert<T = complex<float>> (payload_metadata_insert.hpp:12-324)
template <typename T = std::complex<float>>
class PayloadMetadataInsert : public gr::Block<PayloadMetadataInsert<T>>
{
    static_assert(detail::is_c64<T>, "gr4pm: PayloadMetadataInsert is built for complex<float> items");
    gr4pm_payload_metadata_insert* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;
    std::vector<gr4pm_packet_tag> _tags;
    gr::property_map _syncword_map; // every key of the syncword tag travels on (:104-112)
    static const char* constellation_name(int c) { return c == 0 ? "PILOT" : c == 1 ? "BPSK" : "QPSK"; }

public:
    gr::PortIn<gr::Message, gr::Async> parsed_header;
    gr::PortIn<T> in;
    gr::PortOut<T> out;
    gr::PortOut<gr::Message, gr::Async> ignored_syncword;
    size_t syncword_size = 64;
    size_t header_size = 128;
    double syncword_costas_loop_bandwidth = 0.02;
    double header_costas_loop_bandwidth = 0.01;
    double payload_costas_loop_bandwidth = 0.005;
    bool log = false;
    bool host_output = true;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    PayloadMetadataInsert() = default;
    PayloadMetadataInsert(const PayloadMetadataInsert&) = delete;
    ~PayloadMetadataInsert() { gr4pm_payload_metadata_insert_destroy(_h); }
    void start() // :71-75
    {
        gr4pm_payload_metadata_insert_destroy(_h);
        _h = nullptr;
        gr4pm_payload_metadata_insert_params p{ syncword_size, header_size, syncword_costas_loop_bandwidth,
                                                header_costas_loop_bandwidth, payload_costas_loop_bandwidth, nullptr };
        detail::check(gr4pm_payload_metadata_insert_create(&p, &_h), "PayloadMetadataInsert::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& headerSpan, const gr::ConsumableSpan auto& inSpan,
                                 gr::PublishableSpan auto& outSpan, gr::PublishableSpan auto& ignoredSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        gr4pm_tag tag{};
        size_t n_tags = 0;
        if (this->input_tags_present()) {
            tag = detail::from_map(this->mergedInputTag().map, 0);
            if (tag.flags & GR4PM_TAG_SYNCWORD) _syncword_map = this->mergedInputTag().map;
            n_tags = 1;
        }
        std::vector<gr4pm_header_msg> msgs;
        std::vector<gr::property_map> metas;
        for (const auto& m : headerSpan) { // :207-242
            const auto& meta = m.data.value();
            gr4pm_header_msg hm{};
            hm.invalid_header = meta.contains("invalid_header") ? 1 : 0;
            if (!hm.invalid_header) hm.packet_length = pmtv::cast<uint64_t>(meta.at("packet_length"));
            msgs.push_back(hm);
            metas.push_back(meta);
        }
        const size_t n = std::min(inSpan.size(), detail::max_items()), cap = std::min(outSpan.size(), detail::max_items());
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(cap);
        _tags.resize(8);
        size_t n_out_tags = 0, consumed = 0, produced = 0, used = 0, ignored = 0;
        detail::check(gr4pm_payload_metadata_insert_process(_h, din, n, dout, cap, &tag, n_tags, msgs.data(),
                                                            msgs.size(), 0, _tags.data(), _tags.size(), &n_out_tags,
                                                            &consumed, &produced, &used, &ignored),
                      "PayloadMetadataInsert::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), produced, host_output);
        detail::consumed(hin, consumed * sizeof(c64));
        _din.done();
        size_t hdr = 0;
        for (size_t i = 0; i < n_out_tags; ++i) {
            const auto& t = _tags[i];
            gr::property_map m;
            if (t.kind == GR4PM_PKT_SYNCWORD) m = _syncword_map;                        // :104-112
            if (t.kind == GR4PM_PKT_HEADER_START) m["header_start"] = pmtv::pmt_null(); // :186-194
            if (t.kind == GR4PM_PKT_PAYLOAD) {                                          // :222-234
                while (hdr < used && metas[hdr].contains("invalid_header")) ++hdr;
                m = metas[hdr++];
                m["payload_symbols"] = pmtv::pmt(static_cast<uint64_t>(t.payload_symbols));
                m["payload_bits"] = pmtv::pmt(static_cast<uint64_t>(t.payload_bits));
            }
            if (t.constellation >= 0) m["constellation"] = std::string(constellation_name(t.constellation));
            if (t.loop_bandwidth >= 0) m["loop_bandwidth"] = t.loop_bandwidth;
            out.publishTag(m, static_cast<ssize_t>(t.index));
        }
        size_t ignored_published = 0;
        if (log && ignored > 0 && ignoredSpan.size() > 0) { // :126-147
            ignoredSpan[0] = {};
            ignored_published = 1;
        }
        if (!headerSpan.consume(used)) throw gr::exception("consume failed");
        if (!inSpan.consume(consumed)) throw gr::exception("consume failed");
        ignoredSpan.publish(ignored_published);
        outSpan.publish(produced);
        if (consumed != 0) this->_mergedInputTag.map.clear(); // :288-295
        GR4PM_TRACE_EXIT(consumed, produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- SyncwordRemove
// replaces gr::packet_modem::SyncwordRemove<T = complex<float>> (syncword_remove.hpp:11-112)
template <typename T = std::complex<float>>
class SyncwordRemove : public gr::Block<SyncwordRemove<T>>
{
    static_assert(detail::is_c64<T>, "gr4pm: SyncwordRemove is built for complex<float> items");
    gr4pm_syncword_remove* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din, _dout;

public:
    gr::PortIn<T> in;
    gr::PortOut<T> out;
    size_t syncword_size = 64;
    bool host_output = true;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    SyncwordRemove() = default;
    SyncwordRemove(const SyncwordRemove&) = delete;
    ~SyncwordRemove() { gr4pm_syncword_remove_destroy(_h); }
    void start()
    {
        gr4pm_syncword_remove_destroy(_h);
        _h = nullptr;
        gr4pm_syncword_remove_params p{ syncword_size, nullptr };
        detail::check(gr4pm_syncword_remove_create(&p, &_h), "SyncwordRemove::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        gr4pm_packet_tag tag{}, tout[2];
        size_t n_tags = 0;
        if (this->input_tags_present()) { // :51-64
            tag.index = 0;
            tag.kind = this->mergedInputTag().map.contains("syncword_amplitude") ? GR4PM_PKT_SYNCWORD
                                                                                 : GR4PM_PKT_HEADER_START;
            tag.constellation = -1;
            tag.loop_bandwidth = -1.0;
            n_tags = 1;
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        gr4pm_c64* dout = _dout.out(n);
        size_t n_out_tags = 0, produced = 0;
        detail::check(gr4pm_syncword_remove_process(_h, din, n, dout, &tag, n_tags, tout, 2, &n_out_tags, &produced),
                      "SyncwordRemove::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), produced, host_output);
        detail::consumed(hin, n * sizeof(c64));
        _din.done();
        if (n_out_tags) out.publishTag(this->mergedInputTag().map, static_cast<ssize_t>(tout[0].index)); // :59-62
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(produced);
        if (n != 0) this->_mergedInputTag.map.clear(); // :95-102
        GR4PM_TRACE_EXIT(n, produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- ConstellationLLRDecoder
// replaces gr::packet_modem::ConstellationLLRDecoder<T = float> (constellation_llr_decoder.hpp:13-142)
template <typename T = float>
class ConstellationLLRDecoder : public gr::Block<ConstellationLLRDecoder<T>, gr::Resampling<>>
{
    static_assert(std::is_same_v<T, float>, "gr4pm: ConstellationLLRDecoder is built for T = float");
    gr4pm_constellation_llr_decoder* _h = nullptr;
    detail::DeviceStage<gr4pm_c64> _din;
    detail::DeviceStage<float> _dout;

public:
    gr::PortIn<std::complex<T>> in;
    gr::PortOut<T> out;
    T noise_sigma = 1.0f;
    std::string constellation = "BPSK";
    bool host_output = true;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    ConstellationLLRDecoder() = default;
    ConstellationLLRDecoder(const ConstellationLLRDecoder&) = delete;
    ~ConstellationLLRDecoder() { gr4pm_constellation_llr_decoder_destroy(_h); }
    void settingsChanged(const gr::property_map&, const gr::property_map&) // :55-78 (also driven by tags)
    {
        const int id = detail::constellation_id(constellation);
        if (id == 0) throw gr::exception("constellation " + constellation + " not supported"); // :72-74
        this->input_chunk_size = 1;
        this->output_chunk_size = static_cast<size_t>(id); // :64-71
        gr4pm_constellation_llr_decoder_destroy(_h);
        _h = nullptr;
        gr4pm_constellation_llr_decoder_params p{ noise_sigma, id, nullptr };
        detail::check(gr4pm_constellation_llr_decoder_create(&p, &_h), "ConstellationLLRDecoder::settingsChanged");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) settingsChanged({}, {});
        if (this->input_tags_present()) {
            const auto& map = this->mergedInputTag().map;
            if (map.contains("constellation")) { // the runtime's tag -> settings update
                constellation = pmtv::cast<std::string>(map.at("constellation"));
                settingsChanged({}, {});
            }
            out.publishTag(map, 0); // :93-99
        }
        const size_t n = std::min({ inSpan.size(), outSpan.size() / this->output_chunk_size, detail::max_items() });
        const c64* hin = std::to_address(inSpan.begin());
        const gr4pm_c64* din = _din.in(hin, n);
        float* dout = _dout.out(2 * n);
        size_t produced = 0;
        detail::check(gr4pm_constellation_llr_decoder_process(_h, din, n, dout, 2 * n, nullptr, 0, nullptr, 0, nullptr,
                                                              &produced),
                      "ConstellationLLRDecoder::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), produced, host_output);
        detail::consumed(hin, n * sizeof(c64));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(produced);
        GR4PM_TRACE_EXIT(n, produced);
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- AdditiveScrambler
// replaces gr::packet_modem::AdditiveScrambler<float> / <uint8_t> (additive_scrambler.hpp:24-100)
template <typename T>
class AdditiveScrambler : public gr::Block<AdditiveScrambler<T>>
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, uint8_t>, "gr4pm: AdditiveScrambler is built for float and uint8_t");
    gr4pm_additive_scrambler* _h = nullptr;
    detail::DeviceStage<T> _din, _dout;

public:
    gr::PortIn<T> in;
    gr::PortOut<T> out;
    uint64_t mask = 0x8a, seed = 0x7f, length = 7, count = 0; // :61-64
    std::string reset_tag_key = "";
    bool host_output = true;

    AdditiveScrambler() = default;
    AdditiveScrambler(const AdditiveScrambler&) = delete;
    ~AdditiveScrambler() { gr4pm_additive_scrambler_destroy(_h); }
    void start() // :68
    {
        gr4pm_additive_scrambler_destroy(_h);
        _h = nullptr;
        gr4pm_additive_scrambler_params p{ mask, seed, length, count, std::is_same_v<T, float> ? 1 : 2, nullptr };
        detail::check(gr4pm_additive_scrambler_create(&p, &_h), "AdditiveScrambler::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        if (!_h) start();
        const uint64_t zero = 0;
        const bool reset = !reset_tag_key.empty() && this->input_tags_present() &&
                           this->mergedInputTag().map.contains(reset_tag_key); // :78-80
        const size_t n = std::min({ inSpan.size(), outSpan.size(), detail::max_items() });
        const T* hin = std::to_address(inSpan.begin());
        const T* din = _din.in(hin, n);
        T* dout = _dout.out(n);
        detail::check(gr4pm_additive_scrambler_process(_h, din, n, dout, &zero, reset ? 1 : 0),
                      "AdditiveScrambler::processBulk");
        _dout.publish(std::to_address(outSpan.begin()), n, host_output);
        detail::consumed(hin, n * sizeof(T));
        _din.done();
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        outSpan.publish(n);
        GR4PM_TRACE_EXIT(n, n);
        return gr::work::Status::OK;
    }
};

// AdditiveScrambler<Pdu<T>> (additive_scrambler.hpp:102-159; packet_transmitter_pdu.hpp:119): the LFSR restarts at
// the head of every PDU (:133-134) and after every `count` items (:137-139); tags travel unchanged (:135)
template <detail::PduLike P>
class AdditiveScrambler<P> : public gr::Block<AdditiveScrambler<P>>
{
    using T = typename P::value_type;
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, uint8_t>, "gr4pm: AdditiveScrambler<Pdu> is built for Pdu<float> and Pdu<uint8_t>");
    gr4pm_additive_scrambler* _h = nullptr;
    detail::DeviceStage<T> _din, _dout;

public:
    gr::PortIn<P> in;
    gr::PortOut<P> out;
    uint64_t mask = 0x8a, seed = 0x7f, length = 7, count = 0; // :115-118
    std::string reset_tag_key = "";

    AdditiveScrambler() = default;
    AdditiveScrambler(const AdditiveScrambler&) = delete;
    ~AdditiveScrambler() { gr4pm_additive_scrambler_destroy(_h); }
    void start() // :123
    {
        gr4pm_additive_scrambler_destroy(_h);
        _h = nullptr;
        gr4pm_additive_scrambler_params p{ mask, seed, length, count, std::is_same_v<T, float> ? 1 : 2, nullptr };
        detail::check(gr4pm_additive_scrambler_create(&p, &_h), "AdditiveScrambler<Pdu>::start");
    }
    [[nodiscard]] P processOne(const P& pdu)
    {
        if (!_h) start();
        P pdu_out = pdu; // :135
        const size_t n = pdu.data.size();
        if (n) {
            const uint64_t zero = 0; // the reset at the PDU's first item
            const T* din = _din.in(pdu.data.data(), n);
            T* dout = _dout.out(n);
            detail::check(gr4pm_additive_scrambler_process(_h, din, n, dout, &zero, 1), "AdditiveScrambler<Pdu>::processOne");
            detail::check_hip(hipMemcpy(pdu_out.data.data(), dout, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
            _din.done();
        }
        return pdu_out;
    }
};

// ---------------------------------------------------------------- ZmqPduPubSink
// replaces gr::packet_modem::ZmqPduPubSink<T> (zmq_pdu_pub_sink.hpp:11-44), the sink of PacketReceiver's symbol tap
// (packet_receiver.hpp:163-168: endpoints tcp://*:5000 / :5001).  The socket is the library's own ZMTP 3.0 PUB endpoint
// (gr4pm_zmq_pub_*): no cppzmq, no libzmq; a zmq.SUB peer (scripts/plot_symbols.py) connects as to the reference's.
template <typename T>
class ZmqPduPubSink : public gr::Block<ZmqPduPubSink<T>>
{
    gr4pm_zmq_pub* _h = nullptr;

public:
    gr::PortIn<gr::packet_modem::Pdu<T>> in;
    std::string endpoint = "tcp://*:5555"; // :26

    ZmqPduPubSink() = default;
    ZmqPduPubSink(const ZmqPduPubSink&) = delete;
    ~ZmqPduPubSink() { gr4pm_zmq_pub_destroy(_h); }
    void start() // :29 _socket.bind(endpoint)
    {
        gr4pm_zmq_pub_destroy(_h);
        _h = nullptr;
        detail::check(gr4pm_zmq_pub_create(endpoint.c_str(), &_h), "ZmqPduPubSink::start");
    }
    void stop()
    {
        gr4pm_zmq_pub_destroy(_h);
        _h = nullptr;
    }
    int port() const { return gr4pm_zmq_pub_port(_h); }                 // the bound port (endpoint "tcp://...:*")
    size_t subscribers() const { return gr4pm_zmq_pub_subscribers(_h); }
    void processOne(const gr::packet_modem::Pdu<T>& a) // :31-41: one message, the PDU's raw items
    {
        if (!_h) start();
        detail::check(gr4pm_zmq_pub_send(_h, a.data.data(), a.data.size() * sizeof(T)), "ZmqPduPubSink::processOne");
    }
};

// ---------------------------------------------------------------- HeaderPayloadSplit
// replaces gr::packet_modem::HeaderPayloadSplit<T = float> (header_payload_split.hpp:9-147): T = float is the header
// loop's split (packet_receiver.hpp:136-137), T = std::complex<float> the symbol tap's (zmq_output, :159-162)
template <typename T = float>
class HeaderPayloadSplit : public gr::Block<HeaderPayloadSplit<T>>
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, std::complex<float>>,
                  "gr4pm: HeaderPayloadSplit is built for T = float and T = std::complex<float>");
    gr4pm_header_payload_split* _h = nullptr;
    detail::DeviceStage<T> _din, _dhdr, _dpay;

public:
    gr::PortIn<T> in;
    gr::PortOut<T> header;
    gr::PortOut<T> payload;
    size_t header_size = 256;
    std::string packet_len_tag_key = "packet_len";
    std::string payload_length_key = "payload_bits";
    bool host_output = true;
    constexpr static gr::TagPropagationPolicy tag_policy = gr::TagPropagationPolicy::TPP_CUSTOM;

    HeaderPayloadSplit() = default;
    HeaderPayloadSplit(const HeaderPayloadSplit&) = delete;
    ~HeaderPayloadSplit() { gr4pm_header_payload_split_destroy(_h); }
    void start() // :41-45
    {
        gr4pm_header_payload_split_destroy(_h);
        _h = nullptr;
        gr4pm_header_payload_split_params p{ header_size, nullptr };
        detail::check(gr4pm_header_payload_split_create(&p, &_h), "HeaderPayloadSplit::start");
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& headerSpan,
                                 gr::PublishableSpan auto& payloadSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), headerSpan.size());
        if (!_h) start();
        gr4pm_packet_tag tag{}, ht[2], pt[2];
        size_t n_tags = 0;
        gr::property_map map;
        if (this->input_tags_present()) { // :68-88
            map = this->mergedInputTag().map;
            tag.kind = GR4PM_PKT_HEADER_START;
            tag.constellation = -1;
            tag.loop_bandwidth = -1.0;
            if (map.contains(payload_length_key)) {
                tag.kind = GR4PM_PKT_PAYLOAD;
                tag.payload_bits = pmtv::cast<uint64_t>(map.at(payload_length_key));
                map[packet_len_tag_key] = pmtv::pmt(static_cast<uint64_t>(tag.payload_bits)); // :81
            }
            n_tags = 1;
        }
        // one output per call, like the reference (:97-123)
        const size_t n = std::min({ inSpan.size(), headerSpan.size(), payloadSpan.size(), detail::max_items() });
        const T* hin = std::to_address(inSpan.begin());
        const T* din = _din.in(hin, n);
        T* dh = _dhdr.out(n);
        T* dp = _dpay.out(n);
        size_t nh = 0, np = 0, nht = 0, npt = 0;
        gr4pm_status st;
        if constexpr (std::is_same_v<T, float>)
            st = gr4pm_header_payload_split_process(_h, din, n, dh, &nh, dp, &np, &tag, n_tags, ht, &nht, pt, &npt, 2);
        else
            st = gr4pm_header_payload_split_process_c64(_h, reinterpret_cast<const gr4pm_c64*>(din), n,
                                                        reinterpret_cast<gr4pm_c64*>(dh), &nh,
                                                        reinterpret_cast<gr4pm_c64*>(dp), &np, &tag, n_tags, ht, &nht, pt,
                                                        &npt, 2);
        detail::check(st, "HeaderPayloadSplit::processBulk"); // the unexpected-tag exception of :75-78 included
        _dhdr.publish(std::to_address(headerSpan.begin()), nh, host_output);
        _dpay.publish(std::to_address(payloadSpan.begin()), np, host_output);
        detail::consumed(hin, n * sizeof(T));
        _din.done();
        if (nht) header.publishTag(map, 0);
        if (npt) payload.publishTag(map, 0);
        if (!inSpan.consume(n)) throw gr::exception("consume failed");
        headerSpan.publish(nh);
        payloadSpan.publish(np);
        this->_mergedInputTag.map.clear(); // :125-131
        return gr::work::Status::OK;
    }
};

// ---------------------------------------------------------------- HeaderFecDecoder
// replaces gr::packet_modem::HeaderFecDecoder (header_fec_decoder.hpp:13-359) and with it the
// reference's calls into ldpc-toolbox (:276,285,315-321)
class HeaderFecDecoder : public gr::Block<HeaderFecDecoder, gr::Resampling<1U, 64U, true>>
{
    gr4pm_header_fec_decoder* _h = nullptr;
    detail::DeviceStage<float> _din;
    std::vector<uint8_t> _bytes, _invalid;

public:
    gr::PortIn<float> in;
    gr::PortOut<uint8_t> out;
    // the reference embeds the alist text (header_fec_decoder.hpp:31-258); here it is data: this string, or,
    // when empty, the file header_ldpc_128_32.alist under $GR4PM_DATA_DIR (default: the package's data/)
    std::string alist;

    HeaderFecDecoder() = default;
    HeaderFecDecoder(const HeaderFecDecoder&) = delete;
    ~HeaderFecDecoder() { gr4pm_header_fec_decoder_destroy(_h); }
    void start() // :268-280
    {
        if (_h) throw gr::exception("an LDPC decoder already exists");
        std::string text = alist;
        if (text.empty()) {
            const char* dir = std::getenv("GR4PM_DATA_DIR");
#if defined(GR4PM_DATA_DIR_DEFAULT)
            if (!dir) dir = GR4PM_DATA_DIR_DEFAULT;
#endif
            if (!dir) throw gr::exception("HeaderFecDecoder: set `alist` or GR4PM_DATA_DIR");
            std::ifstream f(std::string(dir) + "/header_ldpc_128_32.alist");
            if (!f) throw gr::exception("HeaderFecDecoder: cannot read header_ldpc_128_32.alist");
            std::stringstream ss;
            ss << f.rdbuf();
            text = ss.str();
        }
        gr4pm_header_fec_decoder_params p{ text.c_str(), 25, nullptr };
        detail::check(gr4pm_header_fec_decoder_create(&p, &_h), "HeaderFecDecoder::start");
    }
    void stop() // :282-288
    {
        gr4pm_header_fec_decoder_destroy(_h);
        _h = nullptr;
    }
    gr::work::Status processBulk(const gr::ConsumableSpan auto& inSpan, gr::PublishableSpan auto& outSpan)
    {
        GR4PM_TRACE_ENTRY(inSpan.size(), outSpan.size());
        const size_t codewords = std::min(inSpan.size() / 256, outSpan.size() / 4); // :293-294
        if (codewords == 0) { // :296-304
            std::ignore = inSpan.consume(0);
            outSpan.publish(0);
            return inSpan.size() < 256 ? gr::work::Status::INSUFFICIENT_INPUT_ITEMS
                                       : gr::work::Status::INSUFFICIENT_OUTPUT_ITEMS;
        }
        const float* hin = std::to_address(inSpan.begin());
        const float* din = _din.in(hin, codewords * 256);
        _bytes.resize(codewords * 4);
        _invalid.resize(codewords);
        detail::check(gr4pm_header_fec_decoder_process(_h, din, codewords, _bytes.data(), _invalid.data()),
                      "HeaderFecDecoder::processBulk");
        detail::consumed(hin, codewords * 256 * sizeof(float));
        _din.done();
        std::copy(_bytes.begin(), _bytes.end(), outSpan.begin());
        for (size_t c = 0; c < codewords; ++c)
            if (_invalid[c]) out.publishTag({ { "invalid_header", pmtv::pmt_null() } }, static_cast<ssize_t>(4 * c)); // :322-326
        if (!inSpan.consume(codewords * 256)) throw gr::exception("consume failed");
        outSpan.publish(codewords * 4);
        GR4PM_TRACE_EXIT(codewords * 256, codewords * 4);
        return gr::work::Status::OK;
    }
};

// firdes::root_raised_cosine<T> (firdes.hpp:29-76): same signature, the library's design routine
namespace firdes {
template <typename T = float>
std::vector<T> root_raised_cosine(double gain, double sampling_freq, double symbol_rate, double alpha, size_t ntaps)
{
    static_assert(std::is_same_v<T, float>, "gr4pm: firdes::root_raised_cosine is built for float");
    std::vector<float> taps(ntaps | 1);
    const size_t n = gr4pm_firdes_root_raised_cosine(gain, sampling_freq, symbol_rate, alpha, ntaps, taps.data());
    if (n == 0) throw gr::exception(std::string("root_raised_cosine: ") + gr4pm_last_error());
    taps.resize(n);
    return taps;
}
} // namespace firdes

} // namespace gr::packet_modem::hip

ENABLE_REFLECTION(gr::packet_modem::hip::SyncwordDetection, in, out, fft_size, samples_per_symbol, rrc_taps,
                  syncword, constellation, min_freq_bin, max_freq_bin, time_threshold, power_threshold);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::Rotator, in, out, phase_incr);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::CoarseFrequencyCorrection, in, out, delay);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::SyncwordDetectionFilter, parsed_header, ignored_syncword, in,
                               out, samples_per_symbol, syncword_size, header_size);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::SymbolFilter, in, out, samples_per_symbol, taps, num_arms, delay);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::CostasLoop, in, out, loop_bandwidth, constellation);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::SyncwordWipeoff, in, out, syncword);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::InterpolatingFirFilter, in, out, interpolation, taps);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::PfbArbResampler, in, out, rate, taps, filter_size);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::PayloadMetadataInsert, parsed_header, in, out, ignored_syncword,
                               syncword_size, header_size, syncword_costas_loop_bandwidth,
                               header_costas_loop_bandwidth, payload_costas_loop_bandwidth, log);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::SyncwordRemove, in, out, syncword_size);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::ConstellationLLRDecoder, in, out, noise_sigma, constellation);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::AdditiveScrambler, in, out, mask, seed, length, count,
                               reset_tag_key);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::HeaderPayloadSplit, in, header, payload, header_size,
                               packet_len_tag_key, payload_length_key);
ENABLE_REFLECTION(gr::packet_modem::hip::HeaderFecDecoder, in, out);
ENABLE_REFLECTION_FOR_TEMPLATE(gr::packet_modem::hip::ZmqPduPubSink, in, endpoint);
