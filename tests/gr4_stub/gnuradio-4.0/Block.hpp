// TEST-ONLY stand-in for the slice of the GNU Radio 4.0 block API that
// gr4-packet-modem_amd/host/gr4pm_gr4_blocks.hpp touches: it exists so that the
// wrapper header goes through a compiler and its processBulk() can be driven on a GPU box where
// gnuradio4 is not installed.  It is NOT an oracle and NOT part of the product: it checks OUR header.
// Written from the API surface the reference blocks use (syncword_detection.hpp:4-7,143-356;
// symbol_filter.hpp:112-252; costas_loop.hpp:92-148), not from gnuradio4 sources.
//
// Checked in both directions (tests/test_gr4_blocks.py): the wrappers compile and run against it, and -- in the
// build container, where /root/reference exists -- so do the reference's OWN block headers (rotator,
// coarse_frequency_correction, symbol_filter, costas_loop, interpolating_fir_filter, pfb_arb_resampler,
// syncword_wipeoff; tests/ref_headers_check.cpp), i.e. the surface declared here is the one the reference uses.
// Settings by name: ENABLE_REFLECTION records the member list, gr::stub::Graph::emplaceBlock<T>(property_map)
// initialises a block from a property_map exactly as the reference's flowgraphs do (packet_receiver.hpp:76-127).
#pragma once
#include <sys/types.h>

#include <algorithm>
#include <bit>
#include <cassert>
#include <cmath>
#include <complex>
#include <memory>
#include <numbers>
#include <numeric>
#include <ranges>
#include <concepts>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <functional>
#include <cstdint>
#include <map>
#include <optional>
#include <span>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <typeinfo>
#include <variant>
#include <vector>

// {fmt} comes with gnuradio4's Block.hpp; the reference uses fmt::format for exception texts, fmt::println under
// #ifdef TRACE (syncword_detection_filter.hpp:117, payload_metadata_insert.hpp:115, ...) and for the reports of its
// apps (message_debug.hpp:39, benchmark_syncword_detection.cpp:18-22,92).  libstdc++ 11 has no <format>: a miniature
// that substitutes "{}" / "{:spec}" (the spec is ignored) with the streamed argument.
namespace fmt {
namespace stub_detail {
template <typename T>
void put(std::ostream& os, const T& v);
template <typename T>
concept Streamable = requires(std::ostream& os, const T& v) { os << v; };
template <typename T>
concept MapLike = requires(const T& m) { m.begin()->first; m.begin()->second; };
template <typename T>
concept RangeLike = requires(const T& r) { r.begin(); r.end(); } && !Streamable<T> && !MapLike<T>;
template <typename T>
concept VariantLike = requires(const T& v) { v.index(); std::variant_size<T>::value; };
template <typename T>
concept OptionalLike = requires(const T& o) { o.has_value(); *o; } && !VariantLike<T>;
template <typename T>
void put(std::ostream& os, const T& v)
{
    if constexpr (std::is_same_v<T, std::monostate>) {
        os << "null";
    } else if constexpr (std::is_same_v<T, bool>) {
        os << (v ? "true" : "false");
    } else if constexpr (std::is_same_v<T, uint8_t> || std::is_same_v<T, int8_t>) {
        os << static_cast<int>(v);
    } else if constexpr (Streamable<T>) {
        os << v;
    } else if constexpr (MapLike<T>) {
        os << "{";
        bool first = true;
        for (const auto& kv : v) {
            os << (first ? "" : ", ");
            put(os, kv.first);
            os << ": ";
            put(os, kv.second);
            first = false;
        }
        os << "}";
    } else if constexpr (VariantLike<T>) {
        std::visit([&](const auto& x) { put(os, x); }, v);
    } else if constexpr (OptionalLike<T>) {
        if (v.has_value()) put(os, *v);
        else os << "(none)";
    } else if constexpr (RangeLike<T>) {
        os << "[";
        size_t n = 0;
        for (const auto& x : v) {
            if (n == 8) {
                os << ", ...";
                break;
            }
            os << (n++ ? ", " : "");
            put(os, x);
        }
        os << "]";
    } else {
        os << "<?>";
    }
}
inline void next_field(std::ostream& os, std::string_view& f)
{
    // copies up to the next replacement field and skips the field; "{{" / "}}" are literal braces
    while (!f.empty()) {
        if (f.size() >= 2 && (f.substr(0, 2) == "{{" || f.substr(0, 2) == "}}")) {
            os << f[0];
            f.remove_prefix(2);
        } else if (f[0] == '{') {
            const size_t e = f.find('}');
            f.remove_prefix(e == std::string_view::npos ? f.size() : e + 1);
            return;
        } else {
            os << f[0];
            f.remove_prefix(1);
        }
    }
}
template <typename... A>
std::string vformat(std::string_view f, const A&... a)
{
    std::ostringstream os;
    ((next_field(os, f), put(os, a)), ...);
    next_field(os, f);
    os << f;
    return os.str();
}
} // namespace stub_detail
template <typename... A>
std::string format(std::string_view f, const A&... a)
{
    return stub_detail::vformat(f, a...);
}
template <typename... A>
void println(std::FILE* to, std::string_view f, const A&... a)
{
    const std::string s = stub_detail::vformat(f, a...) + "\n";
    std::fwrite(s.data(), 1, s.size(), to);
}
template <typename... A>
void println(std::string_view f, const A&... a)
{
    println(stdout, f, a...);
}
template <typename... A>
void print(std::string_view f, const A&... a)
{
    const std::string s = stub_detail::vformat(f, a...);
    std::fwrite(s.data(), 1, s.size(), stdout);
}
} // namespace fmt

// (magic_enum reaches the reference's blocks through gnuradio4's Block.hpp: header_parser.hpp:89 uses it without an include)
#include <magic_enum.hpp>

namespace pmtv {
using pmt = std::variant<std::monostate, bool, int32_t, int64_t, uint64_t, float, double, std::string,
                         std::vector<float>, std::vector<uint8_t>, std::vector<std::complex<float>>, std::vector<double>>;
inline pmt pmt_null() { return pmt{}; }
template <typename T>
T cast(const pmt& p)
{
    return std::visit(
        [](const auto& v) -> T {
            using V = std::decay_t<decltype(v)>;
            if constexpr (std::is_convertible_v<V, T> && !std::is_same_v<V, std::monostate>)
                return static_cast<T>(v);
            else
                throw std::runtime_error("pmtv::cast: incompatible type");
        },
        p);
}
} // namespace pmtv

// std::views::repeat (C++23) is not in this image's libstdc++: the reference fills its history buffers with it
// (symbol_filter.hpp:97, interpolating_fir_filter.hpp:67, pfb_arb_resampler.hpp:108)
#if !defined(__cpp_lib_ranges_repeat)
namespace std::ranges::views {
struct gr4_stub_repeat_fn {
    template <typename T>
    std::vector<T> operator()(const T& v, size_t n) const
    {
        return std::vector<T>(n, v);
    }
};
inline constexpr gr4_stub_repeat_fn repeat{};
} // namespace std::ranges::views
#endif

namespace gr {
namespace meta {
template <size_t N>
struct fixed_string {
    char data[N]{};
    constexpr fixed_string(const char (&s)[N]) { std::copy_n(s, N, data); }
};
} // namespace meta
// `using Description = Doc<R""(...)"">;` inside every block
template <meta::fixed_string>
struct Doc {};

struct exception : std::runtime_error {
    using std::runtime_error::runtime_error;
};
using property_map = std::map<std::string, pmtv::pmt, std::less<>>;
struct Tag {
    ssize_t index = 0;
    property_map map;
};
// Message::data is a std::expected<property_map, Error> in gnuradio4 (message_debug.hpp:37-41 reads .has_value() /
// .value() / .error()); libstdc++ 11 has no <expected>: an optional with an error text
struct MessageData : std::optional<property_map> {
    using std::optional<property_map>::optional;
    using std::optional<property_map>::operator=;
    std::string error() const { return "no data"; }
};
struct Message {
    MessageData data;
};
namespace message {
enum class Command { Set, Get, Subscribe, Unsubscribe, Partial, Final, Ready, Disconnect, Heartbeat, Invalid };
}
enum class ConnectionResult { SUCCESS, FAILED };
struct Async {};
template <bool>
struct BlockingIO {}; // block attribute (tun_source.hpp:15)
template <size_t, size_t, bool = false>
struct RequiredSamples {}; // port attribute (tun_source.hpp:33)
struct Optional {}; // port attribute: the port may stay unconnected (packet_to_stream.hpp:57,159, pdu_to_tagged_stream.hpp:36)
template <auto...>
struct Resampling {};
enum class TagPropagationPolicy { TPP_DONT, TPP_ALL_TO_ALL, TPP_ONE_TO_ONE, TPP_CUSTOM };
namespace work {
enum class Status { ERROR = -100, INSUFFICIENT_OUTPUT_ITEMS = -3, INSUFFICIENT_INPUT_ITEMS = -2, DONE = -1, OK = 0 };
}

// spans as the scheduler hands them to processBulk(): a view plus consume() / publish()
template <typename T>
struct InSpan : std::span<const T> {
    mutable size_t consumed = 0;
    mutable bool consume_called = false;
    InSpan(const T* p, size_t n) : std::span<const T>(p, n) {}
    bool consume(size_t n) const
    {
        if (n > this->size()) return false;
        consumed = n;
        consume_called = true;
        return true;
    }
};
template <typename T>
struct OutSpan : std::span<T> {
    size_t published = 0;
    bool publish_called = false;
    OutSpan(T* p, size_t n) : std::span<T>(p, n) {}
    void publish(size_t n)
    {
        if (n > this->size()) throw exception("publish beyond the span");
        published = n;
        publish_called = true;
    }
};
template <typename S>
concept ConsumableSpan = requires(const S& s) {
    { s.size() } -> std::convertible_to<size_t>;
    { s.consume(size_t{}) } -> std::same_as<bool>;
    s.begin();
};
template <typename S>
concept PublishableSpan = requires(S& s) {
    { s.size() } -> std::convertible_to<size_t>;
    s.publish(size_t{});
    s.begin();
};

namespace stub {
// what Graph::connect() records on the two ports of an edge (gnuradio-4.0/Graph.hpp of the stand-in)
struct Link {
    const void* block = nullptr; // the block at the other end
    std::string port;            // its port name
};
} // namespace stub
template <typename T, typename... Attr>
struct PortIn {
    using value_type = T;
    static constexpr bool is_input = true;
    static constexpr bool is_async = (std::is_same_v<Attr, Async> || ... || false);
    static constexpr bool is_optional = (std::is_same_v<Attr, Optional> || ... || false);
    size_t min_samples = 1, max_samples = static_cast<size_t>(-1);
    std::vector<stub::Link> links;
    size_t buffer_size = 65536; // resizeBuffer(): packet_transmitter_pdu.hpp:53-220 sizes its PDU edges
    ConnectionResult resizeBuffer(size_t n)
    {
        buffer_size = n;
        return ConnectionResult::SUCCESS;
    }
};
template <typename T, typename... Attr>
struct PortOut {
    using value_type = T;
    static constexpr bool is_input = false;
    std::vector<stub::Link> links;
    static constexpr bool is_async = (std::is_same_v<Attr, Async> || ... || false);
    static constexpr bool is_optional = (std::is_same_v<Attr, Optional> || ... || false);
    size_t min_samples = 1, max_samples = static_cast<size_t>(-1);
    size_t buffer_size = 65536;
    ConnectionResult resizeBuffer(size_t n)
    {
        buffer_size = n;
        return ConnectionResult::SUCCESS;
    }
    // what the block published during the current processBulk(): offsets are relative to the out span
    std::vector<Tag> published_tags;
    void publishTag(const property_map& map, ssize_t offset) { published_tags.push_back({ offset, map }); }
};

// built-in message ports (probe_rate.hpp:97, message_debug.hpp:30-31): messages, no samples
struct MsgPortOut {
    using value_type = Message;
    static constexpr bool is_input = false, is_async = true;
    size_t min_samples = 0, max_samples = static_cast<size_t>(-1);
    std::vector<stub::Link> links;
    std::vector<Message> sent; // what sendMessage() left here
};
template <meta::fixed_string Name>
struct MsgPortInNamed {
    using value_type = Message;
    static constexpr bool is_input = true, is_async = true;
    size_t min_samples = 0, max_samples = static_cast<size_t>(-1);
    std::vector<stub::Link> links;
};
using MsgPortIn = MsgPortInNamed<"msgIn">;
template <message::Command>
void sendMessage(MsgPortOut& port, std::string_view /*service*/, std::string_view /*endpoint*/, property_map data)
{
    Message m;
    m.data = std::move(data);
    port.sent.push_back(std::move(m));
}

template <typename Derived, typename... Attr>
struct Block {
    bool stop_requested = false;
    std::vector<std::pair<std::string, std::string>> error_messages;
    void requestStop() { stop_requested = true; }
    void emitErrorMessage(std::string_view where, std::string_view what) { error_messages.emplace_back(where, what); }
    std::string name = "block";
    size_t input_chunk_size = 1, output_chunk_size = 1;
    Tag _mergedInputTag;
    bool input_tags_present() const { return !_mergedInputTag.map.empty(); }
    const Tag& mergedInputTag() const { return _mergedInputTag; }
    template <typename>
    struct DocTag {};
};

} // namespace gr

// ---- settings by name ------------------------------------------------------------------------------------------
namespace gr::stub {
template <typename M>
concept PortLike = requires { typename M::value_type; } && requires(M m) { m.min_samples; };
template <typename T>
struct is_vector : std::false_type {};
template <typename T, typename A>
struct is_vector<std::vector<T, A>> : std::true_type {};

// property_map value -> setting member, with the conversions pmtv allows (arithmetic <-> arithmetic, same vectors)
template <typename M>
void assign(M& member, const pmtv::pmt& v, const char* name)
{
    if constexpr (PortLike<M>) {
        throw exception(std::string("'") + name + "' is a port, not a setting");
    } else if constexpr (std::is_arithmetic_v<M>) {
        member = pmtv::cast<M>(v);
    } else if constexpr (std::is_same_v<M, std::string>) {
        if (!std::holds_alternative<std::string>(v)) throw exception(std::string("setting '") + name + "' wants a string");
        member = std::get<std::string>(v);
    } else if constexpr (is_vector<M>::value) {
        std::visit(
            [&](const auto& src) {
                using S = std::decay_t<decltype(src)>;
                if constexpr (std::is_same_v<S, M>) {
                    member = src;
                } else if constexpr (is_vector<S>::value) {
                    if constexpr (std::is_convertible_v<typename S::value_type, typename M::value_type> &&
                                  std::is_arithmetic_v<typename S::value_type>)
                        member.assign(src.begin(), src.end());
                    else
                        throw exception(std::string("setting '") + name + "': vector of another item type");
                } else {
                    throw exception(std::string("setting '") + name + "' wants a vector");
                }
            },
            v);
    } else {
        throw exception(std::string("setting '") + name + "': type not supported by the test stand-in");
    }
}
template <size_t N>
constexpr bool same_name(const meta::fixed_string<N>& a, const char* b)
{
    for (size_t i = 0; i < N; ++i) {
        if (a.data[i] != b[i]) return false;
        if (b[i] == 0) return true;
    }
    return false;
}
template <meta::fixed_string>
inline constexpr bool no_such_member = false;
// a port found by its RUN-TIME name -- fg.connect(a, "out"s, b, "in#2"s), packet_transmitter_pdu.hpp:273-401: "name" is a
// port member, "name#k" element k of a std::vector of ports (packet_mux.hpp:35,212)
struct PortRef {
    const std::type_info* item = nullptr;
    bool is_input = false;
    std::vector<Link>* links = nullptr;
};
template <typename M>
bool port_named(M& m, std::string_view member, std::string_view wanted, PortRef& ref)
{
    if constexpr (PortLike<M>) {
        if (wanted != member) return false;
        ref = { &typeid(typename M::value_type), M::is_input, &m.links };
        return true;
    } else if constexpr (requires { typename M::value_type; requires PortLike<typename M::value_type>; m.size(); }) {
        if (wanted.size() < member.size() + 2 || wanted.substr(0, member.size()) != member || wanted[member.size()] != '#')
            return false;
        const size_t k = std::stoul(std::string(wanted.substr(member.size() + 1)));
        if (k >= m.size()) return false;
        using P = typename M::value_type;
        ref = { &typeid(typename P::value_type), P::is_input, &m[k].links };
        return true;
    } else {
        return false;
    }
}
template <typename T>
struct Reflect; // specialised by ENABLE_REFLECTION*: size_t apply(T&, const property_map&) -> settings assigned

// what `fg.emplaceBlock<T>({ { "key", value }, ... })` does with the initial settings
struct Graph {
    std::vector<std::shared_ptr<void>> blocks;
    std::vector<std::function<void()>> starters, stoppers; // start() / stop() of the blocks that have them
    size_t needed_a_device = 0;                            // settingsChanged() calls that asked for the GPU (see below)
    template <typename T>
    T& emplaceBlock(property_map settings = {})
    {
        auto p = std::make_shared<T>();
        const size_t n = Reflect<T>::apply(*p, settings);
        if (n != settings.size()) {
            std::string unknown;
            for (const auto& kv : settings) {
                property_map one{ kv };
                T probe;
                if (Reflect<T>::apply(probe, one) == 0) unknown += " " + kv.first;
            }
            throw exception("emplaceBlock: no such setting:" + unknown);
        }
        if constexpr (requires { p->settingsChanged(settings, settings); }) {
            try {
                p->settingsChanged({}, settings);
            } catch (const exception& e) {
                // GR4_STUB_LIFECYCLE=0 (a machine without a GPU: the drop-in blocks have no CPU fallback and say so from
                // the first call that needs the device): note it and go on building the graph
                const char* lc = std::getenv("GR4_STUB_LIFECYCLE");
                if (!(lc && lc[0] == '0') || std::string_view(e.what()).find("no HIP device") == std::string_view::npos) throw;
                ++needed_a_device;
            }
        }
        if constexpr (requires { p->start(); }) starters.push_back([q = p.get()] { q->start(); });
        if constexpr (requires { p->stop(); }) stoppers.push_back([q = p.get()] { q->stop(); });
        blocks.push_back(p);
        return *p;
    }
};
} // namespace gr::stub

#define GR4_STUB_EXPAND(x) x
#define GR4_STUB_FE_1(F, a) F(a)
#define GR4_STUB_FE_2(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_1(F, __VA_ARGS__))
#define GR4_STUB_FE_3(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_2(F, __VA_ARGS__))
#define GR4_STUB_FE_4(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_3(F, __VA_ARGS__))
#define GR4_STUB_FE_5(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_4(F, __VA_ARGS__))
#define GR4_STUB_FE_6(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_5(F, __VA_ARGS__))
#define GR4_STUB_FE_7(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_6(F, __VA_ARGS__))
#define GR4_STUB_FE_8(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_7(F, __VA_ARGS__))
#define GR4_STUB_FE_9(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_8(F, __VA_ARGS__))
#define GR4_STUB_FE_10(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_9(F, __VA_ARGS__))
#define GR4_STUB_FE_11(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_10(F, __VA_ARGS__))
#define GR4_STUB_FE_12(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_11(F, __VA_ARGS__))
#define GR4_STUB_FE_13(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_12(F, __VA_ARGS__))
#define GR4_STUB_FE_14(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_13(F, __VA_ARGS__))
#define GR4_STUB_FE_15(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_14(F, __VA_ARGS__))
#define GR4_STUB_FE_16(F, a, ...) F(a) GR4_STUB_EXPAND(GR4_STUB_FE_15(F, __VA_ARGS__))
#define GR4_STUB_PICK(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, _11, _12, _13, _14, _15, _16, N, ...) N
#define GR4_STUB_FOR_EACH(F, ...)                                                                                    \
    GR4_STUB_EXPAND(GR4_STUB_PICK(__VA_ARGS__, GR4_STUB_FE_16, GR4_STUB_FE_15, GR4_STUB_FE_14, GR4_STUB_FE_13,        \
                                  GR4_STUB_FE_12, GR4_STUB_FE_11, GR4_STUB_FE_10, GR4_STUB_FE_9, GR4_STUB_FE_8,       \
                                  GR4_STUB_FE_7, GR4_STUB_FE_6, GR4_STUB_FE_5, GR4_STUB_FE_4, GR4_STUB_FE_3,          \
                                  GR4_STUB_FE_2, GR4_STUB_FE_1)(F, __VA_ARGS__))
#define GR4_STUB_APPLY_ONE(member)                                                                                   \
    if (auto it_ = m_.find(#member); it_ != m_.end()) {                                                              \
        ::gr::stub::assign(b_.member, it_->second, #member);                                                         \
        ++n_;                                                                                                        \
    }
#define GR4_STUB_REFLECT_BODY(...)                                                                                   \
    {                                                                                                                \
        size_t n_ = 0;                                                                                               \
        GR4_STUB_FOR_EACH(GR4_STUB_APPLY_ONE, __VA_ARGS__)                                                           \
        return n_;                                                                                                   \
    }
// member by compile-time name: fg.connect<"out">(a) (gnuradio-4.0/Graph.hpp of the stand-in)
#define GR4_STUB_MEMBER_ONE(member)                                                                                  \
    if constexpr (::gr::stub::same_name(N_, #member)) return (b_.member);                                            \
    else
#define GR4_STUB_MEMBER_BODY(...)                                                                                    \
    {                                                                                                                \
        GR4_STUB_FOR_EACH(GR4_STUB_MEMBER_ONE, __VA_ARGS__)                                                          \
        static_assert(::gr::stub::no_such_member<N_>, "no reflected member of that name");                           \
    }
#define GR4_STUB_PORT_ONE(member)                                                                                    \
    if (::gr::stub::port_named(b_.member, #member, name_, ref_)) return true;
#define GR4_STUB_PORT_BODY(...)                                                                                      \
    {                                                                                                                \
        GR4_STUB_FOR_EACH(GR4_STUB_PORT_ONE, __VA_ARGS__)                                                            \
        return false;                                                                                                \
    }
#define GR4_STUB_STRIP(...) __VA_ARGS__
#define ENABLE_REFLECTION(Type, ...)                                                                                 \
    template <>                                                                                                      \
    struct gr::stub::Reflect<Type> {                                                                                 \
        static size_t apply(Type& b_, const ::gr::property_map& m_) GR4_STUB_REFLECT_BODY(__VA_ARGS__)                \
        template <::gr::meta::fixed_string N_>                                                                       \
        static decltype(auto) member(Type& b_) GR4_STUB_MEMBER_BODY(__VA_ARGS__)                                     \
        static bool port(Type& b_, std::string_view name_, ::gr::stub::PortRef& ref_) GR4_STUB_PORT_BODY(__VA_ARGS__)\
    }
#define ENABLE_REFLECTION_FOR_TEMPLATE(Tmpl, ...)                                                                    \
    template <typename... Ts_>                                                                                       \
    struct gr::stub::Reflect<Tmpl<Ts_...>> {                                                                         \
        static size_t apply(Tmpl<Ts_...>& b_, const ::gr::property_map& m_) GR4_STUB_REFLECT_BODY(__VA_ARGS__)        \
        template <::gr::meta::fixed_string N_>                                                                       \
        static decltype(auto) member(Tmpl<Ts_...>& b_) GR4_STUB_MEMBER_BODY(__VA_ARGS__)                             \
        static bool port(Tmpl<Ts_...>& b_, std::string_view name_, ::gr::stub::PortRef& ref_) GR4_STUB_PORT_BODY(__VA_ARGS__)\
    }
// ENABLE_REFLECTION_FOR_TEMPLATE_FULL((bool invert, typename TIn, typename TOut), (BinarySlicer<invert, TIn, TOut>), in, out)
#define ENABLE_REFLECTION_FOR_TEMPLATE_FULL(TParams, Type, ...)                                                      \
    template <GR4_STUB_STRIP TParams>                                                                                \
    struct gr::stub::Reflect<GR4_STUB_STRIP Type> {                                                                  \
        using Self_ = GR4_STUB_STRIP Type;                                                                           \
        static size_t apply(Self_& b_, const ::gr::property_map& m_) GR4_STUB_REFLECT_BODY(__VA_ARGS__)               \
        template <::gr::meta::fixed_string N_>                                                                       \
        static decltype(auto) member(Self_& b_) GR4_STUB_MEMBER_BODY(__VA_ARGS__)                                    \
        static bool port(Self_& b_, std::string_view name_, ::gr::stub::PortRef& ref_) GR4_STUB_PORT_BODY(__VA_ARGS__)\
    }
