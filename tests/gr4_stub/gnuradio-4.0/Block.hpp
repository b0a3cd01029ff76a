// TEST-ONLY stand-in for the slice of the GNU Radio 4.0 block API that
// gr4-packet-modem_amd/host/gr4pm_gr4_blocks.hpp touches: it exists so that the
// wrapper header goes through a compiler and its processBulk() can be driven on a GPU box where
// gnuradio4 is not installed.  It is NOT an oracle and NOT part of the product: it checks OUR header.
// Written from the API surface the reference blocks use (syncword_detection.hpp:4-7,143-356;
// symbol_filter.hpp:112-252; costas_loop.hpp:92-148), not from gnuradio4 sources.
#pragma once
#include <sys/types.h>

#include <complex>
#include <concepts>
#include <cstdint>
#include <map>
#include <optional>
#include <span>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

namespace pmtv {
using pmt = std::variant<std::monostate, bool, int32_t, int64_t, uint64_t, float, double, std::string,
                         std::vector<float>, std::vector<uint8_t>, std::vector<std::complex<float>>>;
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

namespace gr {

struct exception : std::runtime_error {
    using std::runtime_error::runtime_error;
};
using property_map = std::map<std::string, pmtv::pmt, std::less<>>;
struct Tag {
    ssize_t index = 0;
    property_map map;
};
struct Message {
    std::optional<property_map> data;
};
struct Async {};
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

template <typename T, typename... Attr>
struct PortIn {
    using value_type = T;
    size_t min_samples = 1, max_samples = static_cast<size_t>(-1);
};
template <typename T, typename... Attr>
struct PortOut {
    using value_type = T;
    size_t min_samples = 1, max_samples = static_cast<size_t>(-1);
    // what the block published during the current processBulk(): offsets are relative to the out span
    std::vector<Tag> published_tags;
    void publishTag(const property_map& map, ssize_t offset) { published_tags.push_back({ offset, map }); }
};

template <typename Derived, typename... Attr>
struct Block {
    std::string name = "block";
    size_t input_chunk_size = 1, output_chunk_size = 1;
    Tag _mergedInputTag;
    bool input_tags_present() const { return !_mergedInputTag.map.empty(); }
    const Tag& mergedInputTag() const { return _mergedInputTag; }
    template <typename>
    struct DocTag {};
};

} // namespace gr

#define ENABLE_REFLECTION(...) static_assert(true)
#define ENABLE_REFLECTION_FOR_TEMPLATE(...) static_assert(true)
#define ENABLE_REFLECTION_FOR_TEMPLATE_FULL(...) static_assert(true)
