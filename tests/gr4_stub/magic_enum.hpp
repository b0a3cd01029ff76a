// TEST-ONLY stand-in (see gnuradio-4.0/Block.hpp) for the magic_enum calls of the reference's blocks: enum_name(e),
// enum_cast<E>(string[, case_insensitive]) (costas_loop.hpp:50,59-61; constellation_llr_decoder.hpp:47,63-64;
// header_parser.hpp:89; packet_type_filter.hpp:33,42,77-79).  The enumerator names come from the compiler's
// __PRETTY_FUNCTION__ for the values 0 .. 31 (what the library itself does, in miniature).
#pragma once
#include <algorithm>
#include <array>
#include <cctype>
#include <optional>
#include <string>
#include <string_view>
#include <type_traits>

namespace magic_enum {
struct case_insensitive_t {};
inline constexpr case_insensitive_t case_insensitive{};

namespace detail {
template <typename E, E V>
constexpr std::string_view name_of()
{
    // gcc: "... [with E = ns::X; E V = ns::X::NAME; std::string_view = ...]", an unnamed value prints as "(ns::X)7"
    // clang: "... [E = ns::X, V = ns::X::NAME]", unnamed: "(ns::X)7"
    std::string_view s = __PRETTY_FUNCTION__;
    const size_t at = s.find("V = ");
    if (at == std::string_view::npos) return {};
    s.remove_prefix(at + 4);
    const size_t end = s.find_first_of(";]");
    if (end != std::string_view::npos) s = s.substr(0, end);
    if (s.empty() || s[0] == '(') return {};
    const size_t colon = s.rfind("::");
    if (colon != std::string_view::npos) s.remove_prefix(colon + 2);
    return s;
}
template <typename E, size_t... I>
constexpr auto names(std::index_sequence<I...>)
{
    return std::array<std::string_view, sizeof...(I)>{ name_of<E, static_cast<E>(I)>()... };
}
template <typename E>
inline constexpr auto table = names<E>(std::make_index_sequence<32>{});
inline bool same(std::string_view a, std::string_view b, bool fold)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) {
        const auto x = static_cast<unsigned char>(a[i]), y = static_cast<unsigned char>(b[i]);
        if (fold ? std::toupper(x) != std::toupper(y) : x != y) return false;
    }
    return true;
}
} // namespace detail

template <typename E>
    requires std::is_enum_v<E>
constexpr std::string_view enum_name(E e)
{
    const auto i = static_cast<size_t>(e);
    return i < detail::table<E>.size() ? detail::table<E>[i] : std::string_view{};
}
template <typename E>
std::optional<E> enum_cast(std::string_view s, case_insensitive_t)
{
    for (size_t i = 0; i < detail::table<E>.size(); ++i)
        if (!detail::table<E>[i].empty() && detail::same(detail::table<E>[i], s, true)) return static_cast<E>(i);
    return std::nullopt;
}
template <typename E>
std::optional<E> enum_cast(std::string_view s)
{
    for (size_t i = 0; i < detail::table<E>.size(); ++i)
        if (!detail::table<E>[i].empty() && detail::same(detail::table<E>[i], s, false)) return static_cast<E>(i);
    return std::nullopt;
}
} // namespace magic_enum
