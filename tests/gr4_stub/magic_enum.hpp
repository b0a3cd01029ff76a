// TEST-ONLY stand-in (see gnuradio-4.0/Block.hpp) for the two magic_enum calls of the reference's hot path
// (costas_loop.hpp:50,59-61): enum_name / enum_cast of gr::packet_modem::Constellation (constellation.hpp is
// included before this header by costas_loop.hpp).
#pragma once
#include <algorithm>
#include <cctype>
#include <optional>
#include <string>
#include <string_view>

namespace magic_enum {
struct case_insensitive_t {};
inline constexpr case_insensitive_t case_insensitive{};
inline std::string_view enum_name(gr::packet_modem::Constellation c)
{
    using C = gr::packet_modem::Constellation;
    return c == C::PILOT ? "PILOT" : c == C::BPSK ? "BPSK" : "QPSK";
}
template <typename E>
std::optional<E> enum_cast(std::string_view s, case_insensitive_t)
{
    std::string u(s);
    std::transform(u.begin(), u.end(), u.begin(), [](unsigned char ch) { return static_cast<char>(std::toupper(ch)); });
    if (u == "PILOT") return E::PILOT;
    if (u == "BPSK") return E::BPSK;
    if (u == "QPSK") return E::QPSK;
    return std::nullopt;
}
} // namespace magic_enum
