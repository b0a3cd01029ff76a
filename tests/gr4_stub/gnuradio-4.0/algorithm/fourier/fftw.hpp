// TEST-ONLY stand-in (see ../../Block.hpp) for gr::algorithm::FFTw as syncword_detection.hpp:6,130,196,240-252 uses it:
// compute(range, std::vector<TOut>&& out = {}) -> forward, un-normalised transform of the range (which may be a
// views::drop | views::take pipeline over the input span).  The transform itself is the CPU oracle's orc_fft
// (oracle/gr4pm_oracle.cpp), so that the reference's own detector code and the oracle's restatement of it see the same
// spectra bit for bit and can be compared exactly (tests/ref_headers_check.cpp).  FFTW itself is not in this image.
#pragma once
#include <complex>
#include <cstddef>
#include <utility>
#include <vector>

extern "C" void orc_fft(const void* in, void* out, size_t n);

namespace gr::algorithm {
template <typename TIn, typename TOut>
struct FFTw {
    static_assert(sizeof(TIn) == 8 && sizeof(TOut) == 8, "complex<float> in and out");
    template <typename R>
    std::vector<TOut> compute(const R& in, std::vector<TOut>&& out = {})
    {
        std::vector<TIn> tmp;
        for (auto&& v : in) tmp.push_back(v);
        out.resize(tmp.size());
        orc_fft(tmp.data(), out.data(), tmp.size());
        return std::move(out);
    }
};
} // namespace gr::algorithm
