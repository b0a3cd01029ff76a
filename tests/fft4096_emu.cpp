// Host emulation of the 4096-point workgroup FFT (gr4-packet-modem_amd/csrc/fft4096_wg.hpp): the 256 threads run
// phase by phase on the CPU; max error of FFT(x) and of FFT(FFT(x) .* t) against a double DFT.
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>
#include "fft4096_wg.hpp"
using namespace gr4pm;
using cd = std::complex<double>;
static std::vector<cd> dft(const std::vector<cd>& x)
{
    const size_t n = x.size();
    std::vector<cd> X(n);
    for (size_t k = 0; k < n; ++k) {
        cd acc = 0;
        for (size_t i = 0; i < n; ++i) {
            const double a = -2.0 * M_PI * static_cast<double>((i * k) % n) / static_cast<double>(n);
            acc += x[i] * cd(std::cos(a), std::sin(a));
        }
        X[k] = acc;
    }
    return X;
}
static std::vector<cf> tw1(16 * 256), tw2(256);
static void fft(std::vector<std::vector<cf>>& r)
{
    std::vector<cf> lds(kX4kItems);
    for (int t = 0; t < 256; ++t) f4k_pass1(t, r[t].data(), tw1.data());
    for (int t = 0; t < 256; ++t) f4k_store1(t, r[t].data(), lds.data());
    for (int t = 0; t < 256; ++t) f4k_load2(t, r[t].data(), lds.data());
    for (int t = 0; t < 256; ++t) f4k_pass2(t, r[t].data(), tw2.data());
    for (int t = 0; t < 256; ++t) f4k_store2(t, r[t].data(), lds.data());
    for (int t = 0; t < 256; ++t) f4k_load3(t, r[t].data(), lds.data());
    for (int t = 0; t < 256; ++t) f4k_pass3(r[t].data());
}
int main()
{
    build_4096_tables(
        [](int k) {
            const double a = -2.0 * M_PI * k / kN4k;
            return mk(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        },
        tw1.data(), tw2.data());
    std::mt19937 rng(11);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<cf> x(kN4k), tp(kN4k);
    for (auto& v : x) v = mk(g(rng), g(rng));
    for (auto& v : tp) v = mk(g(rng), g(rng));
    std::vector<std::vector<cf>> r(256, std::vector<cf>(16));
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 16; ++j) r[t][j] = x[f4k_index(t, j)];
    fft(r);
    std::vector<cd> xd(kN4k);
    for (int i = 0; i < kN4k; ++i) xd[i] = cd(x[i].x, x[i].y);
    const auto X = dft(xd);
    double e1 = 0, m1 = 0;
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 16; ++j) {
            e1 = std::max(e1, std::abs(cd(r[t][j].x, r[t][j].y) - X[f4k_index(t, j)]));
            m1 = std::max(m1, std::abs(X[f4k_index(t, j)]));
        }
    std::printf("fft4096 max_rel_err %.3e\n", e1 / m1);
    std::vector<cd> pd(kN4k);
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 16; ++j) {
            const int k = f4k_index(t, j);
            pd[k] = X[k] * cd(tp[k].x, tp[k].y);
            r[t][j] = cmul(r[t][j], tp[k]);
        }
    fft(r);
    const auto C = dft(pd);
    double e2 = 0, m2 = 0;
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 16; ++j) {
            e2 = std::max(e2, std::abs(cd(r[t][j].x, r[t][j].y) - C[f4k_index(t, j)]));
            m2 = std::max(m2, std::abs(C[f4k_index(t, j)]));
        }
    std::printf("fft4096 second transform max_rel_err %.3e\n", e2 / m2);
    return (e1 / m1 < 1e-6 && e2 / m2 < 1e-6) ? 0 : 1;
}
