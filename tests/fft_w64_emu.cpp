// Host emulation of the one-exchange wave FFT (gr4-packet-modem_amd/csrc/fft2048_w64.hpp): runs the
// 64 lanes phase by phase on the CPU and prints the max error (relative to the largest output) of
// X = FFT(x) and of FFT(X .* t) against a double-precision DFT.  Built and run by
// tests/test_abi_and_host.py (no GPU needed).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>
#include "fft2048_w64.hpp"

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

static std::vector<f4> tT(kW64TwFloat4), tC(kW64TwFloat4);
static std::vector<cf> cc(64);
static bool interleaved = false; // which exchange layout fft_w64() runs

// the whole transform for all 64 lanes: r[lane][j] = v[lane + 64 j] in, X[lane + 64 j] out
static void fft_w64(std::vector<std::vector<cf>>& r)
{
    std::vector<float> xb(kW64BufDwords, 0.f);
    for (int l = 0; l < 64; ++l) dft32(r[l].data());
    if (interleaved) {
        cf* xc = reinterpret_cast<cf*>(xb.data());
        for (int l = 0; l < 64; ++l) w64c_store_ref(l, r[l].data(), xc);
        for (int l = 0; l < 64; ++l) w64c_mid(l, xc, tC.data(), cc[l], r[l].data());
    } else {
        for (int l = 0; l < 64; ++l) w64_store_ref(l, r[l].data(), xb.data());
        for (int l = 0; l < 64; ++l) w64_mid(l, xb.data(), tT.data(), cc[l], r[l].data());
    }
    for (int l = 0; l < 64; ++l) dft32(r[l].data());
}

// the per-bin transform as the correlator runs it: pass A interleaved, mid stage and pass B planar; returns the
// largest difference to the all-interleaved schedule relative to the largest output
static double planar_vs_interleaved(const std::vector<cf>& x)
{
    std::vector<std::vector<cf>> r(64, std::vector<cf>(32)), ref;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) r[l][j] = x[w64_index(l, j)];
    ref = r;
    interleaved = false;
    fft_w64(ref);
    std::vector<float> xb(kW64BufDwords, 0.f);
    for (int l = 0; l < 64; ++l) dft32(r[l].data());
    for (int l = 0; l < 64; ++l) w64_store_ref(l, r[l].data(), xb.data());
    double e = 0, m = 0;
    for (int l = 0; l < 64; ++l) {
        pc b[16];
        w64_mid_p(l, xb.data(), tT.data(), cc[l], b);
        dft32p(b);
        for (int k = 0; k < 16; ++k) {
            const cf lo = mk(b[k].r.x, b[k].i.x), hi = mk(b[k].r.y, b[k].i.y);
            const cf a = ref[l][k], c = ref[l][k + 16];
            e = std::max({ e, (double)std::hypot(lo.x - a.x, lo.y - a.y), (double)std::hypot(hi.x - c.x, hi.y - c.y) });
            m = std::max({ m, (double)std::hypot(a.x, a.y), (double)std::hypot(c.x, c.y) });
        }
    }
    return e / m;
}

int main()
{
    build_w64_tables(
        [](int k) {
            const double a = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        },
        tT.data(), cc.data());
    build_w64_tables_c(
        [](int k) {
            const double a = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        },
        tC.data());
    std::mt19937 rng(7);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<cf> x(kFftN), t(kFftN);
    for (auto& v : x) v = mk(g(rng), g(rng));
    for (auto& v : t) v = mk(g(rng), g(rng));

    // in-lane DFT-32 alone
    {
        std::vector<cf> v(32);
        std::vector<cd> vd(32);
        for (int i = 0; i < 32; ++i) {
            v[i] = x[i];
            vd[i] = cd(x[i].x, x[i].y);
        }
        dft32(v.data());
        const auto ref = dft(vd);
        double e = 0, m = 0;
        for (int i = 0; i < 32; ++i) {
            e = std::max(e, std::abs(cd(v[i].x, v[i].y) - ref[i]));
            m = std::max(m, std::abs(ref[i]));
        }
        std::printf("dft32 max_rel_err %.3e\n", e / m);
    }
    int rc = 0;
    {
        const double e = planar_vs_interleaved(x);
        std::printf("planar pass B vs interleaved max_rel_err %.3e\n", e);
        if (!(e < 5e-7)) rc = 1;
    }
    for (int layout = 0; layout < 2; ++layout) {
    interleaved = layout == 1;
    std::vector<std::vector<cf>> r(64, std::vector<cf>(32));
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) r[l][j] = x[w64_index(l, j)];
    fft_w64(r);
    std::vector<cd> xd(kFftN);
    for (int i = 0; i < kFftN; ++i) xd[i] = cd(x[i].x, x[i].y);
    const auto X = dft(xd);
    double e1 = 0, m1 = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const cd got(r[l][j].x, r[l][j].y);
            e1 = std::max(e1, std::abs(got - X[w64_index(l, j)]));
            m1 = std::max(m1, std::abs(X[w64_index(l, j)]));
        }
    std::printf("layout %d fft1 max_rel_err %.3e\n", layout, e1 / m1);
    // second transform of X .* t, same distribution
    std::vector<cd> pd(kFftN);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int k = w64_index(l, j);
            pd[k] = X[k] * cd(t[k].x, t[k].y);
            r[l][j] = cmul(r[l][j], t[k]);
        }
    fft_w64(r);
    const auto C = dft(pd);
    double e2 = 0, m2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const cd got(r[l][j].x, r[l][j].y);
            e2 = std::max(e2, std::abs(got - C[w64_index(l, j)]));
            m2 = std::max(m2, std::abs(C[w64_index(l, j)]));
        }
    std::printf("layout %d fft2 max_rel_err %.3e\n", layout, e2 / m2);
    if (!(e1 / m1 < 1e-6 && e2 / m2 < 1e-6)) rc = 1;
    }
    return rc;
}
