// Host emulation of the one-wave FFT schedules (gr4-packet-modem_amd/csrc/fft2048_wave.hpp):
// runs the 64 lanes phase by phase on the CPU and prints the max relative error of FFT-1 and
// of FFT-2(FFT-1(x) .* t) against a double-precision DFT.  Built and run by
// tests/test_fft_wave_emulation.py (no GPU needed).
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>
#include "fft2048_wave.hpp"
#include "fft2048_pair.hpp"

using namespace gr4pm;
using cd = std::complex<double>;

static std::vector<cd> dft(const std::vector<cd>& x)
{
    const size_t n = x.size();
    std::vector<cd> X(n);
    // O(n^2) with exact-angle reduction
    for (size_t k = 0; k < n; ++k) {
        cd acc = 0;
        for (size_t i = 0; i < n; ++i) {
            const size_t ph = (i * k) % n;
            const double a = -2.0 * M_PI * static_cast<double>(ph) / static_cast<double>(n);
            acc += x[i] * cd(std::cos(a), std::sin(a));
        }
        X[k] = acc;
    }
    return X;
}

int main()
{
    std::vector<cf> tw1a(kTw1aItems), tw1b(kTw1bItems), twA(kTwAItems), twB(kTwBItems);
    build_twiddle_tables(
        [](int k) {
            const double a = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        },
        tw1a.data(), tw1b.data(), twA.data(), twB.data());
    std::mt19937 rng(42);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<cf> x(kFftN), t(kFftN);
    for (auto& v : x) v = mk(g(rng), g(rng));
    for (auto& v : t) v = mk(g(rng), g(rng));

    std::vector<cf> lds(kExchangeItems);
    std::vector<std::vector<cf>> r(kLanes, std::vector<cf>(kPtsPerLane));
    // ---- FFT-1
    for (int l = 0; l < kLanes; ++l)
        for (int n1 = 0; n1 < 16; ++n1)
            for (int e = 0; e < 2; ++e) r[l][2 * n1 + e] = x[2 * l + e + 128 * n1];
    // every phase runs for all 64 lanes before the next one starts (what the in-order LDS
    // pipeline of one wave guarantees on the GPU); exchanges run in two half-rounds
    for (int l = 0; l < kLanes; ++l) fft1_pass1(l, r[l].data(), tw1a.data());
    for (int h = 0; h < 2; ++h) {
        for (int l = 0; l < kLanes; ++l) fft1_store1(l, r[l].data(), lds.data(), h);
        for (int l = 0; l < kLanes; ++l) fft1_load2(l, r[l].data(), lds.data(), h);
    }
    for (int l = 0; l < kLanes; ++l) fft1_pass2(l, r[l].data(), tw1b.data());
    for (int h = 0; h < 2; ++h) {
        for (int l = 0; l < kLanes; ++l) fft1_store2(l, r[l].data(), lds.data(), h);
        for (int l = 0; l < kLanes; ++l) fft1_load3(l, r[l].data(), lds.data(), h);
    }
    for (int l = 0; l < kLanes; ++l) fft1_pass3(r[l].data());

    std::vector<cd> xd(kFftN);
    for (int i = 0; i < kFftN; ++i) xd[i] = cd(x[i].x, x[i].y);
    const auto X = dft(xd);
    double maxref = 0, maxerr = 0;
    std::vector<int> seen(kFftN, 0);
    for (int l = 0; l < kLanes; ++l)
        for (int j = 0; j < kPtsPerLane; ++j) {
            const int k = fft1_out_index(l, j);
            seen[k]++;
            maxref = std::max(maxref, std::abs(X[k]));
            maxerr = std::max(maxerr, std::abs(cd(r[l][j].x, r[l][j].y) - X[k]));
        }
    int bad = 0;
    for (int k = 0; k < kFftN; ++k) bad += seen[k] != 1;
    std::printf("fft1 relerr %.3e coverage_bad %d\n", maxerr / maxref, bad);

    // ---- product in FFT-1's distribution, then FFT-2
    std::vector<cd> P(kFftN);
    for (int l = 0; l < kLanes; ++l)
        for (int j = 0; j < kPtsPerLane; ++j) {
            const int k = fft1_out_index(l, j);
            r[l][j] = cmul(r[l][j], t[k]);
            P[k] = cd(r[l][j].x, r[l][j].y);
        }
    for (int l = 0; l < kLanes; ++l) fft2_passA(l, r[l].data(), twA.data());
    {
        // loadB overwrites r[16h..] while storeA of the second half still needs r[8q+4..8q+7]:
        // the kernel keeps the pass-A results in a second register set, emulate that
        auto ra = r;
        for (int h = 0; h < 2; ++h) {
            for (int l = 0; l < kLanes; ++l) fft2_storeA(l, ra[l].data(), lds.data(), h);
            for (int l = 0; l < kLanes; ++l) fft2_loadB(l, r[l].data(), lds.data(), h);
        }
    }
    for (int l = 0; l < kLanes; ++l) fft2_passB(l, r[l].data(), twB.data());
    {
        auto rb = r;
        for (int h = 0; h < 2; ++h) {
            for (int l = 0; l < kLanes; ++l) fft2_storeB(l, rb[l].data(), lds.data(), h);
            for (int l = 0; l < kLanes; ++l) fft2_loadC(l, r[l].data(), lds.data(), h);
        }
    }
    for (int l = 0; l < kLanes; ++l) fft2_passC(r[l].data());
    const auto C = dft(P);
    maxref = maxerr = 0;
    std::fill(seen.begin(), seen.end(), 0);
    for (int l = 0; l < kLanes; ++l)
        for (int j = 0; j < kPtsPerLane; ++j) {
            const int k = fft2_out_index(l, j);
            seen[k]++;
            maxref = std::max(maxref, std::abs(C[k]));
            maxerr = std::max(maxerr, std::abs(cd(r[l][j].x, r[l][j].y) - C[k]));
        }
    bad = 0;
    for (int k = 0; k < kFftN; ++k) bad += seen[k] != 1;
    std::printf("fft2 relerr %.3e coverage_bad %d\n", maxerr / maxref, bad);

    // ================= the same two transforms by a PAIR of waves (fft2048_pair.hpp): 128 lanes,
    // 16 points each; must give bit-identical values (same arithmetic, other distribution)
    std::vector<cf> one_wave_C(kFftN);
    for (int l = 0; l < kLanes; ++l)
        for (int j = 0; j < kPtsPerLane; ++j) one_wave_C[fft2_out_index(l, j)] = r[l][j];
    std::vector<cf> tw1p(kTw1pItems), twAp(kTwApItems);
    build_pair_twiddle_tables(
        [](int k) {
            const double a = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        },
        tw1p.data(), twAp.data());
    std::vector<std::vector<cf>> rp(kPairLanes, std::vector<cf>(kPairPts)), bp = rp, cp = rp;
    for (int L = 0; L < kPairLanes; ++L)
        for (int n1 = 0; n1 < 16; ++n1) rp[L][n1] = x[L + 128 * n1];
    for (int L = 0; L < kPairLanes; ++L) fft1p_pass1(L, rp[L].data(), tw1p.data());
    for (int h = 0; h < 2; ++h) { // barrier between the store and the load phase on the GPU
        for (int L = 0; L < kPairLanes; ++L) fft1p_store1(L, rp[L].data(), lds.data(), h);
        for (int L = 64 * h; L < 64 * h + 64; ++L) fft1p_load2(L, bp[L].data(), lds.data());
    }
    for (int L = 0; L < kPairLanes; ++L) fft1p_pass2(L, bp[L].data(), tw1b.data());
    for (int h = 0; h < 2; ++h) {
        for (int L = 64 * h; L < 64 * h + 64; ++L) fft1p_store2(L, bp[L].data(), lds.data());
        for (int L = 0; L < kPairLanes; ++L) fft1p_load3(L, rp[L].data(), lds.data(), h);
    }
    for (int L = 0; L < kPairLanes; ++L) fft1p_pass3(rp[L].data());
    maxref = maxerr = 0;
    std::fill(seen.begin(), seen.end(), 0);
    for (int L = 0; L < kPairLanes; ++L)
        for (int j = 0; j < kPairPts; ++j) {
            const int k = fft1p_out_index(L, j);
            seen[k]++;
            maxref = std::max(maxref, std::abs(X[k]));
            maxerr = std::max(maxerr, std::abs(cd(rp[L][j].x, rp[L][j].y) - X[k]));
        }
    bad = 0;
    for (int k = 0; k < kFftN; ++k) bad += seen[k] != 1;
    std::printf("fft3 relerr %.3e coverage_bad %d\n", maxerr / maxref, bad); // "fft3" = FFT-1 by a pair
    for (int L = 0; L < kPairLanes; ++L)
        for (int j = 0; j < kPairPts; ++j) rp[L][j] = cmul(rp[L][j], t[fft1p_out_index(L, j)]);
    for (int L = 0; L < kPairLanes; ++L) fft2p_passA(L, rp[L].data(), twAp.data());
    for (int h = 0; h < 2; ++h) {
        for (int L = 0; L < kPairLanes; ++L) fft2p_storeA(L, rp[L].data(), lds.data(), h);
        for (int L = 64 * h; L < 64 * h + 64; ++L) fft2p_loadB(L, bp[L].data(), lds.data());
    }
    for (int L = 0; L < kPairLanes; ++L) fft2p_passB(L, bp[L].data(), twB.data());
    for (int h = 0; h < 2; ++h) {
        for (int L = 0; L < kPairLanes; ++L) fft2p_storeB(L, bp[L].data(), lds.data(), h);
        for (int L = 64 * h; L < 64 * h + 64; ++L) fft2p_loadC(L, cp[L].data(), lds.data());
    }
    for (int L = 0; L < kPairLanes; ++L) fft2p_passC(cp[L].data());
    maxref = maxerr = 0;
    std::fill(seen.begin(), seen.end(), 0);
    int differ = 0;
    for (int L = 0; L < kPairLanes; ++L)
        for (int j = 0; j < kPairPts; ++j) {
            const int k = fft2p_out_index(L, j);
            seen[k]++;
            maxref = std::max(maxref, std::abs(C[k]));
            maxerr = std::max(maxerr, std::abs(cd(cp[L][j].x, cp[L][j].y) - C[k]));
            differ += !(cp[L][j].x == one_wave_C[k].x && cp[L][j].y == one_wave_C[k].y);
        }
    bad = 0;
    for (int k = 0; k < kFftN; ++k) bad += seen[k] != 1;
    std::printf("fft4 relerr %.3e coverage_bad %d\n", maxerr / maxref, bad); // "fft4" = FFT-2 by a pair
    std::printf("pair_vs_wave_differing_values %d\n", differ);
    return 0;
}
