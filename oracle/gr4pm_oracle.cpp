/*
 * gr4pm_oracle.cpp -- CPU oracle (test infrastructure only; see gr4pm_oracle.h).
 *
 * Build: g++ -O3 -march=x86-64-v3 -ffp-contract=off -std=c++17 -shared -fPIC (oracle/Makefile; v3 = AVX2, so the
 * .so built in the build container also runs on the GPU box's host CPU).
 * -ffp-contract=off matters: the reference is built for baseline x86-64 (no FMA), so every
 * product and sum below rounds separately, in the order the reference source evaluates them.
 *
 * Citations are relative to /root/reference/blocks/include/gnuradio-4.0/packet-modem/.
 */
#include "gr4pm_oracle.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr float kPiF = 3.14159265358979323846f;

using c64 = std::complex<float>;

/* std::complex<float> operator* without the C99 Annex G inf/nan recovery branch:
 * (a+ib)(c+id) = (ac - bd) + i(ad + bc), four products and two sums, each rounded. */
inline c64 cmul(c64 x, c64 y)
{
    const float a = x.real(), b = x.imag(), c = y.real(), d = y.imag();
    return { a * c - b * d, a * d + b * c };
}
inline c64 fmulc(float t, c64 z) { return { t * z.real(), t * z.imag() }; }

/* gr::HistoryBuffer semantics the blocks rely on: power-of-two capacity, zero-initialised
 * storage, push_back() makes the new item index 0, operator[] is unchecked. */
template <typename T>
struct History {
    std::vector<T> buf;
    size_t mask = 0, head = 0, count = 0;
    explicit History(size_t capacity = 1) { reset(capacity); }
    void reset(size_t capacity)
    {
        size_t cap = 1;
        while (cap < capacity) cap <<= 1;
        buf.assign(cap, T{});
        mask = cap - 1;
        head = 0;
        count = 0;
    }
    size_t capacity() const { return mask + 1; }
    size_t size() const { return count; }
    void push_back(const T& v)
    {
        head = (head + mask) & mask; /* head - 1 mod cap */
        buf[head] = v;
        if (count < capacity()) ++count;
    }
    T& operator[](size_t i) { return buf[(head + i) & mask]; }
    const T& operator[](size_t i) const { return buf[(head + i) & mask]; }
};

/* ------------------------------------------------------------------------------------
 * FFT: FFTW3f contract restated (forward sign e^{-j2pi nk/N}, un-normalised), as a
 * Stockham autosort radix-4 (+ one radix-2 step when log2 N is odd).  Twiddles are
 * computed in double and rounded once.
 * ---------------------------------------------------------------------------------- */
struct FftPlan {
    size_t n = 0;
    std::vector<c64> w; /* w[k] = exp(-j 2 pi k / n), k < n */
    std::vector<c64> a, b;
    explicit FftPlan(size_t n_) : n(n_), w(n_), a(n_), b(n_)
    {
        for (size_t k = 0; k < n; ++k) {
            const double ph = -2.0 * kPi * static_cast<double>(k) / static_cast<double>(n);
            w[k] = { static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)) };
        }
    }
    /* in and out may alias neither a nor b; out receives the natural-order spectrum */
    void forward(const c64* in, c64* out)
    {
        c64* x = a.data();
        c64* y = b.data();
        std::memcpy(x, in, n * sizeof(c64));
        size_t len = n, s = 1;
        while (len >= 4) {
            const size_t m = len / 4;
            const size_t tw_step = n / len;
            for (size_t p = 0; p < m; ++p) {
                const c64 w1 = w[p * tw_step];
                const c64 w2 = w[2 * p * tw_step];
                const c64 w3 = w[3 * p * tw_step];
                const c64* x0 = x + s * p;
                const c64* x1 = x + s * (p + m);
                const c64* x2 = x + s * (p + 2 * m);
                const c64* x3 = x + s * (p + 3 * m);
                c64* y0 = y + s * (4 * p);
                c64* y1 = y0 + s;
                c64* y2 = y1 + s;
                c64* y3 = y2 + s;
                for (size_t q = 0; q < s; ++q) {
                    const c64 A = x0[q], B = x1[q], C = x2[q], D = x3[q];
                    const c64 apc = A + C, amc = A - C, bpd = B + D, bmd = B - D;
                    /* -j * bmd */
                    const c64 jbmd = { bmd.imag(), -bmd.real() };
                    y0[q] = apc + bpd;
                    y1[q] = cmul(w1, amc + jbmd);
                    y2[q] = cmul(w2, apc - bpd);
                    y3[q] = cmul(w3, amc - jbmd);
                }
            }
            std::swap(x, y);
            len = m;
            s *= 4;
        }
        if (len == 2) {
            for (size_t q = 0; q < s; ++q) {
                const c64 A = x[q], B = x[q + s];
                y[q] = A + B;
                y[q + s] = A - B;
            }
            std::swap(x, y);
        }
        std::memcpy(out, x, n * sizeof(c64));
    }
};

} // namespace

extern "C" {

/* ---------------------------------------------------------------- firdes.hpp:29-76 */
size_t orc_rrc_taps(double gain, double sampling_freq, double symbol_rate, double alpha,
                    size_t ntaps, float* out)
{
    ntaps |= 1; /* firdes.hpp:33 */
    const double spb = sampling_freq / symbol_rate;
    std::vector<double> taps(ntaps);
    for (size_t i = 0; i < ntaps; ++i) {
        const double xindx =
            static_cast<double>(static_cast<long>(i) - static_cast<long>(ntaps) / 2);
        const double x1 = kPi * xindx / spb;
        double x2 = 4.0 * alpha * xindx / spb;
        double x3 = x2 * x2 - 1.0;
        double num, den;
        if (std::abs(x3) >= 0.000001) { /* firdes.hpp:45 */
            if (i != ntaps / 2) {
                num = std::cos((1.0 + alpha) * x1) +
                      std::sin((1.0 - alpha) * x1) / (4.0 * alpha * xindx / spb);
            } else {
                num = std::cos((1.0 + alpha) * x1) + (1.0 - alpha) * kPi / (4.0 * alpha);
            }
            den = x3 * kPi;
        } else {
            if (alpha == 1.0) {
                taps[i] = -1.0; /* firdes.hpp:55-57 */
                continue;
            }
            x3 = (1.0 - alpha) * x1;
            x2 = (1.0 + alpha) * x1;
            num = (std::sin(x2) * (1.0 + alpha) * kPi -
                   std::cos(x3) * ((1.0 - alpha) * kPi * spb) / (4.0 * alpha * xindx) +
                   std::sin(x3) * spb * spb / (4.0 * alpha * xindx * xindx));
            den = -32.0 * kPi * alpha * alpha * xindx / spb;
        }
        taps[i] = 4.0 * alpha * num / den;
    }
    double scale = 0.0;
    for (double t : taps) scale += t; /* std::accumulate, firdes.hpp:68 */
    for (size_t i = 0; i < ntaps; ++i) out[i] = static_cast<float>(taps[i] * gain / scale);
    return ntaps;
}

/* ------------------------------------------- packet_transmitter_rrc_taps.hpp:8-28 */
size_t orc_tx_rrc_taps(size_t sps, float* out)
{
    const size_t n = orc_rrc_taps(1.0, static_cast<double>(sps), 1.0, 0.35, sps * 11U, out);
    float sum_abs_max = 0.0f;
    for (size_t j = 0; j < sps; ++j) {
        float sum_abs = 0.0f;
        for (size_t k = j; k < n; k += sps) sum_abs += std::abs(out[k]);
        sum_abs_max = std::max(sum_abs_max, sum_abs);
    }
    const float scale = 0.9f;
    for (size_t i = 0; i < n; ++i) out[i] *= scale / sum_abs_max;
    return n;
}

void orc_fft(const orc_c64* in, orc_c64* out, size_t n)
{
    FftPlan plan(n);
    plan.forward(reinterpret_cast<const c64*>(in), reinterpret_cast<c64*>(out));
}

/* ------------------------------------------------ syncword_detection.hpp:17-357 */
struct HistoryItem { /* :17-29 */
    c64 sample{};
    float correlation_power = 0.0f;
    float correlation_power_left = 0.0f;
    float correlation_power_right = 0.0f;
    c64 correlation{};
    int freq_bin = 0;
    float fft_noise_power = 0.0f;
    bool detection = false;
};

struct orc_sd {
    size_t fft_size, sps;
    std::vector<float> rrc_taps;
    std::vector<uint8_t> syncword;
    std::vector<c64> constellation;
    int min_freq_bin, max_freq_bin;
    uint64_t time_threshold;
    float power_threshold;
    /* state (:118-128) */
    size_t syncword_samples_size = 0;
    std::vector<std::vector<c64>> syncword_fft_conj;
    float syncword_self_corr = 0.0f;
    float best = 0.0f;
    uint64_t best_idx = 0;
    uint64_t items_consumed = 0;
    size_t history_size = 0;
    History<HistoryItem> history{ 2 };
    FftPlan* fft = nullptr;
    std::vector<c64> samples_fft, prod;
    std::vector<std::vector<c64>> correlation;
};

/* output_tag(), :56-115 */
static orc_tag sd_output_tag(const orc_sd* s, const HistoryItem& item, const HistoryItem& prev,
                             const HistoryItem& next, uint64_t index)
{
    const double bin_spacing = kPi / static_cast<double>(s->syncword_samples_size);
    double syncword_freq = static_cast<double>(item.freq_bin) * bin_spacing;
    float syncword_phase = std::arg(item.correlation);
    float correlation_power;
    if (item.freq_bin > s->min_freq_bin && item.freq_bin < s->max_freq_bin) {
        const double a = static_cast<double>(item.correlation_power_left);
        const double b = static_cast<double>(item.correlation_power);
        const double c = static_cast<double>(item.correlation_power_right);
        const double quad = std::clamp((c - a) / (2.0 * (2.0 * b - (a + c))), -0.5, 0.5);
        const double delta_freq = quad * bin_spacing;
        syncword_freq += delta_freq;
        syncword_phase -= static_cast<float>(delta_freq * 0.5 *
                                             static_cast<double>(s->syncword_samples_size));
        if (syncword_phase >= kPiF) {
            syncword_phase -= 2.0f * kPiF;
        } else if (syncword_phase < -kPiF) {
            syncword_phase += 2.0f * kPiF;
        }
        correlation_power =
            static_cast<float>(b + (c - a) * (c - a) / (16.0 * (b - 0.5 * (a + c))));
    } else {
        correlation_power = item.correlation_power;
    }
    const float syncword_amplitude =
        std::sqrt(correlation_power) / (static_cast<float>(s->fft_size) * s->syncword_self_corr);
    const float syncword_power = syncword_amplitude * syncword_amplitude * s->syncword_self_corr;
    const float esn0_db =
        10.0f * std::log10((syncword_power * static_cast<float>(s->sps)) /
                           (item.fft_noise_power * static_cast<float>(s->syncword_samples_size)));
    const double a = static_cast<double>(prev.correlation_power);
    const double b = static_cast<double>(item.correlation_power);
    const double c = static_cast<double>(next.correlation_power);
    const float time_est =
        static_cast<float>(std::clamp((c - a) / (2.0 * (2.0 * b - (a + c))), -0.5, 0.5));
    orc_tag t;
    t.index = index;
    t.amplitude = syncword_amplitude;
    t.phase = syncword_phase;
    t.freq = syncword_freq;
    t.freq_bin = item.freq_bin;
    t.noise_power = item.fft_noise_power;
    t.esn0_db = esn0_db;
    t.time_est = time_est;
    t.flags = 1;
    t.user = 0;
    return t;
}

orc_sd* orc_sd_create(size_t fft_size, size_t sps, const float* rrc_taps, size_t n_taps,
                      const uint8_t* syncword, size_t n_syncword, const orc_c64* constellation,
                      size_t n_constellation, int min_freq_bin, int max_freq_bin,
                      uint64_t time_threshold, float power_threshold)
{
    /* start(), :143-202 */
    if (min_freq_bin > max_freq_bin) return nullptr; /* :145-147 */
    auto* s = new orc_sd;
    s->fft_size = fft_size;
    s->sps = sps;
    s->rrc_taps.assign(rrc_taps, rrc_taps + n_taps);
    s->syncword.assign(syncword, syncword + n_syncword);
    s->constellation.resize(n_constellation);
    for (size_t i = 0; i < n_constellation; ++i)
        s->constellation[i] = { constellation[i].re, constellation[i].im };
    s->min_freq_bin = min_freq_bin;
    s->max_freq_bin = max_freq_bin;
    s->time_threshold = time_threshold;
    s->power_threshold = power_threshold;
    s->syncword_samples_size = (n_syncword - 1) * sps + n_taps; /* :148-149 */
    if (s->syncword_samples_size > fft_size) { /* :150-152 */
        delete s;
        return nullptr;
    }
    s->fft = new FftPlan(fft_size);
    std::vector<c64> syncword_samples(s->syncword_samples_size);
    for (size_t j = 0; j < n_syncword; ++j) { /* :155-160 */
        for (size_t k = 0; k < n_taps; ++k) {
            syncword_samples[j * sps + k] += fmulc(s->rrc_taps[k], s->constellation[syncword[j]]);
        }
    }
    s->syncword_self_corr = 0.0f; /* :161-164 */
    for (auto x : syncword_samples)
        s->syncword_self_corr += x.real() * x.real() + x.imag() * x.imag();
    for (int freq_bin = min_freq_bin; freq_bin <= max_freq_bin; ++freq_bin) { /* :166-189 */
        double phase = 0.0;
        const double phase_incr =
            static_cast<double>(freq_bin) * kPi / static_cast<double>(s->syncword_samples_size);
        std::vector<c64> shifted = syncword_samples;
        for (auto& x : shifted) {
            x = cmul(x, c64{ static_cast<float>(std::cos(phase)),
                             static_cast<float>(std::sin(phase)) });
            phase += phase_incr;
            if (phase >= kPi) {
                phase -= 2.0 * kPi;
            } else if (phase < kPi) { /* sic: :179, reproduces the reference's wrap quirk */
                phase += 2.0 * kPi;
            }
        }
        shifted.resize(fft_size);
        std::vector<c64> f(fft_size);
        s->fft->forward(shifted.data(), f.data());
        for (auto& z : f) z = std::conj(z);
        s->syncword_fft_conj.push_back(std::move(f));
    }
    s->best = 0.0f;
    s->best_idx = 0;
    s->items_consumed = 0;
    s->history_size = 2 * time_threshold + 1;
    s->history.reset(s->history_size + 1); /* bit_ceil(history_size + 1), :198-199 */
    s->samples_fft.resize(fft_size);
    s->prod.resize(fft_size);
    s->correlation.assign(static_cast<size_t>(max_freq_bin - min_freq_bin + 1),
                          std::vector<c64>(fft_size));
    return s;
}

void orc_sd_destroy(orc_sd* s)
{
    if (!s) return;
    delete s->fft;
    delete s;
}
size_t orc_sd_syncword_samples_size(const orc_sd* s) { return s->syncword_samples_size; }
float orc_sd_self_corr(const orc_sd* s) { return s->syncword_self_corr; }
void orc_sd_template(const orc_sd* s, size_t b, orc_c64* out)
{
    std::memcpy(out, s->syncword_fft_conj[b].data(), s->fft_size * sizeof(c64));
}

int orc_sd_process(orc_sd* s, const orc_c64* in_, size_t n_in, orc_c64* out_, size_t* n_done,
                   orc_tag* tags, size_t tags_cap, size_t* n_tags, float* zpow_dbg,
                   int32_t* bin_dbg)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    *n_done = 0;
    if (n_tags) *n_tags = 0;
    const size_t fft_size = s->fft_size;
    if (n_in < fft_size) return 1; /* :215-227 */
    const size_t num_freq_bins = s->syncword_fft_conj.size();
    const size_t stride = fft_size - s->syncword_samples_size + 1; /* :236 */
    const size_t history_size = s->history_size;
    size_t ntag = 0;
    size_t j;
    for (j = 0; j + fft_size <= n_in; j += stride) { /* :238 */
        s->fft->forward(in + j, s->samples_fft.data());
        for (size_t nfreq = 0; nfreq < num_freq_bins; ++nfreq) { /* :246-252 */
            const c64* tpl = s->syncword_fft_conj[nfreq].data();
            for (size_t k = 0; k < fft_size; ++k) s->prod[k] = cmul(s->samples_fft[k], tpl[k]);
            s->fft->forward(s->prod.data(), s->correlation[nfreq].data());
        }
        float fft_noise_power = 0.0f; /* :257-265 */
        for (size_t k = fft_size / 4; k < 3 * fft_size / 4; ++k) {
            const c64 z = s->samples_fft[k];
            fft_noise_power += z.real() * z.real() + z.imag() * z.imag();
        }
        fft_noise_power /= static_cast<float>(fft_size / 2) * static_cast<float>(fft_size);

        for (size_t k = 0; k < stride; ++k) { /* :267-343 */
            const uint64_t curr_idx = s->items_consumed + j + k;
            if (curr_idx - s->best_idx > s->time_threshold) {
                size_t below_threshold = 0;
                const float thr = s->best / s->power_threshold;
                for (size_t u = 0; u < history_size; ++u) {
                    if (s->history[u].correlation_power < thr) ++below_threshold;
                }
                if (2 * below_threshold >= history_size) {
                    const size_t hist_idx = s->best_idx + history_size - curr_idx;
                    s->history[history_size - 1 - hist_idx].detection = true;
                }
                s->best = 0.0f;
                s->best_idx = curr_idx;
            }
            const size_t z_idx = k == 0 ? 0 : fft_size - k;
            size_t best_freq = 0;
            c64 z{};
            float zpow = -1.0f;
            for (size_t nfreq = 0; nfreq < num_freq_bins; ++nfreq) {
                const c64 zz = s->correlation[nfreq][z_idx];
                const float zzpow = zz.real() * zz.real() + zz.imag() * zz.imag();
                if (zzpow > zpow) {
                    best_freq = nfreq;
                    z = zz;
                    zpow = zzpow;
                }
            }
            if (zpow > s->best) {
                s->best = zpow;
                s->best_idx = curr_idx;
            }
            const HistoryItem& pop = s->history[history_size - 1];
            out[j + k] = pop.sample;
            if (pop.detection) {
                if (tags && ntag < tags_cap) {
                    tags[ntag] = sd_output_tag(s, pop, s->history[history_size],
                                               s->history[history_size - 2], curr_idx);
                }
                ++ntag;
            }
            HistoryItem item;
            item.sample = in[j + k];
            item.correlation_power = zpow;
            if (best_freq > 0) {
                const c64 zl = s->correlation[best_freq - 1][z_idx];
                item.correlation_power_left = zl.real() * zl.real() + zl.imag() * zl.imag();
            }
            if (best_freq < num_freq_bins - 1) {
                const c64 zr = s->correlation[best_freq + 1][z_idx];
                item.correlation_power_right = zr.real() * zr.real() + zr.imag() * zr.imag();
            }
            item.correlation = z;
            item.freq_bin = s->min_freq_bin + static_cast<int>(best_freq);
            item.fft_noise_power = fft_noise_power;
            s->history.push_back(item);
            if (zpow_dbg) zpow_dbg[j + k] = zpow;
            if (bin_dbg) bin_dbg[j + k] = item.freq_bin;
        }
    }
    s->items_consumed += j; /* :349 */
    *n_done = j;
    if (n_tags) *n_tags = ntag;
    return 0;
}

/* --------------------------------------- syncword_detection_filter.hpp:54-210 */
struct orc_sdf {
    size_t sps, syncword_size, header_size, allowed_margin = 16;
    bool in_packet = false;
    size_t position = 0, block_until = 0;
};
orc_sdf* orc_sdf_create(size_t sps, size_t syncword_size, size_t header_size)
{
    auto* f = new orc_sdf;
    f->sps = sps;
    f->syncword_size = syncword_size;
    f->header_size = header_size;
    return f;
}
void orc_sdf_destroy(orc_sdf* f) { delete f; }

int orc_sdf_process(orc_sdf* f, const orc_c64* in, size_t n_in, orc_c64* out, size_t out_cap,
                    int tag_flags, size_t n_headers, const uint64_t* header_packet_length,
                    const uint8_t* header_invalid, size_t n_ignored, size_t* consumed_,
                    size_t* header_consumed_, size_t* ignored_consumed_, int* tag_out_flags)
{
    *consumed_ = 0;
    *header_consumed_ = 0;
    *ignored_consumed_ = 0;
    *tag_out_flags = 0;
    if (tag_flags) { /* :75-105 */
        int out_flags = 0;
        bool new_in_packet = false;
        if (tag_flags & 1) { /* syncword_* keys */
            if (!f->in_packet) {
                new_in_packet = true;
                out_flags |= 1;
            }
        }
        if (tag_flags & 2) out_flags |= 2; /* non-syncword keys always pass */
        if (new_in_packet) {
            f->in_packet = true;
            f->position = 0;
            f->block_until = 0;
        }
        *tag_out_flags = out_flags;
    }
    if (!f->in_packet) { /* :107-130 */
        const size_t n = std::min(n_in, out_cap);
        std::memcpy(out, in, n * sizeof(orc_c64));
        *consumed_ = n;
        return 0;
    }
    size_t header_consumed = 0;
    if (f->block_until == 0 && n_headers > 0) { /* :134-153 */
        header_consumed = 1;
        if (header_invalid[0]) {
            f->block_until = 1;
        } else {
            const uint64_t packet_length = header_packet_length[0];
            if (packet_length == 0) return -1; /* :143-145 throws */
            const size_t payload_symbols = (packet_length + 4) * 4;
            f->block_until = f->sps * (f->header_size + f->syncword_size - f->allowed_margin +
                                       payload_symbols);
        }
    }
    size_t ignored_consumed = 0;
    if (f->block_until == 0 && n_ignored > 0) { /* :157-160 */
        ignored_consumed = 1;
        f->block_until = 1;
    }
    size_t consumed = 0;
    const size_t allowed = f->sps * (f->syncword_size + f->header_size + f->allowed_margin);
    if (f->position < allowed) { /* :166-172 */
        const size_t n = std::min({ n_in, out_cap, allowed - f->position });
        std::memcpy(out, in, n * sizeof(orc_c64));
        f->position += n;
        consumed = n;
    }
    if (f->position >= allowed && f->block_until != 0) { /* :174-185 */
        const size_t n = std::min(n_in, out_cap) - consumed;
        std::memcpy(out + consumed, in + consumed, n * sizeof(orc_c64));
        f->position += n;
        consumed += n;
        if (f->position >= f->block_until) f->in_packet = false;
    }
    *consumed_ = consumed;
    *header_consumed_ = header_consumed;
    *ignored_consumed_ = ignored_consumed;
    return 0;
}

/* ------------------------------------- coarse_frequency_correction.hpp:40-98 */
struct orc_cfc {
    c64 exp{ 1.0f, 0.0f }, exp_incr{ 1.0f, 0.0f };
    unsigned counter = 0;
    size_t delay = 0;
    float next_freq = 0.0f;
    long next_freq_delay = 0;
};
orc_cfc* orc_cfc_create(size_t delay)
{
    auto* c = new orc_cfc;
    c->delay = delay;
    return c;
}
void orc_cfc_destroy(orc_cfc* c) { delete c; }

static void cfc_set_freq(orc_cfc* c, float freq) /* :50-59 */
{
    c->exp = { std::cos(freq * static_cast<float>(c->delay)),
               -std::sin(freq * static_cast<float>(c->delay)) };
    c->exp_incr = { std::cos(freq), -std::sin(freq) };
    c->counter = 0;
}
static void cfc_chunk(orc_cfc* c, const c64* in, size_t n, c64* out, bool has_tag, double freq)
{
    if (has_tag) { /* :76-82 */
        c->next_freq = static_cast<float>(freq);
        c->next_freq_delay = static_cast<long>(c->delay);
    }
    for (size_t j = 0; j < n; ++j) { /* :83-96 */
        if (c->next_freq_delay == 0) cfc_set_freq(c, c->next_freq);
        out[j] = cmul(in[j], c->exp);
        c->exp = cmul(c->exp, c->exp_incr);
        if ((++c->counter % 512) == 0) {
            const float r = std::abs(c->exp);
            c->exp = { c->exp.real() / r, c->exp.imag() / r };
        }
        if (c->next_freq_delay >= 0) --c->next_freq_delay;
    }
}
void orc_cfc_process(orc_cfc* c, const orc_c64* in_, size_t n, orc_c64* out_,
                     const uint64_t* tag_index, const double* tag_freq, size_t n_tags)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    size_t pos = 0, t = 0;
    while (pos < n) {
        while (t < n_tags && tag_index[t] < pos) ++t;
        const bool has_tag = t < n_tags && tag_index[t] == pos;
        size_t end = n;
        const size_t nt = has_tag ? t + 1 : t;
        if (nt < n_tags && tag_index[nt] < end) end = tag_index[nt];
        cfc_chunk(c, in + pos, end - pos, out + pos, has_tag, has_tag ? tag_freq[t] : 0.0);
        if (has_tag) ++t;
        pos = end;
    }
}

/* ------------------------------------------------------------ rotator.hpp:44-65 */
struct orc_rot {
    c64 exp{ 1.0f, 0.0f }, exp_incr{ 1.0f, 0.0f };
    unsigned counter = 0;
};
orc_rot* orc_rot_create(float phase_incr)
{
    auto* r = new orc_rot;
    r->exp_incr = { std::cos(phase_incr), std::sin(phase_incr) }; /* :46 */
    return r;
}
void orc_rot_destroy(orc_rot* r) { delete r; }
void orc_rot_process(orc_rot* r, const orc_c64* in_, size_t n, orc_c64* out_)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    for (size_t j = 0; j < n; ++j) { /* :56-65 */
        out[j] = cmul(in[j], r->exp);
        r->exp = cmul(r->exp, r->exp_incr);
        if ((++r->counter % 512) == 0) {
            const float a = std::abs(r->exp);
            r->exp = { r->exp.real() / a, r->exp.imag() / a };
        }
    }
}

/* ------------------------------------------------------- costas_loop.hpp:52-148 */
struct orc_costas {
    float phase = 0.0f, freq = 0.0f, k1 = 0.0f, k2 = 0.0f;
    int constellation = 1;
    double loop_bandwidth = 0.01;
};
/* settingsChanged(), :52-88 */
static void costas_settings_changed(orc_costas* c)
{
    double discriminant_gain = 1.0; /* :62-65 */
    if (c->constellation == 2) discriminant_gain = 1.41421356237309504880;
    const double bw = c->loop_bandwidth;
    const double bw2 = bw * bw, bw3 = bw2 * bw, bw4 = bw2 * bw2;
    const double s = std::cbrt(36.0 * bw2 +
                               std::sqrt(3.0) * std::sqrt(432.0 * bw4 + 848.0 * bw3 +
                                                          624.0 * bw2 + 204.0 * bw + 25.0) +
                               36.0 * bw + 9.0); /* :71-76 */
    const double z = -(-12.0 * bw - 6.0) / (3.0 * std::cbrt(6.0) * (2.0 * bw + 1.0) * s) +
                     (std::cbrt(2.0) * s) / (std::cbrt(9.0) * (2.0 * bw + 1.0)) - 1.0;
    const double k1 = 1.0 - z * z;
    const double k2 = (1.0 - z) * (1.0 - z);
    c->k1 = static_cast<float>(k1 / discriminant_gain);
    c->k2 = static_cast<float>(k2 / discriminant_gain);
}
orc_costas* orc_costas_create(double loop_bandwidth, int constellation)
{
    auto* c = new orc_costas;
    c->constellation = constellation;
    c->loop_bandwidth = loop_bandwidth;
    costas_settings_changed(c);
    return c;
}
void orc_costas_destroy(orc_costas* c) { delete c; }
/* the host libm's sinf / cosf, what std::cos(float) / std::sin(float) of costas_loop.hpp:113-115 call */
void orc_sincosf(const float* x, size_t n, float* s, float* c)
{
    for (size_t i = 0; i < n; ++i) {
        s[i] = sinf(x[i]);
        c[i] = cosf(x[i]);
    }
}
void orc_costas_coeffs(const orc_costas* c, float* k1, float* k2)
{
    *k1 = c->k1;
    *k2 = c->k2;
}
static inline void costas_item(orc_costas* c, const c64* in, c64* out, size_t j);
void orc_costas_process(orc_costas* c, const orc_c64* in_, size_t n, orc_c64* out_,
                        const uint64_t* tag_index, const float* tag_phase, size_t n_tags)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    size_t t = 0;
    for (size_t j = 0; j < n; ++j) {
        while (t < n_tags && tag_index[t] < j) ++t;
        if (t < n_tags && tag_index[t] == j) { /* :101-106 set_phase at chunk head */
            c->phase = tag_phase[t];
            c->freq = 0.0f;
            ++t;
        }
        costas_item(c, in, out, j);
    }
}
void orc_costas_process_packets(orc_costas* c, const orc_c64* in_, size_t n, orc_c64* out_,
                                const orc_ptag* tags, size_t n_tags)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    size_t t = 0;
    for (size_t j = 0; j < n; ++j) {
        while (t < n_tags && tags[t].index < j) ++t;
        for (; t < n_tags && tags[t].index == j; ++t) {
            /* keys that name settings are applied before the chunk is processed, then
             * settingsChanged() runs (:52-88) */
            bool changed = false;
            if (tags[t].constellation >= 0) {
                c->constellation = tags[t].constellation;
                changed = true;
            }
            if (tags[t].loop_bandwidth >= 0.0) {
                c->loop_bandwidth = tags[t].loop_bandwidth;
                changed = true;
            }
            if (changed) costas_settings_changed(c);
            if (tags[t].kind == 1 && (tags[t].syncword.flags & 1)) { /* :101-106 */
                c->phase = tags[t].syncword.phase;
                c->freq = 0.0f;
            }
        }
        costas_item(c, in, out, j);
    }
}
static inline void costas_item(orc_costas* c, const c64* in, c64* out, size_t j)
{
    {
        const c64 lo = { std::cos(c->phase), -std::sin(c->phase) }; /* :114-115 */
        const c64 z = cmul(in[j], lo);
        out[j] = z;
        float error = 0.0f;
        switch (c->constellation) {
        case 0: error = z.imag(); break;
        case 1: error = z.real() * z.imag(); break;
        default:
            error = (z.real() > 0 ? z.imag() : -z.imag()) + (z.imag() > 0 ? -z.real() : z.real());
            break;
        }
        c->freq += c->k2 * error; /* :139-145 */
        c->phase += c->k1 * error + c->freq;
        if (c->phase >= kPiF) {
            c->phase -= 2.0f * kPiF;
        } else if (c->phase < -kPiF) {
            c->phase += 2.0f * kPiF;
        }
    }
}

/* -------------------------------------------------- syncword_wipeoff.hpp:38-90 */
struct orc_wipe {
    std::vector<float> syncword;
    bool in_syncword = false;
    size_t position = 0;
};
orc_wipe* orc_wipe_create(const float* syncword, size_t n)
{
    auto* w = new orc_wipe;
    w->syncword.assign(syncword, syncword + n);
    return w;
}
void orc_wipe_destroy(orc_wipe* w) { delete w; }
static void wipe_chunk(orc_wipe* w, const c64* in, size_t n, c64* out, bool has_tag)
{
    if (!w->in_syncword && has_tag) { /* :53-62 */
        w->in_syncword = true;
        w->position = 0;
    }
    size_t i = 0;
    if (w->in_syncword) { /* :66-75 */
        const size_t m = std::min(n, w->syncword.size() - w->position);
        for (; i < m; ++i) out[i] = fmulc(w->syncword[w->position++], in[i]);
        if (w->position == w->syncword.size()) w->in_syncword = false;
    }
    if (!w->in_syncword) { /* :77-82 */
        for (; i < n; ++i) out[i] = in[i];
    }
}
void orc_wipe_process(orc_wipe* w, const orc_c64* in_, size_t n, orc_c64* out_,
                      const uint64_t* tag_index, size_t n_tags)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    size_t pos = 0, t = 0;
    while (pos < n) {
        while (t < n_tags && tag_index[t] < pos) ++t;
        const bool has_tag = t < n_tags && tag_index[t] == pos;
        size_t end = n;
        const size_t nt = has_tag ? t + 1 : t;
        if (nt < n_tags && tag_index[nt] < end) end = tag_index[nt];
        wipe_chunk(w, in + pos, end - pos, out + pos, has_tag);
        if (has_tag) ++t;
        pos = end;
    }
}

/* ---------------------------------------- interpolating_fir_filter.hpp:42-102 */
struct orc_ifir {
    size_t interpolation;
    std::vector<std::vector<float>> arms;
    History<c64> hist_c;
    History<float> hist_f;
};
orc_ifir* orc_ifir_create(size_t interpolation, const float* taps, size_t n_taps)
{
    if (interpolation == 0) return nullptr; /* :45-47 */
    auto* f = new orc_ifir;
    f->interpolation = interpolation;
    f->arms.resize(interpolation);
    for (size_t j = 0; j < interpolation; ++j)
        for (size_t k = j; k < n_taps; k += interpolation) f->arms[j].push_back(taps[k]);
    const size_t cap = (n_taps + interpolation - 1) / interpolation; /* :63-64 */
    f->hist_c.reset(cap);
    f->hist_f.reset(cap);
    return f;
}
void orc_ifir_destroy(orc_ifir* f) { delete f; }
void orc_ifir_process_c64(orc_ifir* f, const orc_c64* in_, size_t n, orc_c64* out_)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    for (size_t i = 0; i < n; ++i) { /* :93-99 */
        f->hist_c.push_back(in[i]);
        for (const auto& arm : f->arms) {
            c64 acc{ 0.0f, 0.0f };
            for (size_t m = 0; m < arm.size(); ++m) acc = acc + fmulc(arm[m], f->hist_c[m]);
            *out++ = acc;
        }
    }
}
void orc_ifir_process_f32(orc_ifir* f, const float* in, size_t n, float* out)
{
    for (size_t i = 0; i < n; ++i) {
        f->hist_f.push_back(in[i]);
        for (const auto& arm : f->arms) {
            float acc = 0.0f;
            for (size_t m = 0; m < arm.size(); ++m) acc = acc + arm[m] * f->hist_f[m];
            *out++ = acc;
        }
    }
}
void orc_ifir_int(size_t interpolation, const int* taps, size_t n_taps, const int* in, size_t n,
                  int* out)
{
    std::vector<std::vector<int>> arms(interpolation);
    for (size_t j = 0; j < interpolation; ++j)
        for (size_t k = j; k < n_taps; k += interpolation) arms[j].push_back(taps[k]);
    History<int> hist((n_taps + interpolation - 1) / interpolation);
    for (size_t i = 0; i < n; ++i) {
        hist.push_back(in[i]);
        for (const auto& arm : arms) {
            int acc = 0;
            for (size_t m = 0; m < arm.size(); ++m) acc += arm[m] * hist[m];
            *out++ = acc;
        }
    }
}

/* ------------------------------------------------------ symbol_filter.hpp:64-252 */
struct SymfTag {
    long index;
    orc_tag tag;
};
struct orc_symf {
    size_t sps, num_arms, delay;
    std::vector<std::vector<float>> taps;
    History<c64> hist_c;
    History<float> hist_f;
    size_t clock_phase = 0, reset_clock_phase = 0, pfb_arm = 0;
    std::vector<SymfTag> tags;
    float scale = 1.0f;
};
orc_symf* orc_symf_create(size_t sps, const float* taps, size_t n_taps, size_t num_arms,
                          size_t delay)
{
    if (sps == 0 || num_arms == 0) return nullptr; /* :67-73 */
    auto* f = new orc_symf;
    f->sps = sps;
    f->num_arms = num_arms;
    f->delay = delay;
    f->taps.resize(num_arms);
    for (size_t j = 0; j < num_arms; ++j)
        for (size_t k = j; k < n_taps; k += num_arms) f->taps[j].push_back(taps[k]);
    f->hist_c.reset(f->taps[0].size());
    f->hist_f.reset(f->taps[0].size());
    f->reset_clock_phase = (sps - (delay % sps)) % sps; /* :106-107 */
    f->clock_phase = 0;                                  /* start(), :110 */
    return f;
}
void orc_symf_destroy(orc_symf* f) { delete f; }

extern "C++" {
template <typename T>
static inline T symf_filter(const orc_symf* f, const History<T>& h)
{
    const auto& arm = f->taps[f->pfb_arm];
    T acc{};
    if constexpr (std::is_same_v<T, c64>) {
        for (size_t m = 0; m < arm.size(); ++m) acc = acc + fmulc(arm[m], h[m]);
        return fmulc(f->scale, acc);
    } else {
        for (size_t m = 0; m < arm.size(); ++m) acc = acc + arm[m] * h[m];
        return f->scale * acc;
    }
}

/* one processBulk call (:112-252): chunk [in, in+n) with an optional tag at in[0] */
template <typename T>
static void symf_chunk(orc_symf* f, History<T>& hist, const T* in, size_t n, T* out,
                       size_t out_cap, const orc_tag* head_tag, size_t out_base,
                       orc_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* consumed,
                       size_t* produced)
{
    size_t ii = 0, oi = 0;
    auto publish = [&](const orc_tag& t, size_t out_index) {
        if (tags_out && *n_tags_out < tags_cap) {
            tags_out[*n_tags_out] = t;
            tags_out[*n_tags_out].index = out_base + out_index;
        }
        ++*n_tags_out;
    };
    if (head_tag) {
        orc_tag tag = *head_tag;
        long tag_index_adjust = 0;
        if (tag.flags & 1) { /* contains syncword_amplitude, :130 */
            size_t new_clock_phase = f->reset_clock_phase;
            f->scale = 1.0f / tag.amplitude;
            float time_est = tag.time_est;
            if (time_est < 0.0f) { /* :148-156 */
                new_clock_phase = (new_clock_phase + 1) % f->sps;
                time_est += 1.0f;
                tag.phase = static_cast<float>(static_cast<double>(tag.phase) - tag.freq);
            }
            if (f->clock_phase == 0 && new_clock_phase == 1) { /* :160-189 */
                hist.push_back(in[ii++]);
                out[oi] = symf_filter(f, hist);
                while (!f->tags.empty() && f->tags[0].index < static_cast<long>(f->sps / 2)) {
                    publish(f->tags[0].tag, oi);
                    f->tags.erase(f->tags.begin());
                }
                ++oi;
                ++new_clock_phase;
                for (auto& t : f->tags) --t.index;
                tag_index_adjust = -1;
            } else if (f->clock_phase == 1 && new_clock_phase == 0) { /* :192-195 */
                hist.push_back(in[ii++]);
                ++new_clock_phase;
            }
            f->clock_phase = new_clock_phase;
            const float arm = std::round(static_cast<float>(f->num_arms) * time_est);
            f->pfb_arm = std::clamp(static_cast<size_t>(arm), size_t{ 0 }, f->num_arms - 1);
        }
        f->tags.push_back({ static_cast<long>(f->delay) + tag_index_adjust, tag });
    }
    while (oi < out_cap && ii < n) { /* :208-238 */
        hist.push_back(in[ii++]);
        if (f->clock_phase == 0) {
            out[oi] = symf_filter(f, hist);
            while (!f->tags.empty() && f->tags[0].index < static_cast<long>(f->sps / 2)) {
                publish(f->tags[0].tag, oi);
                f->tags.erase(f->tags.begin());
            }
            ++oi;
        }
        ++f->clock_phase;
        if (f->clock_phase >= f->sps) f->clock_phase = 0;
        for (auto& t : f->tags) --t.index;
    }
    *consumed = ii;
    *produced = oi;
}

template <typename T>
static size_t symf_stream(orc_symf* f, History<T>& hist, const T* in, size_t n, T* out,
                          size_t out_cap, const orc_tag* tags_in, size_t n_tags_in,
                          orc_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* consumed_)
{
    size_t pos = 0, t = 0, opos = 0, ntag = 0;
    while (pos < n && opos < out_cap) {
        const bool has_tag = t < n_tags_in && tags_in[t].index == pos;
        size_t end = n;
        const size_t nt = has_tag ? t + 1 : t;
        if (nt < n_tags_in && tags_in[nt].index < end) end = tags_in[nt].index;
        size_t consumed = 0, produced = 0;
        symf_chunk(f, hist, in + pos, end - pos, out + opos, out_cap - opos,
                   has_tag ? &tags_in[t] : nullptr, opos, tags_out, tags_cap, &ntag, &consumed,
                   &produced);
        if (has_tag) ++t;
        pos += consumed;
        opos += produced;
        if (consumed < end - (pos - consumed)) break; /* output full */
    }
    if (n_tags_out) *n_tags_out = ntag;
    if (consumed_) *consumed_ = pos;
    return opos;
}

} /* extern "C++" */

size_t orc_symf_process_c64(orc_symf* f, const orc_c64* in, size_t n, orc_c64* out,
                            size_t out_cap, const orc_tag* tags_in, size_t n_tags_in,
                            orc_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                            size_t* consumed)
{
    return symf_stream<c64>(f, f->hist_c, reinterpret_cast<const c64*>(in), n,
                            reinterpret_cast<c64*>(out), out_cap, tags_in, n_tags_in, tags_out,
                            tags_cap, n_tags_out, consumed);
}
size_t orc_symf_process_f32(orc_symf* f, const float* in, size_t n, float* out, size_t out_cap,
                            size_t* consumed)
{
    return symf_stream<float>(f, f->hist_f, in, n, out, out_cap, nullptr, 0, nullptr, 0, nullptr,
                              consumed);
}

/* -------------------------------------------------- pfb_arb_resampler.hpp:67-182 */
struct orc_arb {
    size_t filter_size, arm_size, decim_rate, last_filter;
    bool rate_is_double;
    double filt_rate_d = 0, phase_acc_d = 0;
    float filt_rate_f = 0, phase_acc_f = 0;
    std::vector<std::vector<float>> taps, diff_taps;
    History<c64> hist;
};
orc_arb* orc_arb_create(double rate, int rate_is_double, const float* taps, size_t n_taps,
                        size_t filter_size)
{
    if (filter_size == 0) return nullptr; /* :70-72 */
    auto* r = new orc_arb;
    r->filter_size = filter_size;
    r->rate_is_double = rate_is_double != 0;
    r->arm_size = (n_taps + filter_size - 1) / filter_size; /* :74 */
    r->taps.resize(filter_size);
    r->diff_taps.resize(filter_size);
    for (size_t j = 0; j < filter_size; ++j) {
        for (size_t k = j; k < n_taps; k += filter_size) r->taps[j].push_back(taps[k]);
        r->taps[j].resize(r->arm_size, 0.0f);
        for (size_t k = j; k < n_taps - 1; k += filter_size) /* :96-98 */
            r->diff_taps[j].push_back(taps[k + 1] - taps[k]);
        r->diff_taps[j].resize(r->arm_size, 0.0f);
    }
    r->hist.reset(r->arm_size);
    if (r->rate_is_double) { /* :115-119 */
        const double float_rate = static_cast<double>(filter_size) / rate;
        r->decim_rate = static_cast<size_t>(std::floor(float_rate));
        r->filt_rate_d = float_rate - static_cast<double>(r->decim_rate);
    } else {
        const float float_rate = static_cast<float>(filter_size) / static_cast<float>(rate);
        r->decim_rate = static_cast<size_t>(std::floor(float_rate));
        r->filt_rate_f = float_rate - static_cast<float>(r->decim_rate);
    }
    r->last_filter = (n_taps / 2) % filter_size;
    return r;
}
void orc_arb_destroy(orc_arb* r) { delete r; }
size_t orc_arb_process(orc_arb* r, const orc_c64* in_, size_t n, orc_c64* out_, size_t out_cap,
                       size_t* consumed)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    size_t ii = 0, oi = 0;
    while (ii < n && oi < out_cap) { /* :134-167 */
        while (r->last_filter >= r->filter_size && ii < n) {
            r->hist.push_back(in[ii++]);
            r->last_filter -= r->filter_size;
        }
        if (r->last_filter >= r->filter_size) break;
        const auto& arm = r->taps[r->last_filter];
        const auto& darm = r->diff_taps[r->last_filter];
        c64 filt{ 0.0f, 0.0f }, diff{ 0.0f, 0.0f };
        for (size_t m = 0; m < r->arm_size; ++m) filt = filt + fmulc(arm[m], r->hist[m]);
        for (size_t m = 0; m < r->arm_size; ++m) diff = diff + fmulc(darm[m], r->hist[m]);
        const float pa =
            r->rate_is_double ? static_cast<float>(r->phase_acc_d) : r->phase_acc_f;
        out[oi++] = filt + fmulc(pa, diff);
        r->last_filter += r->decim_rate;
        if (r->rate_is_double) {
            r->phase_acc_d += r->filt_rate_d;
            if (r->phase_acc_d > 1.0) {
                r->phase_acc_d -= 1.0;
                ++r->last_filter;
            }
        } else {
            r->phase_acc_f += r->filt_rate_f;
            if (r->phase_acc_f > 1.0f) {
                r->phase_acc_f -= 1.0f;
                ++r->last_filter;
            }
        }
    }
    if (consumed) *consumed = ii;
    return oi;
}


/* --------------------------------------- payload_metadata_insert.hpp:37-307 */
struct orc_pmi {
    size_t syncword_size, header_size;
    double syncword_bw, header_bw, payload_bw;
    bool in_packet = false;     /* :37 */
    uint64_t position = 0;      /* :38 */
    size_t payload_symbols = 0; /* :39 */
    uint64_t num_packet = 0;    /* :40 */
};
orc_pmi* orc_pmi_create(size_t syncword_size, size_t header_size, double syncword_bw,
                        double header_bw, double payload_bw)
{
    auto* p = new orc_pmi;
    p->syncword_size = syncword_size;
    p->header_size = header_size;
    p->syncword_bw = syncword_bw;
    p->header_bw = header_bw;
    p->payload_bw = payload_bw;
    return p;
}
void orc_pmi_destroy(orc_pmi* p) { delete p; }

namespace {
struct PmiIo {
    const uint64_t* header_packet_length;
    const uint8_t* header_invalid;
    size_t n_headers, header_pos = 0;
    orc_ptag* tags_out;
    size_t tags_cap, n_tags = 0;
    size_t ignored = 0;
    bool overflow = false;
    void publish(const orc_ptag& t)
    {
        if (tags_out && n_tags < tags_cap) tags_out[n_tags] = t;
        else overflow = true;
        ++n_tags;
    }
};
/* one processBulk() call, :77-307.  tag: the syncword tag at in[0] or nullptr.  out_base:
 * absolute index of out[0].  Returns false when the call made no progress because the
 * header message is missing (the block returns and waits, :243-247). */
bool pmi_chunk(orc_pmi* p, const c64* in, size_t n_in, c64* out, size_t n_out, uint64_t out_base,
               const orc_tag* tag, PmiIo& io, size_t* consumed, size_t* produced)
{
    *consumed = *produced = 0;
    if (tag && (tag->flags & 1)) { /* :96-149 */
        if (!p->in_packet) {
            p->in_packet = true;
            p->position = 0;
            ++p->num_packet;
            orc_ptag t{};
            t.index = out_base;
            t.kind = 1;
            t.constellation = 0; /* PILOT: the syncword modulation has been wiped off */
            t.loop_bandwidth = p->syncword_bw;
            t.syncword = *tag;
            io.publish(t);
        } else {
            ++io.ignored; /* :126-147 (the message itself is only sent when `log` is set) */
        }
    }
    if (!p->in_packet) { /* :150-169: discard all the input */
        *consumed = n_in;
        return true;
    }
    size_t ii = 0, oi = 0;
    bool waiting = false;
    while (oi < n_out && ii < n_in) { /* :174 */
        if (p->position < p->syncword_size) { /* :175-184 */
            const size_t n = std::min({ n_in - ii, n_out - oi,
                                        static_cast<size_t>(p->syncword_size - p->position) });
            std::copy_n(in + ii, n, out + oi);
            ii += n;
            oi += n;
            p->position += n;
        }
        if (p->position == p->syncword_size) { /* :186-194 */
            orc_ptag t{};
            t.index = out_base + oi;
            t.kind = 2;
            t.constellation = 2; /* QPSK */
            t.loop_bandwidth = p->header_bw;
            io.publish(t);
        }
        if (p->syncword_size <= p->position && p->position < p->syncword_size + p->header_size) {
            const size_t n = std::min({ n_in - ii, n_out - oi,
                                        static_cast<size_t>(p->syncword_size + p->header_size -
                                                            p->position) });
            std::copy_n(in + ii, n, out + oi); /* :196-205 */
            ii += n;
            oi += n;
            p->position += n;
        }
        if (p->position == p->syncword_size + p->header_size && oi < n_out && ii < n_in) {
            if (io.header_pos < io.n_headers) { /* :207-242 */
                const size_t hd = io.header_pos;
                if (io.header_invalid[hd]) { /* :212-221 */
                    p->in_packet = false;
                    ii = n_in;
                    ++io.header_pos;
                    break;
                }
                const uint64_t packet_length = io.header_packet_length[hd];
                if (packet_length == 0) return false; /* :224-226 throws */
                constexpr size_t crc_size_bytes = 4;
                p->payload_symbols = (packet_length + crc_size_bytes) * 4;
                orc_ptag t{};
                t.index = out_base + oi;
                t.kind = 3;
                t.constellation = -1;
                t.loop_bandwidth = p->payload_bw;
                t.packet_length = packet_length;
                t.payload_symbols = p->payload_symbols;
                t.payload_bits = p->payload_symbols * 2;
                io.publish(t);
                const size_t n = std::min({ n_in - ii, n_out - oi, p->payload_symbols });
                std::copy_n(in + ii, n, out + oi);
                ii += n;
                oi += n;
                p->position += n;
                ++io.header_pos;
            } else { /* :243-247: wait for the header to be decoded */
                waiting = true;
                break;
            }
        }
        if (p->syncword_size + p->header_size < p->position &&
            p->position < p->syncword_size + p->header_size + p->payload_symbols) { /* :250-261 */
            const size_t n =
                std::min({ n_in - ii, n_out - oi,
                           static_cast<size_t>(p->syncword_size + p->header_size +
                                               p->payload_symbols - p->position) });
            std::copy_n(in + ii, n, out + oi);
            ii += n;
            oi += n;
            p->position += n;
        }
        if (p->position >= p->syncword_size + p->header_size + p->payload_symbols) { /* :263-267 */
            p->in_packet = false;
            ii = n_in;
        }
    }
    *consumed = ii;
    *produced = oi;
    return !(waiting && ii == 0);
}
} // namespace

int orc_pmi_process(orc_pmi* p, const orc_c64* in_, size_t n_in, orc_c64* out_, size_t out_cap,
                    const orc_tag* tags_in, size_t n_tags_in, const uint64_t* header_packet_length,
                    const uint8_t* header_invalid, size_t n_headers, orc_ptag* tags_out,
                    size_t tags_cap, size_t* n_tags_out, size_t* consumed, size_t* produced,
                    size_t* headers_used, size_t* ignored_syncwords)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    PmiIo io{ header_packet_length, header_invalid, n_headers, 0, tags_out, tags_cap };
    size_t pos = 0, opos = 0, t = 0;
    while (pos < n_in) {
        while (t < n_tags_in && tags_in[t].index < pos) ++t;
        const bool has_tag = t < n_tags_in && tags_in[t].index == pos;
        size_t end = n_in;
        const size_t nt = has_tag ? t + 1 : t;
        if (nt < n_tags_in && tags_in[nt].index < end) end = tags_in[nt].index;
        size_t c = 0, q = 0;
        const bool ok = pmi_chunk(p, in + pos, end - pos, out + opos, out_cap - opos, opos,
                                  has_tag ? &tags_in[t] : nullptr, io, &c, &q);
        if (has_tag && (c > 0 || !p->in_packet)) ++t; /* the tag is consumed with its item */
        pos += c;
        opos += q;
        if (!ok || c == 0) break; /* waiting for a header (or output full) */
    }
    *n_tags_out = io.n_tags;
    *consumed = pos;
    *produced = opos;
    *headers_used = io.header_pos;
    *ignored_syncwords = io.ignored;
    return io.overflow ? -1 : 0;
}

/* ------------------------------------------------ syncword_remove.hpp:25-105 */
struct orc_sr {
    size_t syncword_size;
    bool in_syncword = false; /* :25 */
    size_t position = 0;      /* :26 */
};
orc_sr* orc_sr_create(size_t syncword_size)
{
    auto* r = new orc_sr;
    r->syncword_size = syncword_size;
    return r;
}
void orc_sr_destroy(orc_sr* r) { delete r; }
extern "C++" {
namespace {
/* one processBulk() call, :39-105; tag_kind: 0 none, 1 syncword_amplitude tag, else other.
 * Returns the number of items copied to out; *pass_tag = the tag goes out at out[0]. */
template <typename T>
size_t sr_chunk(orc_sr* r, const T* in, size_t n, T* out, int tag_kind, bool* pass_tag)
{
    *pass_tag = false;
    if (!r->in_syncword && tag_kind != 0) { /* :51-64 */
        if (tag_kind == 1) {
            r->in_syncword = true;
            r->position = 0;
        } else {
            *pass_tag = true;
        }
    }
    size_t ii = 0;
    if (r->in_syncword) { /* :67-74 */
        const size_t m = std::min(n, r->syncword_size - r->position);
        ii += m;
        r->position += m;
        if (r->position >= r->syncword_size) r->in_syncword = false;
    }
    size_t produced = 0;
    if (!r->in_syncword) { /* :76-81 */
        produced = n - ii;
        std::copy_n(in + ii, produced, out);
    }
    return produced;
}
} // namespace
} // extern "C++"
size_t orc_sr_process(orc_sr* r, const orc_c64* in_, size_t n, orc_c64* out_, const orc_ptag* tags_in,
                      size_t n_tags_in, orc_ptag* tags_out, size_t tags_cap, size_t* n_tags_out)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    c64* out = reinterpret_cast<c64*>(out_);
    size_t pos = 0, opos = 0, t = 0, n_out_tags = 0;
    while (pos < n) {
        while (t < n_tags_in && tags_in[t].index < pos) ++t;
        /* tags on the same item are merged by the runtime into one map: a syncword_amplitude key
         * anywhere in it makes the chunk a syncword chunk */
        size_t t1 = t;
        int kind = 0;
        while (t1 < n_tags_in && tags_in[t1].index == pos) {
            if (tags_in[t1].kind == 1) kind = 1;
            else if (kind == 0) kind = 2;
            ++t1;
        }
        size_t end = n;
        if (t1 < n_tags_in && tags_in[t1].index < end) end = tags_in[t1].index;
        bool pass = false;
        const size_t q = sr_chunk(r, in + pos, end - pos, out + opos, kind, &pass);
        if (pass) {
            for (size_t u = t; u < t1; ++u) {
                if (tags_out && n_out_tags < tags_cap) {
                    tags_out[n_out_tags] = tags_in[u];
                    tags_out[n_out_tags].index = opos;
                }
                ++n_out_tags;
            }
        }
        t = t1;
        pos = end;
        opos += q;
    }
    if (n_tags_out) *n_tags_out = n_out_tags;
    return opos;
}
size_t orc_sr_process_int(orc_sr* r, const int* in, size_t n, int* out, const uint64_t* tag_index,
                          size_t n_tags)
{
    size_t pos = 0, opos = 0, t = 0;
    while (pos < n) {
        while (t < n_tags && tag_index[t] < pos) ++t;
        const bool has_tag = t < n_tags && tag_index[t] == pos;
        size_t end = n;
        const size_t nt = has_tag ? t + 1 : t;
        if (nt < n_tags && tag_index[nt] < end) end = tag_index[nt];
        bool pass = false;
        opos += sr_chunk(r, in + pos, end - pos, out + opos, has_tag ? 1 : 0, &pass);
        if (has_tag) ++t;
        pos = end;
    }
    return opos;
}

/* ------------------------------------- constellation_llr_decoder.hpp:33-134 */
struct orc_llr {
    float noise_sigma, scale;
    int constellation;
};
orc_llr* orc_llr_create(float noise_sigma, int constellation)
{
    auto* d = new orc_llr;
    d->noise_sigma = noise_sigma;
    d->constellation = constellation;
    d->scale = 2.0f / (noise_sigma * noise_sigma); /* :77 */
    return d;
}
void orc_llr_destroy(orc_llr* d) { delete d; }
size_t orc_llr_process(orc_llr* d, const orc_c64* in_, size_t n, float* out, const orc_ptag* tags_in,
                       size_t n_tags_in, orc_ptag* tags_out, size_t tags_cap, size_t* n_tags_out)
{
    const c64* in = reinterpret_cast<const c64*>(in_);
    size_t t = 0, oi = 0, n_out_tags = 0;
    if (d->constellation != 1 && d->constellation != 2) return static_cast<size_t>(-1); /* :72-74 */
    for (size_t j = 0; j < n; ++j) {
        for (; t < n_tags_in && tags_in[t].index <= j; ++t) {
            if (tags_in[t].index < j) continue;
            if (tags_in[t].constellation >= 0) { /* "constellation" names a setting, :22-23 */
                if (tags_in[t].constellation != 1 && tags_in[t].constellation != 2)
                    return static_cast<size_t>(-1);
                d->constellation = tags_in[t].constellation;
            }
            if (tags_out && n_out_tags < tags_cap) { /* :93-99: out.publishTag(tag.map, 0) */
                tags_out[n_out_tags] = tags_in[t];
                tags_out[n_out_tags].index = oi;
            }
            ++n_out_tags;
        }
        if (d->constellation == 1) { /* :106-110 */
            out[oi++] = d->scale * in[j].real();
        } else { /* :111-116 */
            out[oi++] = d->scale * in[j].real();
            out[oi++] = d->scale * in[j].imag();
        }
    }
    if (n_tags_out) *n_tags_out = n_out_tags;
    return oi;
}


/* --------------------------------------------- additive_scrambler.hpp:45-100 */
struct orc_scr {
    uint64_t mask, seed, length, count;
    uint64_t reg, current_count;
};
orc_scr* orc_scr_create(uint64_t mask, uint64_t seed, uint64_t length, uint64_t count)
{
    auto* s = new orc_scr{ mask, seed, length, count, seed, 0 }; /* start(): reset_lfsr(), :68-74 */
    return s;
}
void orc_scr_destroy(orc_scr* s) { delete s; }
static inline uint8_t scr_step(orc_scr* s, bool tag_reset)
{
    if (tag_reset || (s->count != 0 && s->current_count == s->count)) { /* :78-83 */
        s->reg = s->seed;
        s->current_count = 0;
    }
    const uint8_t lfsr_bit = s->reg & 1; /* :84-87 */
    const uint64_t shift_in = static_cast<uint64_t>(__builtin_parityl(s->reg & s->mask));
    s->reg = (shift_in << s->length) | (s->reg >> 1);
    ++s->current_count;
    return lfsr_bit;
}
void orc_scr_process_f32(orc_scr* s, const float* in, size_t n, float* out, const uint64_t* reset_index,
                         size_t n_resets)
{
    size_t t = 0;
    for (size_t j = 0; j < n; ++j) {
        while (t < n_resets && reset_index[t] < j) ++t;
        const bool reset = t < n_resets && reset_index[t] == j;
        out[j] = scr_step(s, reset) ? -in[j] : in[j]; /* :92-93 */
    }
}
void orc_scr_process_u8(orc_scr* s, const uint8_t* in, size_t n, uint8_t* out, const uint64_t* reset_index,
                        size_t n_resets)
{
    size_t t = 0;
    for (size_t j = 0; j < n; ++j) {
        while (t < n_resets && reset_index[t] < j) ++t;
        const bool reset = t < n_resets && reset_index[t] == j;
        out[j] = in[j] ^ scr_step(s, reset); /* :90 */
    }
}

/* ------------------------------------------ header_payload_split.hpp:25-135 */
struct orc_hps {
    size_t header_size;
    bool in_payload = false;   /* :25 */
    uint64_t position = 0;     /* :26 */
    uint64_t payload_items = 0; /* :27 */
};
orc_hps* orc_hps_create(size_t header_size)
{
    auto* h = new orc_hps;
    h->header_size = header_size;
    return h;
}
void orc_hps_destroy(orc_hps* h) { delete h; }
int orc_hps_process(orc_hps* h, const float* in, size_t n, float* header, size_t* n_header, float* payload,
                    size_t* n_payload, const orc_ptag* tags, size_t n_tags, orc_ptag* header_tags,
                    size_t* n_header_tags, orc_ptag* payload_tags, size_t* n_payload_tags, size_t tags_cap)
{
    size_t pos = 0, hp = 0, pp = 0, t = 0, nht = 0, npt = 0;
    while (pos < n) {
        /* one processBulk() call (:51-135): chunk = [pos, next tag) -- and the block handles one
         * of the two outputs per call, so a chunk may take several calls */
        while (t < n_tags && tags[t].index < pos) ++t;
        size_t t1 = t;
        while (t1 < n_tags && tags[t1].index == pos) ++t1;
        const size_t end = t1 < n_tags ? std::min<size_t>(n, tags[t1].index) : n;
        for (size_t u = t; u < t1; ++u) { /* :68-88 */
            if (tags[u].kind == 3) {
                if (h->in_payload || h->position != h->header_size) return -1;
                h->in_payload = true;
                h->position = 0;
                h->payload_items = tags[u].payload_bits;
            }
        }
        for (size_t u = t; u < t1; ++u) {
            orc_ptag o = tags[u];
            if (h->in_payload) {
                o.index = pp;
                if (payload_tags && npt < tags_cap) payload_tags[npt] = o;
                ++npt;
            } else {
                o.index = hp;
                if (header_tags && nht < tags_cap) header_tags[nht] = o;
                ++nht;
            }
        }
        t = t1;
        size_t cur = pos;
        while (cur < end) {
            if (!h->in_payload && h->position == h->header_size) h->position = 0; /* :90-95 */
            if (!h->in_payload) { /* :97-109 */
                const size_t m = std::min<size_t>(end - cur, h->header_size - h->position);
                std::copy_n(in + cur, m, header + hp);
                hp += m;
                cur += m;
                h->position += m;
            } else { /* :110-123 */
                const size_t m = std::min<size_t>(end - cur, h->payload_items - h->position);
                std::copy_n(in + cur, m, payload + pp);
                pp += m;
                cur += m;
                h->position += m;
                if (h->position >= h->payload_items) {
                    h->in_payload = false;
                    h->position = 0;
                }
            }
        }
        pos = end;
    }
    *n_header = hp;
    *n_payload = pp;
    if (n_header_tags) *n_header_tags = nht;
    if (n_payload_tags) *n_payload_tags = npt;
    return 0;
}

/* ------------------------------------------- header_fec_encoder.hpp:60-107 */
void orc_header_fec_encode(const uint32_t* generator, const uint8_t* in, size_t n_codewords, uint8_t* out)
{
    for (size_t c = 0; c < n_codewords; ++c, in += 4, out += 32) {
        std::copy_n(in, 4, out); /* systematic, :80-81 */
        const uint32_t info = (static_cast<uint32_t>(in[0]) << 24) | (static_cast<uint32_t>(in[1]) << 16) |
                              (static_cast<uint32_t>(in[2]) << 8) | static_cast<uint32_t>(in[3]);
        for (int k = 0; k < 12; ++k) { /* :86-96 */
            uint8_t parity_bits = 0;
            for (int l = 0; l < 8; ++l)
                parity_bits = static_cast<uint8_t>(parity_bits << 1) |
                              static_cast<uint8_t>(__builtin_parity(info & generator[8 * k + l]));
            out[4 + k] = parity_bits;
        }
        std::copy_n(out, 16, out + 16); /* repetition, :98-99 */
    }
}

/* -------- LDPC: horizontal layered A-Min*-BP (restated family of ldpc-toolbox "HLAminstar") */
struct orc_ldpc {
    unsigned n = 0, m = 0;
    std::vector<std::vector<unsigned>> rows; /* variable indices of every check */
    std::vector<unsigned> order;             /* the order the layered schedule visits the checks in */
    float corr[64];                          /* ln(1 + e^-x), x = i / 8 */
    float corr_q8[64];                       /* the same in rounded eighths (8-bit message form) */
};
orc_ldpc* orc_ldpc_create(const char* alist)
{
    std::vector<long> v;
    for (const char* p = alist; *p;) {
        if ((*p >= '0' && *p <= '9')) {
            char* e;
            v.push_back(std::strtol(p, &e, 10));
            p = e;
        } else {
            ++p;
        }
    }
    if (v.size() < 4) return nullptr;
    auto* d = new orc_ldpc;
    d->n = static_cast<unsigned>(v[0]);
    d->m = static_cast<unsigned>(v[1]);
    const unsigned max_col = static_cast<unsigned>(v[2]), max_row = static_cast<unsigned>(v[3]);
    size_t i = 4;
    std::vector<unsigned> colw(d->n), roww(d->m);
    for (auto& w : colw) w = static_cast<unsigned>(v[i++]);
    for (auto& w : roww) w = static_cast<unsigned>(v[i++]);
    /* column lists: skipped (either exactly the weight or padded to max_col with zeros) */
    size_t total_col = 0;
    for (auto w : colw) total_col += w;
    const size_t remaining = v.size() - i;
    size_t total_row = 0;
    for (auto w : roww) total_row += w;
    const bool padded = remaining == static_cast<size_t>(d->n) * max_col + static_cast<size_t>(d->m) * max_row;
    i += padded ? static_cast<size_t>(d->n) * max_col : total_col;
    d->rows.resize(d->m);
    for (unsigned c = 0; c < d->m; ++c) {
        const unsigned cnt = padded ? max_row : roww[c];
        for (unsigned e = 0; e < cnt; ++e) {
            const long x = v[i++];
            if (x > 0) d->rows[c].push_back(static_cast<unsigned>(x - 1));
        }
    }
    for (int k = 0; k < 64; ++k) d->corr[k] = static_cast<float>(std::log1p(std::exp(-k / 8.0)));
    for (int k = 0; k < 64; ++k) d->corr_q8[k] = static_cast<float>(std::nearbyint(8.0 * std::log1p(std::exp(-k / 8.0))));
    /* layers: scan the checks not yet placed in index order, take every one that shares no
     * variable with those already taken in this pass (at most 64 edges per pass); the checks
     * are visited pass by pass.  (A parallel decoder can do a whole pass at once.) */
    std::vector<bool> placed(d->m, false);
    while (d->order.size() < d->m) {
        std::vector<bool> used(d->n, false);
        size_t lanes = 0;
        const size_t before = d->order.size();
        for (unsigned c = 0; c < d->m; ++c) {
            if (placed[c]) continue;
            bool clash = lanes + d->rows[c].size() > 64;
            for (unsigned vv : d->rows[c]) clash = clash || used[vv];
            if (clash) continue;
            for (unsigned vv : d->rows[c]) used[vv] = true;
            lanes += d->rows[c].size();
            placed[c] = true;
            d->order.push_back(c);
        }
        if (d->order.size() == before) break;
    }
    return d;
}
void orc_ldpc_destroy(orc_ldpc* d) { delete d; }
/* q8: the 8-bit message form of the product's decoder (include/gr4pm_hip.h, gr4pm_header_fec_decoder_params::arithmetic
 * == 1) -- LLRs are whole eighths carried in floats, channel LLRs and messages saturate at +-127, posteriors at 16 bits.
 * Not a restatement of the ldpc-toolbox crate (absent): the checker of the product's own second form. */
static inline float ldpc_corr(const orc_ldpc* d, float x, bool q8) /* x >= 0 */
{
    if (q8) return x >= 64.0f ? 0.0f : d->corr_q8[static_cast<int>(x)];
    return x >= 8.0f ? 0.0f : d->corr[static_cast<int>(x * 8.0f)];
}
/* |a| [+] |b| in the magnitude domain */
static inline float ldpc_boxplus(const orc_ldpc* d, float a, float b, bool q8)
{
    const float mn = a < b ? a : b;
    const float r = mn + ldpc_corr(d, a + b, q8) - ldpc_corr(d, a < b ? b - a : a - b, q8);
    return r > 0.0f ? r : 0.0f;
}
static inline float ldpc_sat(float x, float limit) { return std::fmin(std::fmax(x, -limit), limit); }
static int ldpc_decode_impl(orc_ldpc* d, const float* llrs, uint8_t* bits_k, unsigned max_iterations, bool q8)
{
    const unsigned k = d->n - d->m;
    std::vector<float> P(llrs, llrs + d->n);
    if (q8)
        for (auto& p : P) p = ldpc_sat(std::nearbyint(8.0f * p), 127.0f);
    std::vector<std::vector<float>> R(d->m);
    for (unsigned c = 0; c < d->m; ++c) R[c].assign(d->rows[c].size(), 0.0f);
    auto is_codeword = [&]() {
        for (unsigned c = 0; c < d->m; ++c) {
            unsigned parity = 0;
            for (unsigned v : d->rows[c]) parity ^= P[v] < 0.0f ? 1u : 0u;
            if (parity) return false;
        }
        return true;
    };
    int used = -1;
    for (unsigned it = 0; it <= max_iterations; ++it) {
        if (is_codeword()) {
            used = static_cast<int>(it);
            break;
        }
        if (it == max_iterations) break;
        for (unsigned c : d->order) { /* check by check, layer by layer */
            const auto& vs = d->rows[c];
            const size_t dc = vs.size();
            float Q[16] = {}, Qwide[16] = {};
            unsigned neg = 0;
            size_t imin = 0;
            for (size_t e = 0; e < dc; ++e) {
                Q[e] = Qwide[e] = P[vs[e]] - R[c][e];
                if (q8) Q[e] = ldpc_sat(Q[e], 127.0f); /* what the check node sees; the posterior keeps its 16 bits */
                if (Q[e] < 0.0f) neg ^= 1u;
                if (std::fabs(Q[e]) < std::fabs(Q[imin])) imin = e;
            }
            float others = -1.0f; /* [+] over all edges but the minimum */
            for (size_t e = 0; e < dc; ++e) {
                if (e == imin) continue;
                const float a = std::fabs(Q[e]);
                others = others < 0.0f ? a : ldpc_boxplus(d, others, a, q8);
            }
            if (others < 0.0f) others = 0.0f; /* degree-1 check */
            const float all = ldpc_boxplus(d, others, std::fabs(Q[imin]), q8);
            for (size_t e = 0; e < dc; ++e) {
                const float mag = e == imin ? others : all;
                const unsigned s = neg ^ (Q[e] < 0.0f ? 1u : 0u);
                const float r = s ? -mag : mag;
                R[c][e] = r;
                P[vs[e]] = q8 ? ldpc_sat(Qwide[e] + r, 32767.0f) : Q[e] + r;
            }
        }
    }
    for (unsigned b = 0; b < k; ++b) bits_k[b] = P[b] < 0.0f ? 1 : 0;
    return used;
}
int orc_ldpc_decode(orc_ldpc* d, const float* llrs, uint8_t* bits_k, unsigned max_iterations)
{
    return ldpc_decode_impl(d, llrs, bits_k, max_iterations, false);
}
int orc_ldpc_decode_q8(orc_ldpc* d, const float* llrs, uint8_t* bits_k, unsigned max_iterations)
{
    return ldpc_decode_impl(d, llrs, bits_k, max_iterations, true);
}
static void header_fec_decode_impl(orc_ldpc* d, const float* llrs, size_t n_codewords, uint8_t* bytes, uint8_t* invalid, bool q8);
void orc_header_fec_decode_q8(orc_ldpc* d, const float* llrs, size_t n_codewords, uint8_t* bytes, uint8_t* invalid)
{
    header_fec_decode_impl(d, llrs, n_codewords, bytes, invalid, true);
}
void orc_header_fec_decode(orc_ldpc* d, const float* llrs, size_t n_codewords, uint8_t* bytes, uint8_t* invalid)
{
    header_fec_decode_impl(d, llrs, n_codewords, bytes, invalid, false);
}
static void header_fec_decode_impl(orc_ldpc* d, const float* llrs, size_t n_codewords, uint8_t* bytes, uint8_t* invalid, bool q8)
{
    for (size_t c = 0; c < n_codewords; ++c, llrs += 256) {
        float acc[128];
        for (int k = 0; k < 128; ++k) acc[k] = llrs[k] + llrs[128 + k]; /* :308-312 */
        uint8_t bits[32];
        const int ret = ldpc_decode_impl(d, acc, bits, 25, q8); /* :315-321 */
        invalid[c] = ret < 0 ? 1 : 0;
        for (int k = 0; k < 4; ++k) { /* :329-335 */
            uint8_t byte = 0;
            for (int b = 0; b < 8; ++b) byte = static_cast<uint8_t>(byte << 1) | bits[8 * k + b];
            bytes[4 * c + k] = byte;
        }
    }
}


/* ----------------------------------------------------------- crc.hpp:31-156 */
namespace {
struct OrcCrc {
    uint64_t table[256];
    unsigned num_bits;
    uint64_t mask, initial_value, final_xor, rem = 0;
    bool input_reflected, result_reflected;
    uint64_t reflect(uint64_t word) const /* :44-53 */
    {
        uint64_t ret = word & 1;
        for (unsigned i = 1; i < num_bits; ++i) {
            word >>= 1;
            ret = (ret << 1) | (word & 1);
        }
        return ret;
    }
    explicit OrcCrc(const orc_crc_params& p)
        : num_bits(p.num_bits), mask(p.num_bits == 64 ? ~uint64_t{ 0 } : ((uint64_t{ 1 } << p.num_bits) - 1)),
          initial_value(p.initial_value & mask), final_xor(p.final_xor & mask),
          input_reflected(p.input_reflected != 0), result_reflected(p.result_reflected != 0)
    {
        uint64_t poly = p.poly;
        table[0] = 0;
        if (input_reflected) { /* :87-100 */
            poly = reflect(poly);
            uint64_t crc = 1;
            size_t i = 128;
            do {
                crc = (crc & 1) ? (crc >> 1) ^ poly : crc >> 1;
                for (size_t j = 0; j < 256; j += 2 * i) table[i + j] = (crc ^ table[j]) & mask;
                i >>= 1;
            } while (i > 0);
        } else { /* :101-116 */
            const uint64_t msb = uint64_t{ 1 } << (num_bits - 1);
            uint64_t crc = msb;
            size_t i = 1;
            do {
                crc = (crc & msb) ? (crc << 1) ^ poly : crc << 1;
                for (size_t j = 0; j < i; ++j) table[i + j] = (crc ^ table[j]) & mask;
                i <<= 1;
            } while (i < 256);
        }
    }
    uint64_t compute(const uint8_t* data, size_t n) /* :119-156 */
    {
        rem = initial_value;
        if (input_reflected) {
            for (size_t k = 0; k < n; ++k) rem = table[(rem ^ data[k]) & 0xff] ^ (rem >> 8);
        } else {
            for (size_t k = 0; k < n; ++k)
                rem = (table[((rem >> (num_bits - 8)) ^ data[k]) & 0xff] ^ (rem << 8)) & mask;
        }
        if (input_reflected != result_reflected) rem = reflect(rem);
        return rem ^ final_xor;
    }
};
} // namespace
uint64_t orc_crc_compute(const orc_crc_params* p, const uint8_t* data, size_t n)
{
    OrcCrc c(*p);
    return c.compute(data, n);
}
size_t orc_crc_check(const orc_crc_params* p, int swap_endianness, int discard_crc, uint64_t skip_header_bytes,
                     const uint8_t* in, const uint64_t* packet_len, size_t n_packets, uint8_t* out,
                     uint64_t* out_len)
{
    OrcCrc c(*p);
    const size_t crc_bytes = p->num_bits / 8;
    size_t ipos = 0, opos = 0;
    for (size_t k = 0; k < n_packets; ++k) { /* crc_check.hpp:152-208 */
        const size_t len = packet_len[k];
        out_len[k] = 0;
        if (len > crc_bytes) {
            const size_t payload = len - crc_bytes;
            const size_t skip = std::min<size_t>(skip_header_bytes, payload);
            const uint64_t computed = c.compute(in + ipos + skip, payload - skip);
            uint64_t in_packet = 0;
            if (swap_endianness) {
                for (size_t i = len; i-- > payload;) in_packet = (in_packet << 8) | in[ipos + i];
            } else {
                for (size_t i = payload; i < len; ++i) in_packet = (in_packet << 8) | in[ipos + i];
            }
            if (in_packet == computed) {
                const size_t n_out = discard_crc ? payload : len;
                std::copy_n(in + ipos, n_out, out + opos);
                out_len[k] = n_out;
                opos += n_out;
            }
        }
        ipos += len;
    }
    return opos;
}

} /* extern "C" */
