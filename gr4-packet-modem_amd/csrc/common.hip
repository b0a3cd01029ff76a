// common.hip -- error text, device probing and the host-only firdes helper of libgr4pm_hip.so
#include <cmath>
#include <vector>

#include "common.hpp"

namespace gr4pm {

static thread_local char g_error[512] = "";
static thread_local bool g_deferred_sync = false;
bool deferred_sync() { return g_deferred_sync; }
void set_deferred_sync(bool on) { g_deferred_sync = on; }

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

gr4pm_status require_device()
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s): gr4pm has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return GR4PM_ERR_NO_DEVICE;
    }
    return GR4PM_OK;
}

} // namespace gr4pm

extern "C" {

const char* gr4pm_last_error(void) { return gr4pm::g_error; }

void gr4pm_set_deferred_sync(int on) { gr4pm::g_deferred_sync = on != 0; }

const char* gr4pm_version(void) { return "gr4pm-hip 0.1 (gfx950, one-wave FFT-2048 correlator)"; }

int gr4pm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// firdes.hpp:29-76 -- the GR3-equivalent RRC design, evaluated in double and cast
size_t gr4pm_firdes_root_raised_cosine(double gain, double sampling_freq, double symbol_rate,
                                       double alpha, size_t ntaps, float* out)
{
    ntaps |= 1;
    const double pi = 3.14159265358979323846;
    const double spb = sampling_freq / symbol_rate;
    std::vector<double> taps(ntaps);
    double scale = 0.0;
    for (size_t i = 0; i < ntaps; ++i) {
        const double xi = static_cast<double>(static_cast<long>(i) - static_cast<long>(ntaps) / 2);
        const double x1 = pi * xi / spb;
        const double x2 = 4.0 * alpha * xi / spb;
        const double x3 = x2 * x2 - 1.0;
        double num, den;
        if (std::fabs(x3) >= 0.000001) {
            if (i != ntaps / 2)
                num = std::cos((1.0 + alpha) * x1) + std::sin((1.0 - alpha) * x1) / (4.0 * alpha * xi / spb);
            else
                num = std::cos((1.0 + alpha) * x1) + (1.0 - alpha) * pi / (4.0 * alpha);
            den = x3 * pi;
            taps[i] = 4.0 * alpha * num / den;
        } else if (alpha == 1.0) {
            taps[i] = -1.0;
        } else {
            const double a3 = (1.0 - alpha) * x1, a2 = (1.0 + alpha) * x1;
            num = std::sin(a2) * (1.0 + alpha) * pi - std::cos(a3) * ((1.0 - alpha) * pi * spb) / (4.0 * alpha * xi) +
                  std::sin(a3) * spb * spb / (4.0 * alpha * xi * xi);
            den = -32.0 * pi * alpha * alpha * xi / spb;
            taps[i] = 4.0 * alpha * num / den;
        }
        scale += taps[i];
    }
    for (size_t i = 0; i < ntaps; ++i) out[i] = static_cast<float>(taps[i] * gain / scale);
    return ntaps;
}

} // extern "C"
