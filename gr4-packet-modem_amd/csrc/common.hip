// common.hip -- error text, device probing and the host-only firdes helper of libgr4pm_hip.so
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <execinfo.h>
#include <exception>
#include <system_error>
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

const char* experiment_env(const char* name, bool wrong_results)
{
    const char* e = getenv(name);
#ifdef GR4PM_EXPERIMENTS
    if (e && wrong_results)
        fprintf(stderr, "[gr4pm] WARNING: %s=%s is set: a timing experiment that leaves kernels out or replaces them -- "
                        "calls return GR4PM_OK with WRONG outputs\n", name, e);
    return e;
#else
    // the shipped library has no switch that changes results: the timing-only kernels and paths are compiled under
    // -DGR4PM_EXPERIMENTS only (make EXPERIMENTS=1; tools/build_variant.sh)
    if (e && wrong_results) {
        fprintf(stderr, "[gr4pm] %s=%s ignored: timing experiments need a library built with EXPERIMENTS=1\n", name, e);
        return nullptr;
    }
    return e;
#endif
}
unsigned experiment_env_wg(const char* name, unsigned fallback, unsigned lo, unsigned hi)
{
    const char* e = getenv(name);
    if (!e) return fallback;
    const long v = atol(e);
    const unsigned c = static_cast<unsigned>(std::min<long>(std::max<long>(v, lo), hi));
    if (static_cast<long>(c) != v) fprintf(stderr, "[gr4pm] %s=%ld is outside [%u, %u]: using %u\n", name, v, lo, hi, c);
    return c;
}

gr4pm_status exception_status(const char* where) noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory (std::bad_alloc)", where);
        return GR4PM_ERR_NOMEM;
    } catch (const std::system_error& e) {
        set_error("%s: %s (std::system_error %d)", where, e.what(), e.code().value());
        return GR4PM_ERR_INTERNAL;
    } catch (const std::exception& e) {
        set_error("%s: unexpected C++ exception: %s", where, e.what());
        return GR4PM_ERR_INTERNAL;
    } catch (...) {
        set_error("%s: unexpected non-standard exception", where);
        return GR4PM_ERR_INTERNAL;
    }
}

#ifdef GR4PM_TEST_ALLOC_HOOK
// Only in the TEST build of the library (libgr4pm_hip_test.so, `make test_lib`; tests/gr4pm_test_hooks.h): the
// shipped library neither replaces operator new / delete nor exports these entry points, so it is indifferent to
// the host application's allocator (a non-malloc operator new, an ASan host).
// Test-only allocator hook (gr4pm_test_fail_allocations): the library's own operator new -- local to the library (csrc/exports.map), so it
// serves exactly the allocations made by this library's code and nothing else in the process -- counts down and throws
// std::bad_alloc for `count` allocations after letting `after` pass, then disarms itself.  One relaxed atomic load per
// allocation when not armed.
static std::atomic<bool> g_alloc_armed{ false };
static std::atomic<long> g_alloc_pass_left{ 0 }, g_alloc_fail_left{ 0 };
static std::atomic<unsigned long long> g_alloc_calls{ 0 };
#if !defined(__HIP_DEVICE_COMPILE__)
static inline void* hooked_alloc(std::size_t n)
{
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    if (g_alloc_armed.load(std::memory_order_relaxed) && g_alloc_pass_left.fetch_sub(1, std::memory_order_relaxed) <= 0) {
        const long f = g_alloc_fail_left.fetch_sub(1, std::memory_order_relaxed);
        if (f <= 1) g_alloc_armed.store(false, std::memory_order_relaxed);
        if (f > 0) {
            if (getenv("GR4PM_TEST_ALLOC_TRACE")) { // where the injected failure struck (test debugging)
                void* bt[24];
                const int nb = backtrace(bt, 24);
                fprintf(stderr, "[gr4pm] injected allocation failure (%zu bytes) at:\n", n);
                backtrace_symbols_fd(bt, nb, 2);
            }
            return nullptr;
        }
    }
    return std::malloc(n ? n : 1);
}
#endif
#endif // GR4PM_TEST_ALLOC_HOOK

} // namespace gr4pm

#if defined(GR4PM_TEST_ALLOC_HOOK) && !defined(__HIP_DEVICE_COMPILE__)
// (kept out of the dynamic symbol table by csrc/exports.map: the link makes everything but gr4pm_* local)
#define GR4PM_HIDDEN
GR4PM_HIDDEN void* operator new(std::size_t n)
{
    void* p = gr4pm::hooked_alloc(n);
    if (!p) throw std::bad_alloc();
    return p;
}
GR4PM_HIDDEN void* operator new[](std::size_t n)
{
    void* p = gr4pm::hooked_alloc(n);
    if (!p) throw std::bad_alloc();
    return p;
}
GR4PM_HIDDEN void* operator new(std::size_t n, const std::nothrow_t&) noexcept { return gr4pm::hooked_alloc(n); }
GR4PM_HIDDEN void* operator new[](std::size_t n, const std::nothrow_t&) noexcept { return gr4pm::hooked_alloc(n); }
GR4PM_HIDDEN void operator delete(void* p) noexcept { std::free(p); }
GR4PM_HIDDEN void operator delete[](void* p) noexcept { std::free(p); }
GR4PM_HIDDEN void operator delete(void* p, std::size_t) noexcept { std::free(p); }
GR4PM_HIDDEN void operator delete[](void* p, std::size_t) noexcept { std::free(p); }
GR4PM_HIDDEN void operator delete(void* p, const std::nothrow_t&) noexcept { std::free(p); }
GR4PM_HIDDEN void operator delete[](void* p, const std::nothrow_t&) noexcept { std::free(p); }
#endif

extern "C" {

#ifdef GR4PM_TEST_ALLOC_HOOK
void gr4pm_test_fail_allocations(long after, long count)
try {
    gr4pm::g_alloc_armed.store(false);
    if (after < 0) return; // disarm
    gr4pm::g_alloc_fail_left.store(count > 0 ? count : 1);
    gr4pm::g_alloc_pass_left.store(after);
    gr4pm::g_alloc_armed.store(true);
}
GR4PM_ABI_CATCH_VOID
unsigned long long gr4pm_test_allocation_count(void) { return gr4pm::g_alloc_calls.load(std::memory_order_relaxed); }
#endif

const char* gr4pm_last_error(void) { return gr4pm::g_error; }

void gr4pm_set_deferred_sync(int on) { gr4pm::g_deferred_sync = on != 0; }

const char* gr4pm_version(void) { return "gr4pm-hip 0.1 (gfx950, one-wave FFT-2048 correlator)"; }

int gr4pm_device_count(void)
try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
GR4PM_ABI_CATCH_RET(0)

// firdes.hpp:29-76 -- the GR3-equivalent RRC design, evaluated in double and cast
size_t gr4pm_firdes_root_raised_cosine(double gain, double sampling_freq, double symbol_rate,
                                       double alpha, size_t ntaps, float* out)
try {
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
GR4PM_ABI_CATCH_RET(0)

} // extern "C"
