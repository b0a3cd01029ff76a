// cabi_smoke.cpp -- drives the C ABI (include/gr4pm_hip.h) from plain C++ the way a gr::Block
// wrapper would: Rotator<float> over a device buffer, checked against the closed form of
// test/qa_rotator.cpp:33-43, and InterpolatingFirFilter against a direct convolution.
// Built and run by tests/test_gpu_parity.py::test_c_abi_from_cpp (hipcc, links libgr4pm_hip.so).
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include "gr4pm_hip.h"

#define CHECK(x)                                                              \
    do {                                                                      \
        if ((x) != 0) {                                                       \
            std::printf("FAIL %s: %s\n", #x, gr4pm_last_error());             \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main()
{
    if (gr4pm_device_count() < 1) {
        std::printf("FAIL no device\n");
        return 1;
    }
    const size_t n = 100000;
    std::vector<std::complex<float>> ones(n, { 1.0f, 0.0f }), y(n);
    gr4pm_c64 *din, *dout;
    CHECK(hipMalloc(reinterpret_cast<void**>(&din), n * sizeof(gr4pm_c64)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&dout), 4 * n * sizeof(gr4pm_c64)));
    CHECK(hipMemcpy(din, ones.data(), n * sizeof(gr4pm_c64), hipMemcpyHostToDevice));

    gr4pm_rotator_params rp{ 0, 0.1f, 0, 1, nullptr };
    gr4pm_rotator* rot = nullptr;
    CHECK(gr4pm_rotator_create(&rp, &rot));
    CHECK(gr4pm_rotator_process(rot, din, n, n, dout, nullptr, nullptr, 0));
    CHECK(hipMemcpy(y.data(), dout, n * sizeof(gr4pm_c64), hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t i = 0; i < n; ++i) {
        const double ph = static_cast<double>(0.1f) * static_cast<double>(i);
        worst = std::max(worst, static_cast<double>(std::abs(y[i] - std::complex<float>(std::cos(ph), std::sin(ph)))));
    }
    gr4pm_rotator_destroy(rot);
    if (!(worst < 5e-4)) { // qa_rotator.cpp:40
        std::printf("FAIL rotator error %g\n", worst);
        return 1;
    }

    // invalid settings come back as negative status + message (the reference throws)
    gr4pm_interp_fir_params bad{ 0, nullptr, 0, 0, nullptr };
    float tap = 1.0f;
    bad.taps = &tap;
    bad.n_taps = 1;
    gr4pm_interp_fir* f = nullptr;
    if (gr4pm_interp_fir_create(&bad, &f) >= 0) {
        std::printf("FAIL interpolation 0 accepted\n");
        return 1;
    }
    std::vector<float> taps = { 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f };
    gr4pm_interp_fir_params ip{ 3, taps.data(), taps.size(), 0, nullptr };
    CHECK(gr4pm_interp_fir_create(&ip, &f));
    std::vector<std::complex<float>> x(1000), out(3000);
    for (size_t i = 0; i < x.size(); ++i) x[i] = { static_cast<float>(i % 7) - 3.f, static_cast<float>(i % 5) };
    CHECK(hipMemcpy(din, x.data(), x.size() * sizeof(gr4pm_c64), hipMemcpyHostToDevice));
    CHECK(gr4pm_interp_fir_process(f, din, x.size(), dout));
    CHECK(hipMemcpy(out.data(), dout, out.size() * sizeof(gr4pm_c64), hipMemcpyDeviceToHost));
    for (size_t o = 0; o < out.size(); ++o) { // zero-stuffed convolution, integer valued -> exact
        std::complex<float> acc = 0;
        for (size_t k = 0; k < taps.size(); ++k) {
            const long idx = static_cast<long>(o) - static_cast<long>(k);
            if (idx >= 0 && idx % 3 == 0) acc += taps[k] * x[idx / 3];
        }
        if (acc != out[o]) {
            std::printf("FAIL interp fir at %zu\n", o);
            return 1;
        }
    }
    gr4pm_interp_fir_destroy(f);
    std::printf("OK %s\n", gr4pm_version());
    return 0;
}
