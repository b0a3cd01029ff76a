// fft4096_wg.hpp -- 4096-point complex FFT by one 256-thread workgroup, 16 points per thread, three in-register
// DFT-16 passes and two LDS exchanges (BASELINE configs[4]: fft_size 4096, syncword_detection.hpp:133 keeps
// fft_size a free setting).  4096 = 16 x 16 x 16, self-sorting:
//
//   distribution (input AND output): register j of thread t holds v[t + 256 j]
//   pass 1   DFT-16 over j, then x W4096^(t k1)                           (thread t, output k1)
//   exch. 1  thread u = k1 + 16 t0 gathers C[t0 + 16 t1][k1], t1 = 0..15
//   pass 2   DFT-16 over t1, then x W256^(t0 k2)                          (output k2)
//   exch. 2  thread v = k1 + 16 k2 gathers D[k1][t0][k2], t0 = 0..15
//   pass 3   DFT-16 over t0 -> X[k1 + 16 k2 + 256 k3] in register k3 of thread k1 + 16 k2
//
// n = t + 256 j, k = k1 + 16 k', k' = k2 + 16 k3, t = t0 + 16 t1:
//   W4096^(nk) = W16^(j k1) W4096^(t k1) W256^(t k') and W256^(t k') = W16^(t1 k2) W256^(t0 k2) W16^(t0 k3).
// The phases are plain functions of (thread, registers, LDS image) and compile for the host (tests/fft4096_emu.cpp).
#pragma once
#include "fft2048_wave.hpp"

namespace gr4pm {

constexpr int kN4k = 4096, kT4k = 256;
constexpr int kS4k = 17;                    // row stride (complex items) of both exchange images: 16 + 1 pad
constexpr int kX4kItems = 256 * kS4k;       // 4352 complex = 34 KiB
// tw1[k1 * 256 + t] = W4096^(t k1), tw2[k2 * 16 + t0] = W256^(t0 k2)
template <typename W>
inline void build_4096_tables(W w, cf* tw1, cf* tw2)
{
    for (int k1 = 0; k1 < 16; ++k1)
        for (int t = 0; t < 256; ++t) tw1[k1 * 256 + t] = w((t * k1) % kN4k);
    for (int k2 = 0; k2 < 16; ++k2)
        for (int t0 = 0; t0 < 16; ++t0) tw2[k2 * 16 + t0] = w((16 * t0 * k2) % kN4k);
}
GR4PM_HD void f4k_pass1(int t, cf* r, const cf* tw1)
{
    dft16(r);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) r[k1] = cmul(r[k1], tw1[k1 * 256 + t]);
}
GR4PM_HD void f4k_store1(int t, const cf* r, cf* lds)
{
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) lds[t * kS4k + k1] = r[k1];
}
GR4PM_HD void f4k_load2(int u, cf* r, const cf* lds)
{
    const int k1 = u & 15, t0 = u >> 4;
#pragma unroll
    for (int t1 = 0; t1 < 16; ++t1) r[t1] = lds[(t0 + 16 * t1) * kS4k + k1];
}
GR4PM_HD void f4k_pass2(int u, cf* r, const cf* tw2)
{
    const int t0 = u >> 4;
    dft16(r);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) r[k2] = cmul(r[k2], tw2[k2 * 16 + t0]);
}
GR4PM_HD void f4k_store2(int u, const cf* r, cf* lds) // row (k1, t0) = u
{
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[u * kS4k + k2] = r[k2];
}
GR4PM_HD void f4k_load3(int v, cf* r, const cf* lds)
{
    const int k1 = v & 15, k2 = v >> 4;
#pragma unroll
    for (int t0 = 0; t0 < 16; ++t0) r[t0] = lds[(k1 + 16 * t0) * kS4k + k2];
}
GR4PM_HD void f4k_pass3(cf* r) { dft16(r); }
GR4PM_HD int f4k_index(int t, int j) { return t + 256 * j; }

} // namespace gr4pm
