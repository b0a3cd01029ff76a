// fft2048_pair.hpp -- the 2048-point transforms of fft2048_wave.hpp executed by TWO wavefronts
// (128 lanes, 16 points per lane) instead of one.  Same decomposition (FFT-1: 16 x 16 x 8,
// FFT-2: 8 x 16 x 16), same butterflies, same twiddle values, same operation order per output
// element -- a lane of the pair simply owns ONE of the sub-transform groups where a lane of the
// single wave owns two or four (what fft2048_wave.hpp indexes with q / e / h is part of the
// lane number L = 0..127 here).  Results are therefore bit-identical to the one-wave schedules;
// what changes is the register working set (half) and that an exchange crosses the two waves,
// so its store and load phases are separated by a barrier of the pair's workgroup.
//
// Exchanges run in two half-rounds through the same 9 KiB buffer as the one-wave version; in
// each half all 128 lanes store 8 items and the 64 lanes of wave h load 16 (FFT-1's second
// exchange: the other way round).
#pragma once
#include "fft2048_wave.hpp"

namespace gr4pm {

// keeps hipcc from hoisting a whole table of twiddle reads above the butterflies in front of
// them (it trades registers for latency hiding that the other waves of the SIMD provide)
#if defined(__HIP_DEVICE_COMPILE__)
#define GR4PM_PAIR_FENCE() asm volatile("" ::: "memory")
#else
#define GR4PM_PAIR_FENCE() (void)0
#endif

constexpr int kPairLanes = 128;
constexpr int kPairPts = 16;
// per-lane-ordered twiddle tables of the pair schedules:
//   tw1p[(k1-1)][L]          = W2048^(L k1)                                   k1 = 1..15  (15 x 128)
//   twAp[(ka-1)*2 + q][L]    = W2048^(n2 ka), n2 = (L + 128 q)/16 + 16 (L%16), ka = 1..7   (14 x 128)
// tw1b and twB of fft2048_wave.hpp are used as they are.
constexpr int kTw1pItems = 15 * 128, kTwApItems = 14 * 128;
template <typename W>
inline void build_pair_twiddle_tables(W w, cf* tw1p, cf* twAp)
{
    for (int k1 = 1; k1 < 16; ++k1)
        for (int L = 0; L < 128; ++L) tw1p[(k1 - 1) * 128 + L] = w((L * k1) % kFftN);
    for (int ka = 1; ka < 8; ++ka)
        for (int q = 0; q < 2; ++q)
            for (int L = 0; L < 128; ++L) {
                const int n2 = ((L + 128 * q) >> 4) + 16 * (L & 15);
                twAp[((ka - 1) * 2 + q) * 128 + L] = w((n2 * ka) % kFftN);
            }
}

// ======================================================================= FFT-1
// r[n1] = x[L + 128*n1] on entry.
GR4PM_HD void fft1p_pass1(int L, cf* r, const cf* tw1p)
{
    dft16(r);
    GR4PM_PAIR_FENCE();
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) r[k1] = cmul(r[k1], tw1p[(k1 - 1) * 128 + L]);
}
// half h: rows k1 in [8h, 8h+8), every lane stores its 8 items of those rows
GR4PM_HD void fft1p_store1(int L, const cf* r, cf* lds, int h)
{
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) lds[kk * kS1 + L] = r[8 * h + kk];
}
// lanes of wave h (L/64 == h) read their 16 items: k1 = L/8 (row k1 - 8h), m = L%8
GR4PM_HD void fft1p_load2(int L, cf* b, const cf* lds)
{
    const int kk = (L >> 3) & 7, m = L & 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = lds[kk * kS1 + m + 8 * i];
}
GR4PM_HD void fft1p_pass2(int L, cf* b, const cf* tw1b)
{
    const int m = L & 7;
    dft16(b);
    GR4PM_PAIR_FENCE();
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b[k2] = cmul(b[k2], tw1b[(k2 - 1) * 8 + m]);
}
// half h: the lanes of wave h store their 16 items in row L - 64h
GR4PM_HD void fft1p_store2(int L, const cf* b, cf* lds)
{
    const int row = L & 63;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[row * kS2 + k2] = b[k2];
}
// half h: every lane reads the 8 items of its group q = h: k1 = (L + 128h)/16 (local L/16), k2 = L%16
GR4PM_HD void fft1p_load3(int L, cf* r, const cf* lds, int h)
{
    const int k1loc = L >> 4, k2 = L & 15;
#pragma unroll
    for (int m = 0; m < 8; ++m) r[8 * h + m] = lds[(k1loc * 8 + m) * kS2 + k2];
}
// on exit r[8*q + k3] = X[k1 + 16*k2 + 256*k3], (k1, k2) = ((L + 128 q) / 16, L % 16)
GR4PM_HD void fft1p_pass3(cf* r)
{
    dft8(r);
    dft8(r + 8);
}
GR4PM_HD int fft1p_out_index(int L, int j)
{
    const int q = j >> 3, k3 = j & 7;
    return ((L + 128 * q) >> 4) + 16 * (L & 15) + 256 * k3;
}

// ======================================================================= FFT-2
// r[8*q + n1] = P[n2 + 256*n1], n2 = (L + 128 q)/16 + 16*(L%16) on entry.
GR4PM_HD void fft2p_passA(int L, cf* r, const cf* twAp)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        dft8(r + 8 * q);
        GR4PM_PAIR_FENCE();
#pragma unroll
        for (int ka = 1; ka < 8; ++ka) r[8 * q + ka] = cmul(r[8 * q + ka], twAp[((ka - 1) * 2 + q) * 128 + L]);
    }
}
// half h: rows (ka*16 + m) with ka in [4h, 4h+4); every lane stores 8 items
GR4PM_HD void fft2p_storeA(int L, const cf* r, cf* lds, int h)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int m = (L + 128 * q) >> 4, i = L & 15;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) lds[(kk * 16 + m) * kSA + i] = r[8 * q + 4 * h + kk];
    }
}
// the lanes of wave h read their 16 items: row L - 64h  (L = ka*16 + m)
GR4PM_HD void fft2p_loadB(int L, cf* b, const cf* lds)
{
    const int row = L & 63;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = lds[row * kSA + i];
}
GR4PM_HD void fft2p_passB(int L, cf* b, const cf* twB)
{
    const int m = L & 15;
    dft16(b);
    GR4PM_PAIR_FENCE();
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b[k2] = cmul(b[k2], twB[(k2 - 1) * 16 + m]);
}
// half h: rows (k2*8 + ka) with k2 in [8h, 8h+8); every lane stores 8 items (ka = L/16, m = L%16)
GR4PM_HD void fft2p_storeB(int L, const cf* b, cf* lds, int h)
{
    const int ka = L >> 4, m = L & 15;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) lds[(kk * 8 + ka) * kSB + m] = b[8 * h + kk];
}
// the lanes of wave h read their 16 items: row L - 64h  (L = k2*8 + ka)
GR4PM_HD void fft2p_loadC(int L, cf* c, const cf* lds)
{
    const int row = L & 63;
#pragma unroll
    for (int m = 0; m < 16; ++m) c[m] = lds[row * kSB + m];
}
// on exit c[k3] = C[L + 128*k3]
GR4PM_HD void fft2p_passC(cf* c) { dft16(c); }
GR4PM_HD int fft2p_out_index(int L, int j) { return L + 128 * j; }

} // namespace gr4pm
