// correlate_4096.hpp -- k_correlate_4096: the overlap-save correlator for fft_size = 4096 (BASELINE configs[4]:
// 1024 taps requested -> 1025, syncword of 1277 samples, stride 2820) on the 16 x 16 x 16 workgroup FFT of
// fft4096_wg.hpp.  One 256-thread workgroup per block, 16 points per thread, spectrum and running maximum in
// registers, templates (natural order, [bin][4096]) read coalesced from L2 -- thread t needs T[t + 256 j].
// Replaces syncword_detection.hpp:238-252,300-313 for that size (the radix-2 LDS kernel k_correlate_generic keeps
// every other power of two).
#pragma once
#include "fft2048_w64.hpp"
#include "fft4096_wg.hpp"

namespace gr4pm {
namespace {

// VAR (GR4PM_C4096_VARIANT when the handle is created; 0 - 7 bit-identical; measured per 2^26 samples on one box, round 4,
// tools/c4096_variants.py: 9 bins 1.221 / 1.014 / 1.468 / 1.112 / 1.455 / 1.273 / 1.387 / 1.217 ms for 0 .. 7, one bin
// 0.307 / 0.249 / 0.463 / 0.314 / 0.418 / 0.374 / 0.424 / 0.384):
//   1  (DEFAULT) the fifteen pass-1 twiddles W4096^(t k1) of the thread live in registers for the whole block (they
//      are the same for the forward transform and for every bin) instead of fifteen L1 / L2 loads per transform; hipcc
//      then needs 128 VGPRs instead of 158 (no per-transform address arithmetic): four waves per SIMD instead of three
//   2  the next bin's sixteen template values are requested before the current bin's transform starts
//   4  two exchange images (pass 1 -> 2 through A, pass 2 -> 3 through B): the two barriers that only protect an
//      image against being overwritten while the transform before is still reading it go away (2 per transform, not 4)
//      -- but 70 KiB of LDS per workgroup leave two workgroups per CU: slower
//   9  variant 1 with the FMA-form DFT-16 of fft2048_w64.hpp: 314 instead of 326 packed instructions per bin, powers
//      differ in the last bits; not adopted
template <int VAR>
__device__ __forceinline__ void f4k_fft(int t, cf* r, cf* lds, cf* ldsB, const cf* __restrict__ tw1, const cf* tw1r,
                                        const cf* tw2)
{
    if (VAR & 8) { // FMA-form butterflies (fft2048_w64.hpp: dft16f), the registers of variant 1; powers differ in the last bits
        dft16f(r);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) r[k1] = cmul(r[k1], tw1r[k1]);
    } else if (VAR & 1) {
        dft16(r);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) r[k1] = cmul(r[k1], tw1r[k1]);
    } else {
        f4k_pass1(t, r, tw1);
    }
    if (!(VAR & 4)) __syncthreads(); // the previous transform's last reads of the image are done
    f4k_store1(t, r, lds);
    __syncthreads();
    f4k_load2(t, r, lds);
    if (VAR & 8) {
        dft16f(r);
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) r[k2] = cmul(r[k2], tw2[k2 * 16 + (t >> 4)]);
    } else {
        f4k_pass2(t, r, tw2);
    }
    if (!(VAR & 4)) __syncthreads();
    f4k_store2(t, r, (VAR & 4) ? ldsB : lds);
    __syncthreads();
    f4k_load3(t, r, (VAR & 4) ? ldsB : lds);
    if (VAR & 8) dft16f(r);
    else f4k_pass3(r);
}

// grid (n_blocks, n_channels); tmpl: [bin][4096] conjugated template spectra, natural order
template <int VAR>
__global__ __launch_bounds__(kT4k) void k_correlate_4096(const cf* __restrict__ in, size_t in_stride, uint32_t n_blocks,
                                                         uint32_t stride_s, int n_bins, const cf* __restrict__ tmpl,
                                                         const cf* __restrict__ tw1, const cf* __restrict__ tw2g,
                                                         float* __restrict__ zpow, size_t z_stride, uint32_t noise_rel)
{
    __shared__ cf lds[(VAR & 4) ? 2 * kX4kItems : kX4kItems];
    __shared__ cf tw2[256];
    __shared__ float noise_part[kT4k / 64];
    cf* ldsB = lds + ((VAR & 4) ? kX4kItems : 0);
    const int t = threadIdx.x;
    tw2[t] = tw2g[t];
    const uint32_t b = blockIdx.x;
    const cf* x = in + static_cast<size_t>(blockIdx.y) * in_stride + static_cast<size_t>(b) * stride_s + t;
    float* zo = zpow + static_cast<size_t>(blockIdx.y) * z_stride + static_cast<size_t>(b) * stride_s;
    cf r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = x[256 * j];
    cf tw1r[16];
    if (VAR & 9) {
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) tw1r[k1] = tw1[k1 * 256 + t];
    }
    __syncthreads();
    f4k_fft<VAR>(t, r, lds, ldsB, tw1, tw1r, tw2); // hpp:239-241
    cf X[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) X[j] = r[j];
    if (noise_rel) {
        // hpp:257-265: the block's noise power is the energy of the spectrum's middle half, bins N/4 .. 3N/4 - 1 =
        // registers 4 .. 11 of every thread.  One float per block behind the channel's powers, unnormalised, as
        // k_correlate_w64 leaves it: k_tags reads it for the few blocks that hold a detection (k_tags_generic
        // transformed every such block again, with all its bins: 1.9 of 8.3 ms per 2^28 samples at nine bins).
        float e = 0.0f;
#pragma unroll
        for (int j = 4; j < 12; ++j) e = fmaf(X[j].y, X[j].y, fmaf(X[j].x, X[j].x, e));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) e += __shfl_xor(e, d);
        if ((t & 63) == 0) noise_part[t >> 6] = e;
        __syncthreads();
        if (t == 0) {
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < kT4k / 64; ++w) sum += noise_part[w];
            zpow[static_cast<size_t>(blockIdx.y) * z_stride + noise_rel + 1 + b] = sum;
        }
    }
    float zmax[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) zmax[j] = -1.0f; // hpp:303
    cf tn[16];
    if (VAR & 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) tn[j] = tmpl[t + 256 * j];
    }
    for (int bin = 0; bin < n_bins; ++bin) {
        const cf* tb = tmpl + static_cast<size_t>(bin) * kN4k + t;
        if (VAR & 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = cmul(X[j], tn[j]); // hpp:247-249
            const cf* tnext = tmpl + static_cast<size_t>(min(bin + 1, n_bins - 1)) * kN4k + t;
#pragma unroll
            for (int j = 0; j < 16; ++j) tn[j] = tnext[256 * j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = cmul(X[j], tb[256 * j]); // hpp:247-249
        }
        f4k_fft<VAR>(t, r, lds, ldsB, tw1, tw1r, tw2); // hpp:250-251
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float pw = fmaf(r[j].y, r[j].y, r[j].x * r[j].x); // hpp:307-308
            asm("v_max_f32 %0, %1, %2" : "=v"(zmax[j]) : "v"(zmax[j]), "v"(pw));
        }
    }
    // lag k <-> correlation index (N - k) mod N (hpp:300); register j of thread t holds index t + 256 j
    // Round 5 (as in k_correlate_w64): register j >= 1 holds the lags 4096 - 256 j - t <= 4096 - 256 j, so every register
    // from jw = (4096 - stride_s) / 256 + 1 on is stored by every thread (configs[4]: stride 2820, jw = 5), register
    // jw - 1 by the threads whose lag is below the stride, register 0 by thread 0 (lag 0) and nothing in between: one
    // uniform branch per block instead of a lane compare, an exec mask and a branch around each of the 16 stores.
    const uint32_t jw = (static_cast<uint32_t>(kN4k) - stride_s) / 256u + 1u;
    if (jw == 5u && !(VAR & 16)) { // (VAR & 16: round 4's stores, for A/B)
        if (t == 0) zo[0] = zmax[0];
        if (static_cast<uint32_t>(kN4k - 1024 - t) < stride_s) zo[kN4k - 1024 - t] = zmax[4];
#pragma unroll
        for (int j = 5; j < 16; ++j) zo[kN4k - (t + 256 * j)] = zmax[j];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t lag = static_cast<uint32_t>((kN4k - (t + 256 * j)) & (kN4k - 1));
            if (lag < stride_s) zo[lag] = zmax[j];
        }
    }
    (void)n_blocks;
}

} // namespace
} // namespace gr4pm
