// correlate_4096.hpp -- k_correlate_4096: the overlap-save correlator for fft_size = 4096 (BASELINE configs[4]:
// 1024 taps requested -> 1025, syncword of 1277 samples, stride 2820) on the 16 x 16 x 16 workgroup FFT of
// fft4096_wg.hpp.  One 256-thread workgroup per block, 16 points per thread, spectrum and running maximum in
// registers, templates (natural order, [bin][4096]) read coalesced from L2 -- thread t needs T[t + 256 j].
// Replaces syncword_detection.hpp:238-252,300-313 for that size (the radix-2 LDS kernel k_correlate_generic keeps
// every other power of two).
#pragma once
#include "fft4096_wg.hpp"

namespace gr4pm {
namespace {

__device__ __forceinline__ void f4k_fft(int t, cf* r, cf* lds, const cf* __restrict__ tw1, const cf* tw2)
{
    f4k_pass1(t, r, tw1);
    __syncthreads(); // the previous transform's last reads of the image are done
    f4k_store1(t, r, lds);
    __syncthreads();
    f4k_load2(t, r, lds);
    f4k_pass2(t, r, tw2);
    __syncthreads();
    f4k_store2(t, r, lds);
    __syncthreads();
    f4k_load3(t, r, lds);
    f4k_pass3(r);
}

// grid (n_blocks, n_channels); tmpl: [bin][4096] conjugated template spectra, natural order
__global__ __launch_bounds__(kT4k) void k_correlate_4096(const cf* __restrict__ in, size_t in_stride, uint32_t n_blocks,
                                                         uint32_t stride_s, int n_bins, const cf* __restrict__ tmpl,
                                                         const cf* __restrict__ tw1, const cf* __restrict__ tw2g,
                                                         float* __restrict__ zpow, size_t z_stride)
{
    __shared__ cf lds[kX4kItems];
    __shared__ cf tw2[256];
    const int t = threadIdx.x;
    tw2[t] = tw2g[t];
    const uint32_t b = blockIdx.x;
    const cf* x = in + static_cast<size_t>(blockIdx.y) * in_stride + static_cast<size_t>(b) * stride_s + t;
    float* zo = zpow + static_cast<size_t>(blockIdx.y) * z_stride + static_cast<size_t>(b) * stride_s;
    cf r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = x[256 * j];
    __syncthreads();
    f4k_fft(t, r, lds, tw1, tw2); // hpp:239-241
    cf X[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) X[j] = r[j];
    float zmax[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) zmax[j] = -1.0f; // hpp:303
    for (int bin = 0; bin < n_bins; ++bin) {
        const cf* tb = tmpl + static_cast<size_t>(bin) * kN4k + t;
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = cmul(X[j], tb[256 * j]); // hpp:247-249
        f4k_fft(t, r, lds, tw1, tw2);                                  // hpp:250-251
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float pw = fmaf(r[j].y, r[j].y, r[j].x * r[j].x); // hpp:307-308
            asm("v_max_f32 %0, %1, %2" : "=v"(zmax[j]) : "v"(zmax[j]), "v"(pw));
        }
    }
    // lag k <-> correlation index (N - k) mod N (hpp:300); register j of thread t holds index t + 256 j
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t lag = static_cast<uint32_t>((kN4k - (t + 256 * j)) & (kN4k - 1));
        if (lag < stride_s) zo[lag] = zmax[j];
    }
    (void)n_blocks;
}

} // namespace
} // namespace gr4pm
