// VALU-only rate of the in-lane DFT-32 (fft2048_w64.hpp) at two waves per SIMD: how close does hipcc's
// schedule of the inline-asm packed arithmetic get to the 4-cycles-per-packed-instruction floor?
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -I gr4-packet-modem_amd/csrc tools/dft32_rate.hip -o tools/dft32_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fft2048_w64.hpp"
using namespace gr4pm;
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void k(const cf* in, cf* out, int iters)
{
    cf v[32];
    for (int j = 0; j < 32; ++j) v[j] = in[threadIdx.x + 64 * j];
    for (int it = 0; it < iters; ++it) {
        dft32(v);
#pragma unroll
        for (int j = 0; j < 32; ++j) asm volatile("" : "+v"(v[j]));
    }
    for (int j = 0; j < 32; ++j) out[blockIdx.x * 1024 * 32 + threadIdx.x + 1024 * j] = v[j];
}
int main()
{
    cf *in, *out;
    hipMalloc(&in, 1 << 20);
    hipMalloc(&out, 256 * 1024 * 32 * sizeof(cf));
    hipMemset(in, 0x3c, 1 << 20);
    const int iters = 2000;
    for (int w = 1; w <= 4; ++w) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (w == 1) k<1><<<256, 256>>>(in, out, iters);
            if (w == 2) k<2><<<256, 512>>>(in, out, iters);
            if (w == 3) k<3><<<256, 768>>>(in, out, iters);
            if (w == 4) k<4><<<256, 1024>>>(in, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // per SIMD: w waves x iters transforms
        printf("%d wave(s)/SIMD: %.3f ms, %.1f ns per DFT-32 per SIMD (228 packed + ~0 scalar => %.1f ns at 4 cycles and 2.2 GHz)\n", w, ms,
               ms * 1e6 / (iters * w), 228 * 4 / 2.2);
    }
    return 0;
}
