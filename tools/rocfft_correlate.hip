// tools/rocfft_correlate.hip -- the MULTI-KERNEL overlap-save correlator north_star names ("overlap-save FFT correlation
// on rocFFT"), built only as a measured BASELINE for k_correlate_w64 ("a rocFFT-based multi-kernel
// version to show the fusion win with rocprof").  Never product: libgr4pm_hip.so does not link rocFFT.
//
//   per call (syncword_detection.hpp:238-313 at fft_size 2048, stride 1752):
//     X      = rocFFT forward, batch = blocks, input read in place from the stream (idist = stride: overlapping blocks)
//     per bin b:  P = X .* T_b (k_mul)  ->  C = rocFFT forward of P ("IFFT computed as FFT", hpp:250-251)
//                 zmax = max(zmax, |C|^2) (k_pow_max; the first bin stores)
//     zpow[b S + lag] = zmax[(N - lag) mod N], lag < S   (k_scatter)
//   HBM traffic by construction: X written 16 B/point once, then per bin 16 read + 16 written (product), 16 + 16
//   (transform, if rocFFT needs one pass; a 2048-point transform is one kernel), 16 read + 4 + 4 (power / maximum):
//   ~72 B per point and bin + 36 -> at nine bins ~680 B per POINT, ~800 B per input SAMPLE (2048 / 1752), against the
//   fused kernel's 12.3 B per sample.
//
//   rocfft_correlate.bin [items = 2^26] [bins = 4] [reps = 5]  -> one JSON line (HIP events around the whole call)
// Templates are random unit-magnitude spectra: the timing and the traffic do not depend on their values.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                                  \
        }                                                                              \
    } while (0)
#define RK(x)                                                                          \
    do {                                                                               \
        rocfft_status s_ = (x);                                                        \
        if (s_ != rocfft_status_success) {                                             \
            fprintf(stderr, "%s: rocfft status %d\n", #x, static_cast<int>(s_));       \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

constexpr int N = 2048, L = 297, S = N - L + 1;

__global__ __launch_bounds__(256) void k_mul(const float2* __restrict__ X, const float2* __restrict__ T, float2* __restrict__ P,
                                             size_t n_points)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_points) return;
    const float2 x = X[i], t = T[i & (N - 1)];
    P[i] = make_float2(x.x * t.x - x.y * t.y, x.x * t.y + x.y * t.x); // hpp:247-249
}
__global__ __launch_bounds__(256) void k_pow_max(const float2* __restrict__ C, float* __restrict__ zmax, size_t n_points, int first)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_points) return;
    const float2 c = C[i];
    const float p = c.x * c.x + c.y * c.y; // hpp:307-308
    zmax[i] = first ? p : fmaxf(zmax[i], p);
}
__global__ __launch_bounds__(256) void k_scatter(const float* __restrict__ zmax, float* __restrict__ zpow, size_t n_blocks)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_blocks * S) return;
    const size_t b = i / S, lag = i - b * S;
    zpow[i] = zmax[b * N + ((N - lag) & (N - 1))]; // hpp:300
}

int main(int argc, char** argv)
{
    const size_t items = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 26);
    const int bins = argc > 2 ? atoi(argv[2]) : 4, n_bins = 2 * bins + 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t n_blocks = (items - N) / S + 1, n_points = n_blocks * N;
    float2 *x, *X, *P, *C, *T;
    float *zmax, *zpow;
    CK(hipMalloc(&x, items * sizeof(float2)));
    CK(hipMalloc(&X, n_points * sizeof(float2)));
    CK(hipMalloc(&P, n_points * sizeof(float2)));
    CK(hipMalloc(&C, n_points * sizeof(float2)));
    CK(hipMalloc(&T, static_cast<size_t>(n_bins) * N * sizeof(float2)));
    CK(hipMalloc(&zmax, n_points * sizeof(float)));
    CK(hipMalloc(&zpow, n_blocks * S * sizeof(float)));
    {
        std::vector<float2> h(items);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&] {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            return static_cast<float>(static_cast<double>(s >> 11) / 9007199254740992.0 - 0.5);
        };
        for (auto& v : h) v = make_float2(rnd(), rnd());
        CK(hipMemcpy(x, h.data(), items * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<float2> t(static_cast<size_t>(n_bins) * N);
        for (auto& v : t) v = make_float2(rnd(), rnd());
        CK(hipMemcpy(T, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    RK(rocfft_setup());
    auto make_plan = [&](size_t idist, rocfft_plan* plan, rocfft_execution_info* info, void** work) -> int {
        rocfft_plan_description d = nullptr;
        RK(rocfft_plan_description_create(&d));
        const size_t stride1[1] = { 1 };
        RK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved,
                                                   nullptr, nullptr, 1, stride1, idist, 1, stride1, N));
        const size_t len[1] = { N };
        RK(rocfft_plan_create(plan, rocfft_placement_notinplace, rocfft_transform_type_complex_forward, rocfft_precision_single, 1,
                              len, n_blocks, d));
        RK(rocfft_plan_description_destroy(d));
        size_t wb = 0;
        RK(rocfft_plan_get_work_buffer_size(*plan, &wb));
        RK(rocfft_execution_info_create(info));
        *work = nullptr;
        if (wb) {
            CK(hipMalloc(work, wb));
            RK(rocfft_execution_info_set_work_buffer(*info, *work, wb));
        }
        return 0;
    };
    rocfft_plan fwd, per_bin;
    rocfft_execution_info ifwd, ibin;
    void *wfwd, *wbin;
    if (make_plan(S, &fwd, &ifwd, &wfwd)) return 1;   // overlapping blocks straight from the stream
    if (make_plan(N, &per_bin, &ibin, &wbin)) return 1;
    const unsigned gp = static_cast<unsigned>((n_points + 255) / 256), gs = static_cast<unsigned>((n_blocks * S + 255) / 256);
    auto call = [&]() -> int {
        void* in1[1] = { x };
        void* out1[1] = { X };
        RK(rocfft_execute(fwd, in1, out1, ifwd)); // hpp:239-241
        for (int b = 0; b < n_bins; ++b) {
            hipLaunchKernelGGL(k_mul, dim3(gp), dim3(256), 0, nullptr, X, T + static_cast<size_t>(b) * N, P, n_points);
            void* in2[1] = { P };
            void* out2[1] = { C };
            RK(rocfft_execute(per_bin, in2, out2, ibin)); // hpp:250-251
            hipLaunchKernelGGL(k_pow_max, dim3(gp), dim3(256), 0, nullptr, C, zmax, n_points, b == 0 ? 1 : 0);
        }
        hipLaunchKernelGGL(k_scatter, dim3(gs), dim3(256), 0, nullptr, zmax, zpow, n_blocks);
        return 0;
    };
    if (call()) return 1;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r)
        if (call()) return 1;
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= static_cast<float>(reps);
    const double samples = static_cast<double>(n_blocks) * S;
    printf("{\"what\": \"rocFFT multi-kernel overlap-save correlator (baseline, not product)\", \"fft_size\": %d, \"stride\": %d, "
           "\"bins\": %d, \"items\": %zu, \"blocks\": %zu, \"ms_per_call\": %.4f, \"gsps\": %.2f, \"kernels_per_call\": %d}\n",
           N, S, n_bins, items, n_blocks, ms, samples / ms / 1e6, 2 + 3 * n_bins);
    return 0;
}
