// overlap_save_pattern.hip -- what the overlap-save ACCESS PATTERN of the syncword correlator reaches with no transform at
// all: every block reads 2048 complex samples (16 KiB) of which the next block re-reads the last
// 296 (stride 1752, syncword_detection.hpp:238-252) and writes one 4-byte power per lag < stride (hpp:300-313).  HBM sees
// 8 B read + 4 B written per sample; the fabric sees 8 x 2048 / 1752 + 4 = 13.35 B per sample.
//
// Forms (one wave owns one block, as in k_correlate_w64 / k_correlate_w64_one; nothing but one FMA per sample):
//   ld8   : lane l loads x[l + 64 j], j = 0 .. 31        (global_load_dwordx2, 512 B per wave instruction) -- the kernels' form
//   ld16  : lane l loads x[2 l + 128 j] and its neighbour (global_load_dwordx4, 1 KiB per wave instruction), j = 0 .. 15
//   st4   : lane l stores lag 2048 - l - 64 j            (global_store_dword, 256 B per wave instruction, descending lanes)
//   st4a  : the same ascending (lane l stores lag base + l)
//   st16  : lane l stores four consecutive lags           (global_store_dwordx4, 1 KiB per wave instruction)
// plus `copy`: a plain float4 streaming kernel with the same HBM bytes (8 B read + 4 B written per sample) as the ceiling of
// the traffic alone.  Waves per CU and the nt policy are swept.  Standalone: hipcc --offload-arch=gfx950 -O3, no torch.
//
//   ./overlap_save_pattern.bin [log2 samples = 28] [rounds = 20]
// `bpwN`: workgroups of WAVES x N blocks (round 5's launch: the last round of workgroups is partly empty);
// `balanced`: even shares over ROUNDS x CUs workgroups (round 6).
// prints one line per form: ms per launch, TB/s of ALGORITHMIC bytes (12 B per sample), Gsamples/s.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                  \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

constexpr int kN = 2048;
constexpr uint32_t kStride = 1752;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <typename T> struct native { typedef T type; };
template <> struct native<float2> { typedef v2f type; };
template <> struct native<float4> { typedef v4f type; };
template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p)
{
    typedef typename native<T>::type V;
    const V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p)) : *reinterpret_cast<const V*>(p);
    return *reinterpret_cast<const T*>(&v);
}
template <bool NT, typename T>
__device__ __forceinline__ void st(T* p, T v)
{
    typedef typename native<T>::type V;
    if (NT)
        __builtin_nontemporal_store(*reinterpret_cast<const V*>(&v), reinterpret_cast<V*>(p));
    else
        *reinterpret_cast<V*>(p) = *reinterpret_cast<const V*>(&v);
}

// LD: 8 or 16 bytes per lane and load; ST: 0 = dword descending (the kernels'), 1 = dword ascending, 2 = dwordx4
template <int LD, int ST, bool NTL, bool NTS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_pattern(const float2* __restrict__ in, float* __restrict__ zpow,
                                                       uint32_t n_blocks, uint32_t blocks_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blocks_per_wave == 0: an even split of the blocks over the grid (the host makes it a multiple of the CU count)
    uint32_t item = blocks_per_wave ? blockIdx.x * WAVES * blocks_per_wave + wave
                                    : static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * n_blocks / gridDim.x) + wave;
    const uint32_t end = blocks_per_wave ? min(n_blocks, (blockIdx.x + 1) * WAVES * blocks_per_wave)
                                         : static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * n_blocks / gridDim.x);
    if (item >= end) return;
    float2 X[32];
    auto load = [&](uint32_t b) {
        const float2* x = in + static_cast<size_t>(b) * kStride;
        if (LD == 8) {
#pragma unroll
            for (int j = 0; j < 32; ++j) X[j] = ld<NTL>(x + lane + 64 * j);
        } else {
            const float4* x4 = reinterpret_cast<const float4*>(x) + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 v = ld<NTL>(x4 + 64 * j);
                X[2 * j] = float2{ v.x, v.y }, X[2 * j + 1] = float2{ v.z, v.w };
            }
        }
    };
    load(item);
    for (;;) {
        float pw[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) pw[j] = fmaf(X[j].y, X[j].y, X[j].x * X[j].x);
        const uint32_t next = item + WAVES;
        const bool has_next = next < end;
        if (has_next) load(next);
        float* zo = zpow + static_cast<size_t>(item) * kStride;
        if (ST == 0) {
            float* zl = zo + (kN - lane);
#pragma unroll
            for (int j = 5; j < 32; ++j) st<NTS>(zl - 64 * j, pw[j]); // lags 1 .. 1728: every lane (k_correlate_w64's whole rows)
            if (static_cast<uint32_t>(kN - 256 - lane) < kStride) st<NTS>(zl - 256, pw[4]);
            if (lane == 0) st<NTS>(zo, pw[0]);
        } else if (ST == 1) {
#pragma unroll
            for (int j = 0; j < 27; ++j) st<NTS>(zo + lane + 64 * j, pw[j]);
            if (lane + 64 * 27 < kStride) st<NTS>(zo + lane + 64 * 27, pw[27]);
        } else {
            float4* z4 = reinterpret_cast<float4*>(zo) + lane; // 1752 = 438 float4 = 6 x 64 + 54
#pragma unroll
            for (int j = 0; j < 6; ++j) st<NTS>(z4 + 64 * j, float4{ pw[4 * j], pw[4 * j + 1], pw[4 * j + 2], pw[4 * j + 3] });
            if (lane < 54) st<NTS>(z4 + 384, float4{ pw[24], pw[25], pw[26], pw[27] });
        }
        if (!has_next) break;
        item = next;
    }
}

// the traffic alone: 8 B read + 4 B written per sample, float4 both ways, grid-stride
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n_out4)
{
    // out float4 i (4 powers) <- in float4 2 i, 2 i + 1 (4 samples)
    for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n_out4; i += gridDim.x * 256ul) {
        const float4 a = ld<NT>(in + 2 * i), b = ld<NT>(in + 2 * i + 1);
        st<NT>(out + i, float4{ fmaf(a.y, a.y, a.x * a.x), fmaf(a.w, a.w, a.z * a.z), fmaf(b.y, b.y, b.x * b.x),
                                fmaf(b.w, b.w, b.z * b.z) });
    }
}

struct Row {
    const char* name;
    float ms_med, ms_min;
};

int main(int argc, char** argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    const size_t n = size_t(1) << lg;
    const uint32_t n_blocks = static_cast<uint32_t>((n - kN) / kStride + 1);
    const size_t used = size_t(n_blocks) * kStride;
    float2* in;
    float* z;
    CK(hipMalloc(&in, (n + kN) * sizeof(float2)));
    CK(hipMalloc(&z, (used + kN) * sizeof(float)));
    CK(hipMemset(in, 0, (n + kN) * sizeof(float2)));
    CK(hipMemset(z, 0, (used + kN) * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs; 2^%d samples, %u blocks of %d at stride %u; %d launches per form after 10 warm ones\n", prop.name, cus,
           lg, n_blocks, kN, kStride, rounds);
    printf("# TB/s = 12 B per sample (8 read + 4 written: the HBM bytes) / time; fabric bytes are 13.35 per sample\n");
    printf("%-44s %9s %9s %8s %8s\n", "form", "ms median", "ms min", "TB/s", "Gsps");
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 10; ++i) launch();
        std::vector<float> t;
        for (int i = 0; i < rounds; ++i) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        CK(hipGetLastError());
        std::sort(t.begin(), t.end());
        const float med = t[t.size() / 2];
        printf("%-44s %9.4f %9.4f %8.3f %8.1f\n", name, med, t[0], used * 12.0 / med * 1e-9, used / med * 1e-6);
        fflush(stdout);
    };
#define PAT(LD, ST, NTL, NTS, W, BPW)                                                                                  \
    run("ld" #LD " st" #ST " ntl" #NTL " nts" #NTS " waves" #W " bpw" #BPW, [&] {                                      \
        const uint32_t per_wg = W * BPW;                                                                               \
        hipLaunchKernelGGL((k_pattern<LD, ST, NTL, NTS, W>), dim3((n_blocks + per_wg - 1) / per_wg), dim3(W * 64), 0, 0, in, z,   \
                           n_blocks, BPW);                                                                             \
    })
    // BAL: the grid is ROUNDS x CUs workgroups with even shares (what the library launches since round 6)
#define BAL(LD, ST, W, ROUNDS)                                                                                        \
    run("ld" #LD " st" #ST " waves" #W " balanced, " #ROUNDS " rounds of one workgroup per CU", [&] {                 \
        hipLaunchKernelGGL((k_pattern<LD, ST, false, false, W>), dim3(cus * ROUNDS), dim3(W * 64), 0, 0, in, z, n_blocks, 0); \
    })
    run("copy float4 (8 B in + 4 B out per sample)", [&] {
        hipLaunchKernelGGL(k_copy<false>, dim3(cus * 8), dim3(256), 0, 0, reinterpret_cast<const float4*>(in),
                           reinterpret_cast<float4*>(z), used / 4);
    });
    run("copy float4 nt", [&] {
        hipLaunchKernelGGL(k_copy<true>, dim3(cus * 8), dim3(256), 0, 0, reinterpret_cast<const float4*>(in),
                           reinterpret_cast<float4*>(z), used / 4);
    });
    // the kernels' own form and occupancy: 12 waves per CU (k_correlate_w64_one), 8 (k_correlate_w64)
    PAT(8, 0, false, false, 12, 6);
    PAT(8, 0, false, false, 8, 6);
    PAT(8, 0, false, false, 16, 6);
    PAT(8, 0, true, false, 12, 6);
    PAT(8, 0, false, true, 12, 6);
    PAT(8, 0, true, true, 12, 6);
    PAT(8, 1, false, false, 12, 6);
    PAT(8, 2, false, false, 12, 6);
    // 16-byte loads
    PAT(16, 0, false, false, 12, 6);
    PAT(16, 0, false, false, 8, 6);
    PAT(16, 0, false, false, 16, 6);
    PAT(16, 0, true, false, 12, 6);
    PAT(16, 0, false, true, 12, 6);
    PAT(16, 0, true, true, 12, 6);
    PAT(16, 1, false, false, 12, 6);
    PAT(16, 2, false, false, 12, 6);
    PAT(16, 2, true, true, 12, 6);
    PAT(16, 2, false, false, 16, 6);
    PAT(16, 2, false, false, 8, 6);
    // blocks per wave (how long a wave lives)
    PAT(16, 0, false, false, 12, 2);
    PAT(16, 0, false, false, 12, 16);
    PAT(8, 0, false, false, 12, 2);
    PAT(8, 0, false, false, 12, 16);
    BAL(8, 0, 12, 8);
    BAL(8, 0, 12, 4);
    BAL(8, 0, 12, 16);
    BAL(8, 0, 8, 12);
    BAL(16, 0, 12, 8);
    BAL(16, 2, 12, 8);
    return 0;
}
