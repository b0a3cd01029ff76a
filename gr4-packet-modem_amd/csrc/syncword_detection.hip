// syncword_detection.hip -- MI355X implementation of gr::packet_modem::SyncwordDetection
// (reference: syncword_detection.hpp:32-357) behind the C ABI of include/gr4pm_hip.h.
//
// Data flow of one process() call (all kernels on the handle's stream, nothing leaves HBM
// except the sparse tag records):
//
//   k_correlate_w64  overlap-save correlator (correlate_w64.hpp): one 64-lane wave per 2048-sample block; FFT,
//                    x B templates, FFT, |.|^2, max over bins, all in registers/LDS; reads
//                    8 B/sample, writes the 4 B/sample correlation power `zpow`
//                    (hpp:238-252,300-313).  k_correlate / k_correlate_pair (round 1, below),
//                    k_correlate_4096 (fft_size 4096), k_correlate_generic (other sizes)
//   k_candidates_wave  B(p) = zpow[p] >= max(zpow[p+1..p+T]) as a bitmap (sliding max in registers, T = 768;
//                    k_candidates with LDS for other T)
//   k_tile_tables    for every tile and every possible entry point, where the greedy peak
//   k_group_tables   scan of hpp:267-298,314-317 leaves the tile (then a group of tiles); then
//   k_scan_entries   the entry point of each tile; then the candidates the scan visits per tile
//   k_tile_visit     and, for all of them in parallel, the median test (hpp:273-295)
//   k_median_tests
//   k_tags           for every detection that is emitted in this call: noise power from the transform of its
//                    block, the correlation at its lag for every bin from its definition; writes a raw record
//   k_compact_pending  drops the emitted detections, channel state to pinned host memory
//   k_delay_copy     out[i] = in[i - (2T+1)] (hpp:318-319,342); skipped when the caller reads the input in place
//
// Why the detector can be parallel although the reference is a sequential greedy scan:
// with r the item after the last reset (hpp:296-297), the next item whose history is tested
// is the FIRST p >= r with zpow[p] >= max(zpow[p+1..p+T]) ("candidate"), and the following
// reset happens at p+T+1 (proof in HISTORY.md section 4).  Candidates are a per-item predicate; the
// scan is a monotone map r -> r' that composes tile by tile.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <deque>
#include <string>
#include <vector>

#include "common.hpp"
#include "fft2048_wave.hpp"
#include "fft2048_pair.hpp"
#include "fft2048_w64.hpp"

#ifndef GR4PM_ABL
#define GR4PM_ABL 0
#endif

namespace gr4pm {

namespace {

constexpr int kWavesPerWg = 4; // 256 threads; two workgroups per CU (68 KiB LDS each) = 2 waves per SIMD
constexpr int kMaxBins = 64;
#ifndef GR4PM_TILE_W
#define GR4PM_TILE_W 32768
#endif
constexpr uint32_t kTileW = GR4PM_TILE_W; // items per detector tile
// look-ahead depth: fronts (correlator, candidates, tables) of up to kAhead future calls in flight
constexpr int kAhead = 2, kSets = kAhead + 1, kCarry = kAhead + 2;

struct RawTag { // device -> host
    uint64_t pos; // absolute item index of the detection
    float zx, zy; // correlation at pos, best bin
    float zpow, left, right, prev, next, noise;
    int32_t bin_idx; // 0-based
    int32_t pad;
};

__device__ __forceinline__ void wave_lds_sync()
{
    // The LDS pipeline executes the DS instructions of one wave in issue order, so a read of
    // another lane's write needs no hardware wait, only the guarantee that the COMPILER keeps
    // the program order.  (A wavefront-scope fence would also cover global memory and make
    // hipcc drain the prefetched template loads with s_waitcnt vmcnt(0) at every exchange.)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// FFT-1: natural-order block in r (see fft1_pass1) -> X in FFT-1's output distribution.
// Exchanges run in two half-rounds through the wave's 9 KiB LDS buffer.
__device__ __forceinline__ void fft1_wave(int lane, cf* r, cf* lds, const cf* tw1a, const cf* tw1b)
{
    fft1_pass1(lane, r, tw1a);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wave_lds_sync();
        fft1_store1(lane, r, lds, h);
        wave_lds_sync();
        fft1_load2(lane, r, lds, h);
    }
    fft1_pass2(lane, r, tw1b);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wave_lds_sync();
        fft1_store2(lane, r, lds, h);
        wave_lds_sync();
        fft1_load3(lane, r, lds, h);
    }
    fft1_pass3(r);
}
// FFT-2: r (FFT-1's distribution) -> out (consecutive indices on consecutive lanes)
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <typename Hook = NoHook>
__device__ __forceinline__ void fft2_wave(int lane, cf* r, cf* out, cf* lds, const cf* twA, const cf* twB,
                                          Hook mid = Hook{})
{
    fft2_passA(lane, r, twA);
    cf b[32];
#if GR4PM_ABL == 1 || GR4PM_ABL == 3 || GR4PM_ABL == 4 /* timing-only ablation: no LDS exchanges */
#pragma unroll
    for (int j = 0; j < 32; ++j) b[j] = r[(j * 5 + 3) & 31];
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wave_lds_sync();
        fft2_storeA(lane, r, lds, h);
        wave_lds_sync();
        fft2_loadB(lane, b, lds, h);
    }
#endif
    fft2_passB(lane, b, twB);
    mid(); // the caller's mid-transform work (template hand-off in k_correlate)
#if GR4PM_ABL == 1 || GR4PM_ABL == 3 || GR4PM_ABL == 4
#pragma unroll
    for (int j = 0; j < 32; ++j) out[j] = b[(j * 7 + 1) & 31];
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wave_lds_sync();
        fft2_storeB(lane, b, lds, h);
        wave_lds_sync();
        fft2_loadC(lane, out, lds, h);
    }
#endif
    fft2_passC(out);
}

// templates are stored per bin as [16][64 lanes] float4 = (T[k(lane, 2jp)], T[k(lane, 2jp+1)])
// with k = fft1_out_index(lane, j): each lane reads its 32 values with 16 conflict-free /
// coalesced 16-byte reads (from the LDS copy in k_correlate, from global in k_tags).
__device__ __forceinline__ void mul_template(int lane, const cf* X, cf* r, const float4* tp)
{
#pragma unroll
    for (int jp = 0; jp < 16; ++jp) {
        const float4 t = tp[jp * 64 + lane];
        r[2 * jp] = cmul(X[2 * jp], mk(t.x, t.y));
        r[2 * jp + 1] = cmul(X[2 * jp + 1], mk(t.z, t.w));
    }
}

// LDS map of k_correlate (one array, 16-byte aligned), in float4 units:
//   [0, 1024)              the current template (16 KiB)
//   [1024, 1024+896)       twA  (28 x 64 cf = 14 KiB)
//   [1920, 1920+120)       twB  (15 x 16 cf = 1.9 KiB)
//   [2040, 2040+4*576)     four per-wave exchange buffers of 1152 cf (9 KiB) each
// = 68 KiB per workgroup -> two workgroups per CU.  They run out of phase, so one's prologue
// (HBM loads, FFT-1) and barriers are covered by the other's transforms.
constexpr int kLdsTmpl = 0, kLdsTwA = 1024, kLdsTwB = kLdsTwA + kTwAItems / 2, kLdsExch = kLdsTwB + kTwBItems / 2;
constexpr int kLdsCtl = kLdsExch + kWavesPerWg * (kExchangeItems / 2);   // one float4: hand-off words
constexpr int kLdsTotal = kLdsCtl + 1;                                  // float4 units
static_assert(2 * kLdsTotal * 16 <= 160 * 1024, "LDS budget for two workgroups per CU");
constexpr int kCorrThreads = kWavesPerWg * 64;
constexpr int kTmplPerThread = 1024 / kCorrThreads; // float4 per thread per template

// ------------------------------------------------------------------ k_correlate
// grid (ceil(n_blocks / 4), n_channels), 256 threads = 4 waves, one overlap-save block per
// wave; the waves walk the B templates together so each template is staged into LDS once per
// workgroup (global loads issued a whole transform ahead, written to LDS between two barriers).
__global__ __launch_bounds__(kCorrThreads, 2) void k_correlate(const cf* __restrict__ in, size_t in_stride,
                                                               uint32_t n_blocks, uint32_t stride_s,
                                                               int n_bins, const float4* __restrict__ tmpl,
                                                               const cf* __restrict__ tw1a,
                                                               const cf* __restrict__ tw1b,
                                                               const float4* __restrict__ twAB,
                                                               float* __restrict__ zpow, size_t z_stride,
                                                               unsigned* __restrict__ fault, int spin_limit)
{
    __shared__ float4 lds4[kLdsTotal];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t b_raw = blockIdx.x * kWavesPerWg + wave;
    const bool active = b_raw < n_blocks;
    const uint32_t b = active ? b_raw : n_blocks - 1; // idle waves shadow the last block
    cf* lds = reinterpret_cast<cf*>(lds4 + kLdsExch) + wave * kExchangeItems;
#if defined(GR4PM_TW_L1) /* measured, not adopted: FFT-2 twiddles through L1 instead of LDS (22 % of the LDS
                            traffic): hipcc hoists the loads, 256 VGPRs + spills, 1.42 ms instead of 1.01;
                            pass-A table only (GR4PM_TWA_L1): 1.29 ms */
    const cf* twA = reinterpret_cast<const cf*>(twAB);
    const cf* twB = twA + kTwAItems;
#elif defined(GR4PM_TWA_L1)
    const cf* twA = reinterpret_cast<const cf*>(twAB);
    const cf* twB = reinterpret_cast<const cf*>(lds4 + kLdsTwB);
#else
    const cf* twA = reinterpret_cast<const cf*>(lds4 + kLdsTwA);
    const cf* twB = reinterpret_cast<const cf*>(lds4 + kLdsTwB);
#endif
    const cf* x = in + static_cast<size_t>(blockIdx.y) * in_stride + static_cast<size_t>(b) * stride_s;
    float* zo = zpow + static_cast<size_t>(blockIdx.y) * z_stride + static_cast<size_t>(b) * stride_s;

    // stage the FFT-2 twiddle tables (twA ++ twB contiguous in global) and template 0
    for (int i = tid; i < (kTwAItems + kTwBItems) / 2; i += kCorrThreads) lds4[kLdsTwA + i] = twAB[i];
#pragma unroll
    for (int u = 0; u < kTmplPerThread; ++u) lds4[kLdsTmpl + u * kCorrThreads + tid] = tmpl[u * kCorrThreads + tid];

    cf r[32];
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4* xp = reinterpret_cast<const float4*>(x);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float4 v = xp[lane + 64 * n1];
            r[2 * n1] = mk(v.x, v.y);
            r[2 * n1 + 1] = mk(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            r[2 * n1] = x[2 * lane + 128 * n1];
            r[2 * n1 + 1] = x[2 * lane + 1 + 128 * n1];
        }
    }
    fft1_wave(lane, r, lds, tw1a, tw1b);
    cf X[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) X[j] = r[j];
    float zmax[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) zmax[j] = -1.0f; // hpp:303
    // Template hand-off without workgroup barriers.  Two monotonic words in LDS:
    //   consumed = number of (wave, bin) template reads finished      (ds_add by every wave)
    //   ready    = index of the template currently in the LDS buffer  (written by wave 0)
    // Wave 0 copies template bin+1 global -> LDS with the DMA path (global_load_lds: no VGPRs,
    // no ds_write) in the middle of its own transform, once consumed == 4 (bin + 1); everybody
    // waits for ready >= bin before reading.  LDS operations of a wave execute in order, so
    // the ds_add of a wave is performed after its template reads.  All spins are bounded, and a spin
    // that runs out does NOT carry on silently with a stale template: it raises the handle's fault word
    // (pinned host memory), which process() turns into GR4PM_ERR_HIP after its stream synchronisation.
    unsigned* ctl = reinterpret_cast<unsigned*>(lds4 + kLdsCtl);
    if (tid == 0) {
        ctl[0] = 0; // consumed
        ctl[1] = 0; // ready
    }
    __syncthreads();
    auto wait_ge = [&](int word, unsigned v) {
        for (int guard = 0; guard < spin_limit; ++guard) {
            if (__hip_atomic_load(ctl + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= v) return;
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    for (int bin = 0; bin < n_bins; ++bin) {
        const bool more = bin + 1 < n_bins;
        cf p[32], c[32];
        wait_ge(1, static_cast<unsigned>(bin));
        mul_template(lane, X, p, lds4 + kLdsTmpl); // hpp:247-249
        wave_lds_sync(); // keep the template reads ahead of the counter update in program order
        if (lane == 0) __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        auto hand_off = [&]() {
            if (!more || wave != 0) return; // wave-uniform
            wait_ge(0, static_cast<unsigned>(kWavesPerWg * (bin + 1)));
            const float4* tg = tmpl + static_cast<size_t>(bin + 1) * 1024;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // one wave moves the whole 16 KiB template: 16 x 1 KiB, LDS dst = base + lane * 16 B
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tg + u * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds4 + kLdsTmpl + u * 64), 16,
                                                 0, 0);
            }
        };
        fft2_wave(lane, p, c, lds, twA, twB, hand_off); // hpp:250-251
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            // hpp:307-308: the best bin's power; max() == the strict-> scan for the VALUE (the
            // bin index is recomputed by k_tags for detections only).  One v_max_f32 (fmaxf would
            // add a canonicalising second one).
            const float pw = fmaf(c[j].y, c[j].y, c[j].x * c[j].x);
            asm("v_max_f32 %0, %1, %2" : "=v"(zmax[j]) : "v"(zmax[j]), "v"(pw));
        }
        if (more && wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA has landed in LDS
            if (lane == 0) __hip_atomic_store(ctl + 1, static_cast<unsigned>(bin + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (!active) return;
    // lag k <-> correlation index (N - k) mod N (hpp:300); lanes hold consecutive indices
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int k = fft2_out_index(lane, j);
        const uint32_t lag = static_cast<uint32_t>((kFftN - k) & (kFftN - 1));
        if (lag < stride_s) zo[lag] = zmax[j];
    }
}

// ------------------------------------------------------------------ k_correlate_pair
// EXPERIMENTAL (GR4PM_CORRELATOR=pair at handle creation; default is k_correlate).  The same
// correlator with TWO waves per overlap-save block (fft2048_pair.hpp): 16 points per lane, so
// three workgroups of two blocks (12 waves) fit a CU instead of 8 waves.  The exchanges cross
// the two waves of a block: every half-round is store | sync | load | sync (workgroup barrier of
// the 4 waves, or pairwise flag words with -DGR4PM_PAIR_FLAGS).  Template hand-off as in
// k_correlate.  Bit-identical to k_correlate (tests: host emulation and
// test_two_waves_per_block_correlator_is_bit_identical).
// Round-1 status: 1.18 ms per 2^26 samples at 9 bins against 1.01 ms for k_correlate.  The VALU
// work per block is the same (1132 packed + ~460 other instructions), but hipcc wants 182 VGPRs
// for it (168 are allowed at three waves per SIMD: 12 dwords spill inside the bin loop), the
// eight synchronisations per transform cost what the third wave gains, and LDS instructions per
// block go from 194 to 290.  At two waves per SIMD without spills it runs at 1.17 ms.
// LDS map (float4 units): [0, 1024) template | [1024, 1920) twAp | [1920, 2040) twB |
// [2040, 2040 + 2 * 576) two exchange buffers of 9 KiB | control words = 51 KiB per workgroup
constexpr int kPairWaves = 4, kPairThreads = kPairWaves * 64, kPairBlocksPerWg = kPairWaves / 2;
constexpr int kPLdsTmpl = 0, kPLdsTwA = 1024, kPLdsTwB = kPLdsTwA + kTwApItems / 2, kPLdsExch = kPLdsTwB + kTwBItems / 2;
constexpr int kPLdsTotal = kPLdsExch + kPairBlocksPerWg * (kExchangeItems / 2);
static_assert(3 * (kPLdsTotal + 2) * 16 <= 160 * 1024, "LDS budget for three workgroups per CU");

[[maybe_unused]] __device__ __forceinline__ void pair_sync()
{
    // LDS operations of the other wave are not ordered with ours: wait for our own stores /
    // loads to have been performed, then meet at the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void mul_template_pair(int L, const cf* X, cf* r, const float4* tp)
{
#pragma unroll
    for (int jp = 0; jp < 8; ++jp) {
        if ((jp & 1) == 0) asm volatile("" ::: "memory"); // at most two template reads in flight (registers)
        const float4 t = tp[jp * 128 + L];
        r[2 * jp] = cmul(X[2 * jp], mk(t.x, t.y));
        r[2 * jp + 1] = cmul(X[2 * jp + 1], mk(t.z, t.w));
    }
}

// pairwise meeting point without a workgroup barrier: two monotonic words per pair in LDS, one
// per wave.  A wave publishes the step it has reached -- LDS operations of one wave execute in
// issue order, so the word is written after the stores (or loads) in front of it have been
// performed -- and waits until its partner has published the same step.
struct PairSync {
    unsigned* mine;
    const unsigned* other;
    unsigned* fault;
    int spin_limit;
    unsigned step = 0;
    __device__ __forceinline__ void operator()(int lane)
    {
        ++step;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(mine, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool met = false;
        for (int guard = 0; guard < 4 * spin_limit; ++guard) {
            const unsigned v = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(v) >= step) {
                met = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!met && lane == 0) __hip_atomic_fetch_or(fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("" ::: "memory");
    }
};

__global__ __launch_bounds__(kPairThreads, 3) void k_correlate_pair(
    const cf* __restrict__ in, size_t in_stride, uint32_t n_blocks, uint32_t stride_s, int n_bins,
    const float4* __restrict__ tmplp, const cf* __restrict__ tw1p, const cf* __restrict__ tw1b,
    const float4* __restrict__ twABp, float* __restrict__ zpow, size_t z_stride, unsigned* __restrict__ fault,
    int spin_limit)
{
    __shared__ float4 lds4[kPLdsTotal + 2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wv = wave & 1, pair = wave >> 1;
    const int L = lane + 64 * wv;
    const uint32_t b_raw = blockIdx.x * kPairBlocksPerWg + pair;
    const bool active = b_raw < n_blocks;
    const uint32_t b = active ? b_raw : n_blocks - 1; // idle pairs shadow the last block
    cf* lds = reinterpret_cast<cf*>(lds4 + kPLdsExch) + pair * kExchangeItems;
    const cf* twA = reinterpret_cast<const cf*>(lds4 + kPLdsTwA);
    const cf* twB = reinterpret_cast<const cf*>(lds4 + kPLdsTwB);
    const cf* x = in + static_cast<size_t>(blockIdx.y) * in_stride + static_cast<size_t>(b) * stride_s;
    float* zo = zpow + static_cast<size_t>(blockIdx.y) * z_stride + static_cast<size_t>(b) * stride_s;
    // control words: [0] consumed, [1] ready (template hand-off, as in k_correlate), [4 + 2 pair + wv] pair steps
    unsigned* ctl = reinterpret_cast<unsigned*>(lds4 + kPLdsTotal);

    for (int i = tid; i < (kTwApItems + kTwBItems) / 2; i += kPairThreads) lds4[kPLdsTwA + i] = twABp[i];
#pragma unroll
    for (int u = 0; u < 1024 / kPairThreads; ++u) lds4[kPLdsTmpl + u * kPairThreads + tid] = tmplp[u * kPairThreads + tid];
    if (tid < 8) ctl[tid] = 0;

    cf r[16], bb[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) r[n1] = x[L + 128 * n1];
    __syncthreads(); // twiddles, template 0 and control words staged
#if defined(GR4PM_PAIR_NOSYNC) /* timing-only ablation: wrong results */
    auto sync = [&](int) { wave_lds_sync(); };
#elif defined(GR4PM_PAIR_FLAGS) /* measured: 1.27 ms per 2^26 samples against 1.18 with the workgroup barrier */
    PairSync sync{ ctl + 4 + 2 * pair + wv, ctl + 4 + 2 * pair + (wv ^ 1), fault, spin_limit };
#else
    auto sync = [&](int) { pair_sync(); };
#endif
    auto wait_ge = [&](int word, unsigned v) {
        for (int guard = 0; guard < spin_limit; ++guard) {
            if (__hip_atomic_load(ctl + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= v) return;
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    // ---- FFT-1
    fft1p_pass1(L, r, tw1p);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        fft1p_store1(L, r, lds, h);
        sync(lane);
        if (wv == h) fft1p_load2(L, bb, lds);
        sync(lane);
    }
    fft1p_pass2(L, bb, tw1b);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (wv == h) fft1p_store2(L, bb, lds);
        sync(lane);
        fft1p_load3(L, r, lds, h);
        sync(lane);
    }
    fft1p_pass3(r);
    cf X[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) X[j] = r[j];
    float zmax[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) zmax[j] = -1.0f; // hpp:303

    for (int bin = 0; bin < n_bins; ++bin) {
        const bool more = bin + 1 < n_bins;
        cf p[16], c[16];
        wait_ge(1, static_cast<unsigned>(bin));
        mul_template_pair(L, X, p, lds4 + kPLdsTmpl); // hpp:247-249
        wave_lds_sync(); // keep the template reads ahead of the counter update in program order
        if (lane == 0) __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        fft2p_passA(L, p, twA);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fft2p_storeA(L, p, lds, h);
            sync(lane);
            if (wv == h) fft2p_loadB(L, bb, lds);
            sync(lane);
        }
        fft2p_passB(L, bb, twB);
        if (more && wave == 0) {
            // template hand-off as in k_correlate: once all four waves have read template `bin`, move
            // template bin + 1 global -> LDS with the DMA path, 16 x 1 KiB
            wait_ge(0, static_cast<unsigned>(kPairWaves * (bin + 1)));
            const float4* tg = tmplp + static_cast<size_t>(bin + 1) * 1024;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tg + u * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds4 + kPLdsTmpl + u * 64), 16,
                                                 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fft2p_storeB(L, bb, lds, h);
            sync(lane);
            if (wv == h) fft2p_loadC(L, c, lds);
            sync(lane);
        }
        fft2p_passC(c); // hpp:250-251
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float pw = fmaf(c[j].y, c[j].y, c[j].x * c[j].x); // as in k_correlate (hpp:307-308)
            asm("v_max_f32 %0, %1, %2" : "=v"(zmax[j]) : "v"(zmax[j]), "v"(pw));
        }
        if (more && wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA has landed in LDS
            if (lane == 0) __hip_atomic_store(ctl + 1, static_cast<unsigned>(bin + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (!active) return;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = fft2p_out_index(L, j);
        const uint32_t lag = static_cast<uint32_t>((kFftN - k) & (kFftN - 1));
        if (lag < stride_s) zo[lag] = zmax[j];
    }
}

} // namespace
} // namespace gr4pm
#include "correlate_w64.hpp"
#include "correlate_w64_one.hpp"
#include "correlate_4096.hpp"
namespace gr4pm {
namespace {

// ------------------------------------------------------------------ k_candidates
// B(p) = zpow[p] >= max(zpow[p+1 .. p+T]) for local positions [0, cnt) of a channel; z points
// at local position 0 and is readable up to cnt + T - 1.  One workgroup = 1024 positions.
// The window of lane l of 64-block k is: the rest of block k, the full blocks k+1 .. k+tq-1
// (+ block k+tq when l + T%64 >= 64) and a prefix of the last block.  Almost every position
// already fails against the rest of its block and the full-block maxima (wave-uniform), so
// the prefix maxima of the last block are only computed for the rare survivors.
// inclusive prefix maximum over the 64 lanes with DPP row shifts / row broadcasts (no LDS
// round trips, unlike __shfl_*): lane l ends with max(v[0..l]); lane 63 holds the wave maximum
__device__ __forceinline__ float wave_prefix_max(float v)
{
    // v_max_f32 with a DPP source: a lane whose source lane does not exist (or whose row is masked out) keeps its
    // value, which is what an inclusive scan step wants -- ONE instruction a step (the builtin form above costs a
    // v_mov of -inf, a v_mov_dpp and two v_max: fmaxf canonicalises).  s_nop 1: the two wait states between a VALU
    // write of a VGPR and its use as a DPP source.
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return v;
}
// value of lane l-1 (lane 0 gets -inf): wave_shr:1
__device__ __forceinline__ float wave_prev(float v)
{
    const float ninf = -INFINITY;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ninf),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// items per workgroup: the T-item halo behind a tile is read (and prefix-maximised) again by the
// next workgroup, so a larger tile means less duplicated work (4096: 19 % instead of 75 % at T = 768)
constexpr uint32_t kCandBlocks = 64, kCandTile = kCandBlocks * 64;
__global__ __launch_bounds__(256) void k_candidates(const float* __restrict__ zbase, size_t z_stride,
                                                    uint32_t cnt, uint32_t T,
                                                    unsigned long long* __restrict__ bitmap,
                                                    size_t bm_stride)
{
    extern __shared__ float sm[];
    const uint32_t tile0 = blockIdx.x * kCandTile;
    const float* z = zbase + static_cast<size_t>(blockIdx.y) * z_stride;
    unsigned long long* bmp = bitmap + static_cast<size_t>(blockIdx.y) * bm_stride;
    const uint32_t span = kCandTile + T;       // values needed: local [tile0, tile0 + span)
    const uint32_t nblk = kCandBlocks + (T >> 6) + 2u; // 64-item blocks incl. the last (partial) ones
    float* s = sm;                             // values, nblk * 64
    float* bmax = sm + nblk * 64;              // block maxima, nblk
    const uint32_t avail = cnt + T;            // readable items
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // all global loads of this wave are issued before the first one is used
    constexpr int kMaxBlkPerWave = 37; // nblk <= 16 + 128 + 2 for T <= 8192
    for (uint32_t b0 = wave; b0 < nblk; b0 += 4 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t blk = b0 + 4 * u;
            const uint32_t i = blk * 64 + lane, g = tile0 + i;
            v[u] = (blk < nblk && i < span && g < avail) ? z[g] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t blk = b0 + 4 * u;
            if (blk < nblk) {
                s[blk * 64 + lane] = v[u];
                const float m = wave_prefix_max(v[u]);
                if (lane == 63) bmax[blk] = m;
            }
        }
    }
    (void)kMaxBlkPerWave;
    __syncthreads();
    const uint32_t tq = T >> 6, tr = T & 63;
    // maximum of the full blocks kb+1 .. kb+tq-1 (inside every item's window), once per block
    float* mfull = bmax + nblk;
    if (threadIdx.x < kCandBlocks && T >= 128) {
        float mu = -INFINITY;
        for (uint32_t k = threadIdx.x + 1; k < threadIdx.x + tq; ++k) mu = fmaxf(mu, bmax[k]);
        mfull[threadIdx.x] = mu;
    }
    __syncthreads();
    for (uint32_t kb = wave; kb < kCandBlocks; kb += 4) { // the blocks whose flags this workgroup owns
        unsigned long long word;
        if (T < 128) { // tiny windows: plain scan
            const uint32_t i = kb * 64 + lane;
            float m = -INFINITY;
            for (uint32_t u = 1; u <= T; ++u) m = fmaxf(m, s[i + u]);
            word = __ballot(tile0 + i < cnt && s[i] >= m);
        } else {
            // lanes run over the block BACKWARDS (lane l <-> item 63 - l) so that "the rest of
            // my block" is an exclusive prefix over lanes
            const int it = 63 - lane;
            const uint32_t i = kb * 64 + it;
            const float v = s[i];
            float m = wave_prev(wave_prefix_max(v));
            m = fmaxf(m, mfull[kb]);
            const bool wrap = it + tr >= 64; // the window also covers the whole of block kb+tq
            if (wrap) m = fmaxf(m, bmax[kb + tq]);
            bool flag = tile0 + i < cnt && v >= m;
            if (__ballot(flag)) { // survivors: prefix of the last block, up to item (it + tr) % 64
                const float pa = wave_prefix_max(s[(kb + tq) * 64 + lane]);
                const float pb = wave_prefix_max(s[(kb + tq + 1) * 64 + lane]);
                const int src = (it + tr) & 63;
                const float qa = __shfl(pa, src), qb = __shfl(pb, src);
                flag = flag && v >= (wrap ? qb : qa);
            }
            word = __brevll(__ballot(flag)); // back to item order
        }
        if (lane == 0) bmp[(tile0 >> 6) + kb] = word;
    }
}

// The same flags WITHOUT LDS, for T = 64 TQ (the default 768 = 12 x 64): k_candidates wants 20 KiB of LDS per
// workgroup and therefore only gets a CU between two correlator workgroups (151 KiB each) -- behind the
// look-ahead it ran 4x slower than alone and stretched the correlator with it.  Here one wave walks `chain`
// consecutive T-item blocks with everything in registers (van Herk / Gil-Werman with block size T):
//     max(z[p+1 .. p+T]) = max( suffix maximum of p's block from p+1,  prefix maximum of the next block up to p+T )
// and p+T has the same offset in the next block as p in its own, i.e. the same lane and register.
// Item i of a block <-> register i / 64, lane 63 - i % 64 (lanes reversed, so that the suffix inside a row is a
// DPP prefix over lanes and "the item after mine" is wave_shr:1); rows are coalesced 256-byte loads.
// Per block and wave: TQ loads, 2 TQ DPP scans, 2 TQ ds_bpermute (crossbar only, no LDS allocation), TQ ballots.
// Round 4: (i) LAZY rows.  The window of an item of row r is: the rest of its own row, the rows behind it in its block
// and the rows in front of it in the next block -- whole rows, whose maxima are wave-uniform (SGPRs) -- and the part of
// row r of the next block up to the item's own offset.  U = the maximum over those whole rows costs scalar
// instructions only; a row whose own maximum is below U cannot hold a candidate (strictly below: ties count, hpp:314),
// and on noise that is 11 rows of 12.  Only the other rows pay for the item-order prefix of the next block's row (two
// ds_bpermute + a scan), the shifted suffix, the compare and the ballot: 39 -> ~12 vector instructions per row on
// average, identical flags (test_candidate_kernels_agree).  (ii) the suffix scans of four rows in ONE asm statement:
// four independent chains fill each other's DPP wait states (was: an s_nop 1 in front of every step).
// Uniform float maxima are kept as the usual monotone integer key of their bit pattern (scalar ALU has no float compare
// on gfx950; round 5: the key itself is kept, so a maximum is one s_max_i32): correct for every float, -inf (the fill
// behind the stream) and -1 included.
__device__ __forceinline__ int fkey(int b) { return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ void wave_prefix_max4(float& a, float& b, float& c, float& d)
{
#define GR4PM_STEP(ctl)                                    \
    "v_max_f32_dpp %0, %0, %0 " ctl "\n\t"                 \
    "v_max_f32_dpp %1, %1, %1 " ctl "\n\t"                 \
    "v_max_f32_dpp %2, %2, %2 " ctl "\n\t"                 \
    "v_max_f32_dpp %3, %3, %3 " ctl "\n\t"
    asm volatile("s_nop 1\n\t" GR4PM_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") GR4PM_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                     GR4PM_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") GR4PM_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                         GR4PM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") GR4PM_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef GR4PM_STEP
}
// Round 5, MEDIAN: the wave also holds the T-item block BEFORE the current one, i.e. the whole history [p - T, p + T] of
// every candidate p of the current block (the 2T + 1 powers the reference's test looks at, hpp:273-279, when the scan
// reaches p) -- so the median test of a candidate costs ballots and population counts on registers, and its result goes
// into a second bitmap (`passmap`: candidate AND 2 count(history < z[p] / power_threshold) >= 2T + 1).  The scan
// (k_tile_visit) then only has to look its visited candidates up: the separate pass over the powers (k_median_tests:
// 6 KB read per visited candidate, 1.2 GB and 205 us per 2^28 samples) and the visit list are gone.  Every candidate
// is tested, visited or not: on noise there is about one per block, and about two in three are visited.
// hpp:273-279 for the candidate at row R (compile time), lane lp (uniform) of the current block, on registers: its history
// is the items at or behind its own offset in the block before, the whole current block, and the items up to its own
// offset in the next block = 2T + 1 powers.  Counted per lane (a compare and an add-with-carry per row: two vector
// instructions, nothing on the scalar unit -- a ballot, a population count and a scalar add per row were three issue
// slots), summed over the wave once.  Uniform result.
template <int TQ, int R>
__device__ __forceinline__ bool median_in_registers(const float* prv, const float* cur, const float* nxt, int lp, int lane,
                                                    float power_threshold)
{
    constexpr uint32_t T = TQ * 64;
    const float best = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[R]), lp));
    const float thr = best / power_threshold; // hpp:275
    uint32_t acc = 0;
    const bool at_or_before = lane <= lp, at_or_after = lane >= lp;
    // the 2 TQ - 1 rows that lie wholly inside the history, four at a time (a compare's mask is read by its add three
    // instructions later: no wait states to pad), then the two partial rows
    float full[2 * TQ - 1];
    {
        int n = 0;
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
            if (q > R) full[n++] = prv[q];
            full[n++] = cur[q];
            if (q < R) full[n++] = nxt[q];
        }
    }
#pragma unroll
    for (int q = 0; q + 4 <= 2 * TQ - 1; q += 4) {
        unsigned long long m0, m1, m2, m3;
        asm("v_cmp_lt_f32_e64 %1, %5, %9\n\t"
            "v_cmp_lt_f32_e64 %2, %6, %9\n\t"
            "v_cmp_lt_f32_e64 %3, %7, %9\n\t"
            "v_cmp_lt_f32_e64 %4, %8, %9\n\t"
            "v_addc_co_u32_e64 %0, %1, %0, 0, %1\n\t"
            "v_addc_co_u32_e64 %0, %2, %0, 0, %2\n\t"
            "v_addc_co_u32_e64 %0, %3, %0, 0, %3\n\t"
            "v_addc_co_u32_e64 %0, %4, %0, 0, %4"
            : "+v"(acc), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
            : "v"(full[q]), "v"(full[q + 1]), "v"(full[q + 2]), "v"(full[q + 3]), "v"(thr));
    }
#pragma unroll
    for (int q = (2 * TQ - 1) / 4 * 4; q < 2 * TQ - 1; ++q) acc += full[q] < thr ? 1u : 0u;
    acc += (at_or_before && prv[R] < thr) ? 1u : 0u;
    acc += (at_or_after && nxt[R] < thr) ? 1u : 0u;
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(acc));
    const uint32_t below = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(acc), 63));
    return 2u * below >= 2u * T + 1u; // hpp:279
}
template <int TQ, int R>
__device__ __forceinline__ bool median_dispatch(int r, const float* prv, const float* cur, const float* nxt, int lp, int lane,
                                                float power_threshold)
{
    if constexpr (R >= TQ) {
        return false;
    } else {
        if (r == R) return median_in_registers<TQ, R>(prv, cur, nxt, lp, lane, power_threshold); // (uniform: a scalar branch)
        return median_dispatch<TQ, R + 1>(r, prv, cur, nxt, lp, lane, power_threshold);
    }
}

template <int TQ, bool MEDIAN>
__global__ __launch_bounds__(64) void k_candidates_wave(const float* __restrict__ zbase, size_t z_stride, uint32_t cnt,
                                                        uint32_t n_words, uint32_t chain,
                                                        unsigned long long* __restrict__ bitmap, size_t bm_stride,
                                                        unsigned long long* __restrict__ passmap,
                                                        unsigned long long* __restrict__ defermap, float power_threshold)
{
    constexpr uint32_t T = TQ * 64;
    const int lane = threadIdx.x, rl = 63 - lane;
    const float* z = zbase + static_cast<size_t>(blockIdx.y) * z_stride;
    unsigned long long* bmp = bitmap + static_cast<size_t>(blockIdx.y) * bm_stride;
    unsigned long long* pmp = MEDIAN ? passmap + 2 * static_cast<size_t>(blockIdx.y) * bm_stride : nullptr;
    (void)defermap;
    // Which candidates are tested here.  (1) Only candidates the scan CAN visit.  The scan visits p when a reset falls
    // into (prev(p), p], prev(p) = the candidate before p; resets are v + T + 1 for visited candidates v: so some
    // candidate must lie in [prev(p) - T, p - T - 1] -- for p and prev(p) in one block that is "a candidate of the block
    // before at an offset in [offset(prev(p)), offset(p) - 1]", two bitmap look-ups.  A packet stream has about five
    // candidates a block (the correlation's side lobes around every local maximum) and the scan visits one.  The first
    // candidate of a block, and every candidate where the look-up needs flags this wave does not hold (the first block of
    // its chain), counts as visitable.  (2) At most kTestsPerBlock per block (constant input -- the all-zero stream of the
    // reference's own benchmarks -- makes every item a candidate: 768 a block, of which the scan visits one).
    // Every candidate that is NOT tested is DEFERRED (second word of the pair): should the scan visit it after all,
    // k_resolve_visited hands it to k_median_tests<true>, which reads its history from memory as in round 4.  The rules
    // above therefore decide what a candidate costs, never what the detector finds.
    constexpr uint32_t kTestsPerBlock = 8;
    const uint32_t avail = cnt + T; // readable items
    const uint32_t n_blk = (n_words + TQ - 1) / TQ;
    const uint32_t b0 = blockIdx.x * chain;
    if (b0 >= n_blk) return;
    const uint32_t b1 = min(b0 + chain, n_blk);
    auto load = [&](uint32_t b, float* v) {
        if ((b + 1) * T <= avail) {
            // (uniform) the whole block is readable -- every block but the stream's last ones: a uniform base, the lane's
            // offset and the row as the instruction's immediate offset, instead of an index, a compare, an exec mask, a
            // branch and a 64-bit address per row (60 of the kernel's 374 vector instructions per block)
            const char* zb = reinterpret_cast<const char*>(z + static_cast<size_t>(b) * T);
            const uint32_t voff = 4u * static_cast<uint32_t>(rl);
#pragma unroll
            for (int r = 0; r < TQ; ++r)
                v[r] = *reinterpret_cast<const float*>(zb + static_cast<size_t>(voff) + 256 * r);
            return;
        }
#pragma unroll
        for (int r = 0; r < TQ; ++r) {
            const uint32_t g = b * T + 64u * r + rl;
            v[r] = g < avail ? z[g] : -INFINITY;
        }
    };
    // per block: rowmax[r] = the monotone integer key (fkey) of the maximum of row r (uniform: readlane, i.e. an SGPR); the
    // uniform maxima below are integer maxima of keys.  The scans themselves (suffix maxima inside the rows) are NOT kept
    // (round 5): only a row that can hold a candidate needs its own, one or two rows a block, and recomputing those costs
    // six instructions where keeping all of them cost 24 registers (two blocks' worth) and twelve copies per block.
    auto scans = [&](const float* v, int* rowmax) {
        float t[TQ];
#pragma unroll
        for (int r = 0; r < TQ; ++r) t[r] = v[r];
        if constexpr (TQ % 4 == 0) {
#pragma unroll
            for (int r = 0; r < TQ; r += 4) wave_prefix_max4(t[r], t[r + 1], t[r + 2], t[r + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < TQ; ++r) t[r] = wave_prefix_max(t[r]);
        }
#pragma unroll
        for (int r = 0; r < TQ; ++r) rowmax[r] = fkey(__builtin_amdgcn_readlane(__builtin_bit_cast(int, t[r]), 63));
    };
    const int ninf = fkey(__builtin_bit_cast(int, -INFINITY));
    float cur[TQ], nxt[TQ], prv[MEDIAN ? TQ : 1];
    int rowmax[TQ];
    load(b0, cur);
    if (MEDIAN) {
        // the block before the chain's first one: positions (b0 - 1) T ..; for b0 == 0 that is the T items before local
        // position 0, which the caller keeps readable (the powers' carry: what k_median_tests read there)
#pragma unroll
        for (int r = 0; r < TQ; ++r) {
            const long long g = static_cast<long long>(b0) * T - static_cast<long long>(T) + 64 * r + rl;
            prv[r] = g < static_cast<long long>(avail) ? z[g] : -INFINITY; // (behind the stream: in nobody's history)
        }
    }
    scans(cur, rowmax);
    unsigned long long before_mine = 0; // the candidate words of the block before (lane r: row r), known from b0 + 1 on
    for (uint32_t b = b0; b < b1; ++b) {
        load(b + 1, nxt);
        int nrowmax[TQ], before[TQ]; // before[r]: maximum of rows 0 .. r-1 of the next block
        scans(nxt, nrowmax);
        before[0] = ninf;
#pragma unroll
        for (int r = 1; r < TQ; ++r) before[r] = max(before[r - 1], nrowmax[r - 1]);
        unsigned long long mine = 0;
        int after = ninf; // maximum of the rows behind r in this block: built while r walks down
#pragma unroll
        for (int r = TQ - 1; r >= 0; --r) {
            const int U = max(after, before[r]);
            unsigned long long word = 0;
            if (rowmax[r] >= U) { // (uniform: a scalar branch)
                const float fwd = __shfl(nxt[r], rl);              // the next block's row r in item order
                const float pr = __shfl(wave_prefix_max(fwd), rl); // its prefix maximum up to the lane's own offset
                const float sfx = wave_prefix_max(cur[r]);         // suffix maximum inside the row (reversed lanes)
                const float m = fmaxf(fmaxf(wave_prev(sfx), __builtin_bit_cast(float, fkey(U))), pr); // (fkey: an involution)
                const uint32_t pos = b * T + 64u * r + rl;
                word = __brevll(__ballot(pos < cnt && cur[r] >= m)); // back to item order
            }
            if (lane == r) mine = word;
            after = max(after, rowmax[r]);
        }
        unsigned long long mine_pass = 0, mine_defer = 0;
        if (MEDIAN) {
            // the block's candidates in position order (a uniform walk over the rows that have any: usually one or two)
            unsigned long long rows = __ballot(mine != 0);
            uint32_t tests_left = kTestsPerBlock;
            int prev_pos = -1; // offset of the candidate before the current one inside this block
            const bool have_before = b != b0;
            while (rows) {
                const int r = __ffsll(static_cast<long long>(rows)) - 1;
                rows &= rows - 1;
                const unsigned long long word =
                    (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine >> 32), r))) << 32) |
                    static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine), r));
                unsigned long long todo = word, passed = 0, deferred = 0;
                if (tests_left == 0 || __popcll(word) > 8) {
                    // a row full of candidates (ties: constant input), or the block's budget is spent: the whole row is
                    // deferred without a look at its candidates one by one
                    deferred = word;
                    prev_pos = 64 * r + 63 - __clzll(static_cast<long long>(word));
                    todo = 0;
                }
                while (todo) {
                    const int bit = __ffsll(static_cast<long long>(todo)) - 1;
                    todo &= todo - 1;
                    const int o = 64 * r + bit;
                    bool visitable = true;
                    if (prev_pos >= 0) {
                        // a candidate of the block before at an offset in [prev_pos, o - 1]?
                        visitable = false;
                        if (have_before) {
                            const int lo = prev_pos, hi = o - 1, rlo = lo >> 6, rhi = hi >> 6;
                            unsigned long long mask = ~0ull;
                            if (lane == rlo) mask &= ~0ull << (lo & 63);
                            if (lane == rhi) mask &= ~0ull >> (63 - (hi & 63));
                            visitable = __ballot(lane >= rlo && lane <= rhi && (before_mine & mask) != 0) != 0;
                        }
                    }
                    prev_pos = o;
                    if (visitable && tests_left != 0) {
                        --tests_left;
                        if (median_dispatch<TQ, 0>(r, prv, cur, nxt, 63 - bit, lane, power_threshold)) passed |= 1ull << bit;
                    } else {
                        deferred |= 1ull << bit;
                    }
                }
                if (lane == r) {
                    mine_pass = passed;
                    mine_defer = deferred;
                }
            }
            before_mine = mine;
        }
        const uint32_t w = b * TQ + lane;
        if (lane < TQ && w < n_words) {
            bmp[w] = mine;
            if (MEDIAN) // (pass word, defer word) side by side: k_resolve_visited reads both with one 16-byte load
                reinterpret_cast<ulonglong2*>(pmp)[w] = make_ulonglong2(mine_pass, mine_defer);
        }
#pragma unroll
        for (int r = 0; r < TQ; ++r) {
            if (MEDIAN) prv[r] = cur[r];
            cur[r] = nxt[r];
            rowmax[r] = nrowmax[r];
        }
    }
}


// first set bit at local position >= r and < hi, or hi if none
__device__ __forceinline__ uint32_t next_candidate(const unsigned long long* bm, uint32_t r, uint32_t hi)
{
    if (r >= hi) return hi;
    uint32_t w = r >> 6;
    unsigned long long word = bm[w] & (~0ull << (r & 63));
    const uint32_t wlast = (hi - 1) >> 6;
    while (true) {
        if (word) {
            const uint32_t p = (w << 6) + static_cast<uint32_t>(__ffsll(static_cast<long long>(word)) - 1);
            return p < hi ? p : hi;
        }
        if (w == wlast) return hi;
        word = bm[++w];
    }
}

// table[tile][e], e in [0, T]: entering tile at local lo + e, the scan leaves it wanting to
// resume at hi + table (>= hi).  All entries that meet the same first candidate share one walk,
// so one wave handles a tile: every candidate inside the entry window [lo, lo+T] gets a lane
// that walks from it, plus one walk for the entries behind the window's last candidate.
// The whole bitmap of one tile (kTileW / 64 words) in the registers of one wave, lane l holding words
// l, l + 64, ...: the walks of k_tile_tables / k_tile_visit then cost a ballot and a shuffle per step instead of
// a global-memory round trip (they were the whole cost of those kernels: 150 + 175 us per 2^26 items).
constexpr int kTileChunks = static_cast<int>(kTileW / 4096);
static_assert(kTileW % 4096 == 0 && kTileChunks >= 1 && kTileChunks <= 16, "tile = whole 4096-item chunks");
struct TileBits {
    unsigned long long w[kTileChunks];
    uint32_t lo;
    __device__ __forceinline__ void load(const unsigned long long* bm, uint32_t lo_, uint32_t hi, int lane)
    {
        lo = __builtin_amdgcn_readfirstlane(lo_);
        const uint32_t w0 = lo >> 6, wlast = (hi - 1) >> 6;
#pragma unroll
        for (int k = 0; k < kTileChunks; ++k) {
            const uint32_t wi = w0 + 64u * k + lane;
            unsigned long long v = wi <= wlast ? bm[wi] : 0ull;
            if (wi == wlast && (hi & 63u)) v &= (1ull << (hi & 63u)) - 1ull; // nothing at or beyond hi
            w[k] = v;
        }
    }
    // word number j (0-based inside the tile), wave-uniform j
    __device__ __forceinline__ unsigned long long word(uint32_t j) const
    {
        unsigned long long v = 0;
#pragma unroll
        for (int k = 0; k < kTileChunks; ++k)
            if (static_cast<uint32_t>(k) == (j >> 6)) v = w[k];
        const unsigned long long x = __shfl(v, static_cast<int>(j & 63));
        // wave-uniform by construction: tell the compiler (scalar registers, scalar branches)
        const uint32_t lo32 = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x));
        const uint32_t hi32 = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x >> 32));
        return (static_cast<unsigned long long>(hi32) << 32) | lo32;
    }
    // first set bit at local position >= r and < hi, or hi if none (r wave-uniform, lo <= r)
    __device__ __forceinline__ uint32_t next(uint32_t r_, uint32_t hi_, int lane) const
    {
        // r and hi are wave-uniform; without the hint every chunk below runs under an exec mask, every step
        // executes all of them, and a step costs ~2000 cycles instead of ~100
        const uint32_t r = __builtin_amdgcn_readfirstlane(r_), hi = __builtin_amdgcn_readfirstlane(hi_);
        if (r >= hi) return hi;
        const uint32_t j0 = (r - lo) >> 6; // word of r inside the tile
#pragma unroll
        for (int k = 0; k < kTileChunks; ++k) {
            if (static_cast<uint32_t>(k) < (j0 >> 6)) continue;
            unsigned long long wd = w[k];
            const uint32_t j = 64u * k + lane;
            if (j < j0) wd = 0ull;
            else if (j == j0) wd &= ~0ull << (r & 63);
            const unsigned long long any = __ballot(wd != 0ull);
            if (any) {
                const int src = __ffsll(static_cast<long long>(any)) - 1;
                const unsigned long long hit = __shfl(wd, src);
                const uint32_t p = __builtin_amdgcn_readfirstlane(
                    lo + ((64u * k + src) << 6) + static_cast<uint32_t>(__ffsll(static_cast<long long>(hit)) - 1));
                return p < hi ? p : hi;
            }
        }
        return hi;
    }
    __device__ __forceinline__ uint32_t walk(uint32_t c, uint32_t hi, uint32_t T, int lane) const
    {
        uint32_t r = c + T + 1;
        while (r < hi) {
            const uint32_t p = next(r, hi, lane);
            if (p >= hi) return 0;
            r = p + T + 1;
        }
        return r - hi;
    }
};
__device__ __forceinline__ uint32_t walk_from_candidate(const unsigned long long* bm, uint32_t c, uint32_t hi,
                                                        uint32_t T)
{
    uint32_t r = c + T + 1;
    while (r < hi) {
        const uint32_t p = next_candidate(bm, r, hi);
        if (p >= hi) return 0; // scan position ends at hi
        r = p + T + 1;
    }
    return r - hi;
}
// the walking form (one walk per candidate of the entry window): kept for tiles with more candidates than the
// list form below holds (all-zero input: every item is a candidate)
__device__ __forceinline__ void tile_tables_walks(const TileBits& tb, const unsigned long long* bm, uint32_t* tab,
                                                  uint32_t lo, uint32_t hi, uint32_t wend, uint32_t T, int lane)
{
    uint32_t prev = lo; // first entry position not yet filled
    for (uint32_t w0 = lo; w0 < wend; w0 += 64) { // kTileW is a multiple of 64: words are aligned
        unsigned long long word = tb.word((w0 - lo) >> 6);
        if (wend - w0 < 64) word &= (1ull << (wend - w0)) - 1ull;
        if (!word) continue;
        const bool mine = (word >> lane) & 1ull;
        uint32_t res = 0;
        if (__popcll(word) <= 4) {
            // few candidates (the usual case on noise): walk them one after the other with the
            // whole wave scanning the bitmap, 4096 positions per load
            unsigned long long rest = word;
            while (rest) {
                const int bit = __ffsll(static_cast<long long>(rest)) - 1;
                rest &= rest - 1;
                const uint32_t rr = tb.walk(w0 + bit, hi, T, lane);
                if (lane == bit) res = rr;
            }
        } else if (mine) { // dense candidates (e.g. all-zero input): every step hits at once
            res = walk_from_candidate(bm, w0 + lane, hi, T);
        }
        // entries (previous candidate, candidate] share the candidate's result.  Filled by the whole wave with
        // coalesced stores (one lane per entry), not by every candidate's lane walking its own range: the
        // per-lane loops were ~T scattered 4-byte stores per tile and the whole cost of this kernel.
        const int topbit = 63 - __clzll(word), firstbit = __ffsll(static_cast<long long>(word)) - 1;
        // (a) the gap since the last candidate of an earlier word, up to this word's first candidate's word start
        const uint32_t res_first = __shfl(res, firstbit);
        for (uint32_t p = prev + lane; p < w0; p += 64) tab[p - lo] = res_first;
        // (b) inside this word: lane l's entry belongs to the first candidate at or above l
        if (lane <= topbit) {
            const unsigned long long at_or_above = word >> lane; // non-zero: topbit >= lane
            const int nb = lane + __ffsll(static_cast<long long>(at_or_above)) - 1;
            const uint32_t v = __shfl(res, nb);
            if (w0 + lane >= prev) tab[w0 + lane - lo] = v;
        } else {
            (void)__shfl(res, lane); // keep the shuffle convergent
        }
        prev = w0 + (63 - __clzll(word)) + 1;
    }
    if (prev < wend) { // entries behind the last candidate of the window share the next candidate
        uint32_t res = 0;
        const uint32_t c = tb.next(prev, hi, lane);
        if (c < hi) res = tb.walk(c, hi, T, lane);
        for (uint32_t p = prev + lane; p < wend; p += 64) tab[p - lo] = res;
    }
}


// table[tile][e], e in [0, T]: entering the tile at local lo + e, the scan leaves it wanting to resume at
// hi + table (>= hi).  LIST FORM: the tile's candidates (a few dozen on noise) are compacted into LDS in position
// order; every candidate learns the candidate the scan meets next (binary search for pos + T + 1), the chain is
// closed by pointer jumping (log2 rounds, all candidates at once), and an entry's value is the value of the first
// candidate at or behind it.  No serial walk: the walking form cost ~1 us per step and 150 us per 2^26 items.
constexpr uint32_t kListCap = 512, kListEnd = 0xffffffffu;
// inclusive prefix sum over the 64 lanes on the DPP path (row shifts by 1, 2, 4, 8 inside the rows of 16, then lane 15 of
// a row onto the next row and lane 31 onto the upper half): six vector adds, no LDS crossbar -- six ds_bpermute round
// trips per chunk of the bitmap, eight chunks a tile, were a tenth of k_tile_tables
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
    int v = static_cast<int>(x);
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
    return static_cast<uint32_t>(v);
}
__global__ __launch_bounds__(64) void k_tile_tables(const unsigned long long* __restrict__ bitmap,
                                                    size_t bm_stride, uint32_t cnt, uint32_t T, uint32_t n_tiles,
                                                    uint32_t* __restrict__ table, size_t table_stride)
{
#ifndef GR4PM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3); // latency-bound, few waves: win the issue arbitration against throughput kernels
#endif
    __shared__ uint32_t cpos[kListCap], jmp[kListCap], val[kListCap];
    const uint32_t tile = blockIdx.x;
    const int lane = threadIdx.x;
    const unsigned long long* bm = bitmap + static_cast<size_t>(blockIdx.y) * bm_stride;
    uint32_t* tab = table + static_cast<size_t>(blockIdx.y) * table_stride + static_cast<size_t>(tile) * (T + 1);
    const uint32_t lo = tile * kTileW;
    const uint32_t hi = min(lo + kTileW, cnt);
    const uint32_t wend = min(lo + T + 1, hi); // entry window [lo, wend)
    // entries that start at or beyond hi keep their position: table = (lo + e) - hi
    for (uint32_t e = (hi - lo) + lane; e <= T; e += 64) tab[e] = lo + e - hi;
    TileBits tb;
    tb.load(bm, lo, hi, lane);
    // ---- candidates in position order: chunk k, then lane, then bit
    uint32_t n = 0; // wave-uniform running count
    bool dense = false;
#pragma unroll
    for (int k = 0; k < kTileChunks; ++k) {
        unsigned long long wd = tb.w[k];
        const uint32_t mine = static_cast<uint32_t>(__popcll(wd));
        const uint32_t incl = wave_inclusive_sum(mine); // over the lanes
        const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        if (n + total > kListCap) {
            dense = true;
            break;
        }
        uint32_t at = n + incl - mine;
        const uint32_t base = lo + ((64u * k + lane) << 6);
        while (wd) {
            const uint32_t pos = base + static_cast<uint32_t>(__ffsll(static_cast<long long>(wd)) - 1);
            wd &= wd - 1;
            cpos[at++] = pos; // positions at or beyond hi cannot occur: the bitmap ends at cnt
        }
        n += total;
    }
    if (dense) {
        tile_tables_walks(tb, bm, tab, lo, hi, wend, T, lane);
        return;
    }
    wave_lds_sync();
    auto lower_bound = [&](uint32_t q) { // first list index with cpos >= q, n if none
        uint32_t a = 0, b = n;
        while (a < b) {
            const uint32_t m = (a + b) >> 1;
            if (cpos[m] < q) a = m + 1;
            else b = m;
        }
        return a;
    };
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t target = cpos[i] + T + 1;
        uint32_t jv = kListEnd, v = 0;
        if (target >= hi) {
            v = target - hi;
        } else {
            const uint32_t nx = lower_bound(target);
            if (nx < n) jv = nx; // else: the scan position ends at hi (value 0)
        }
        jmp[i] = jv;
        val[i] = v;
    }
    wave_lds_sync();
    for (int round = 0; round < 32; ++round) { // chain length <= n <= 512: at most 10 doubling rounds
        bool changed = false;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t jv = jmp[i];
            if (jv != kListEnd) {
                const uint32_t jj = jmp[jv];
                if (jj == kListEnd) {
                    val[i] = val[jv];
                    jmp[i] = kListEnd;
                } else {
                    jmp[i] = jj;
                }
                changed = true;
            }
        }
        wave_lds_sync();
        if (!__any(changed)) break;
    }
    // ---- entries: the value of the first candidate at or behind the entry position, 0 if there is none.  The search
    // runs over the candidates of the entry window and the one behind them only (cpos[n_win] >= wend > q): two or three
    // of a hundred on a packet stream, two steps instead of seven for each of the window's 769 entries.
    const uint32_t n_win = lower_bound(wend);
    for (uint32_t q = lo + lane; q < wend; q += 64) {
        uint32_t a = 0, b = n_win;
        while (a < b) {
            const uint32_t m = (a + b) >> 1;
            if (cpos[m] < q) a = m + 1;
            else b = m;
        }
        tab[q - lo] = a < n ? val[a] : 0u;
    }
    (void)n_tiles;
}

struct ChanState {
    unsigned long long r;        // absolute item where the greedy scan resumes
    unsigned int det_cnt;        // pending detections
    unsigned int rec_cnt;        // raw tag records written by the current call
    unsigned int overflow;
    unsigned int vis_cnt;        // candidates visited by the scan in the current call (k_tile_visit)
    unsigned int def_cnt;        // ... of which k_resolve_visited found untested (deferred): k_median_tests<true>
    unsigned int pad;
};

// The scan position after tile t is r_{t+1} = f_t(r_t) with f_t given by table row t; walking
// all tiles is a chain of dependent lookups.  It is done in three steps so that only
// n_tiles / kGroup lookups are truly serial:
//   k_group_tables   (parallel) for every group of kGroup tiles and every entry offset: where
//                    the scan leaves the group
//   k_scan_entries   (one workgroup per channel) thread 0 walks the groups, then one thread
//                    per group walks the tiles of its group from its known entry
#ifndef GR4PM_GROUP
#define GR4PM_GROUP 32
#endif
constexpr uint32_t kGroup = GR4PM_GROUP;
__device__ __forceinline__ unsigned long long walk_tiles(unsigned long long r, uint32_t t0, uint32_t t1,
                                                         uint32_t cnt, uint32_t T,
                                                         const uint32_t* __restrict__ tab,
                                                         int32_t* __restrict__ entry)
{
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t lo = t * kTileW;
        const uint32_t hi = min(lo + kTileW, cnt);
        if (r >= hi) {
            if (entry) entry[t] = -1;
            continue;
        }
        const uint32_t e = r > lo ? static_cast<uint32_t>(r) - lo : 0; // <= T by construction
        if (entry) entry[t] = static_cast<int32_t>(e);
        r = static_cast<unsigned long long>(hi) + tab[static_cast<size_t>(t) * (T + 1) + e];
    }
    return r;
}
__global__ void k_group_tables(uint32_t cnt, uint32_t T, uint32_t n_tiles, const uint32_t* __restrict__ table,
                               size_t table_stride, unsigned long long* __restrict__ gtable, size_t gtable_stride)
{
#ifndef GR4PM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3); // latency-bound, few waves: win the issue arbitration against throughput kernels
#endif
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = blockIdx.y, ch = blockIdx.z;
    if (e > T) return;
    const uint32_t t0 = grp * kGroup, t1 = min(t0 + kGroup, n_tiles);
    const unsigned long long r0 = static_cast<unsigned long long>(t0) * kTileW + e;
    gtable[ch * gtable_stride + static_cast<size_t>(grp) * (T + 1) + e] =
        walk_tiles(r0, t0, t1, cnt, T, table + ch * table_stride, nullptr);
}
// where the scan that enters group g0 at local position r leaves group g1 - 1; ge (if given): the entry of each group
__device__ __forceinline__ unsigned long long walk_groups(unsigned long long r, uint32_t g0, uint32_t g1, uint32_t cnt,
                                                          uint32_t T, uint32_t n_tiles,
                                                          const unsigned long long* __restrict__ gtab,
                                                          unsigned long long* __restrict__ ge)
{
    for (uint32_t g = g0; g < g1; ++g) {
        if (ge) ge[g] = r;
        const unsigned long long lo = static_cast<unsigned long long>(g) * kGroup * kTileW;
        const uint32_t t1 = min((g + 1) * kGroup, n_tiles);
        const unsigned long long hi = min(static_cast<unsigned long long>(t1) * kTileW, static_cast<unsigned long long>(cnt));
        if (r >= hi) continue;
        const unsigned long long e = r > lo ? r - lo : 0; // <= T by construction
        r = gtab[static_cast<size_t>(g) * (T + 1) + e];
    }
    return r;
}
// Round 6, a third level: for every SUPERGROUP of kSuper groups and every entry offset, where the scan leaves the
// supergroup.  The walk over the groups was 256 dependent loads by one thread per 2^28 samples (~ 0.4 us each: 100 us,
// the longest kernel of the detector's tail); with this table it is 16 + 16, the second 16 by 16 threads side by side.
// Launched with the front (look-ahead stream) behind k_group_tables, only for calls of more than kSuper groups.
constexpr uint32_t kSuper = 16;
__global__ void k_super_tables(uint32_t cnt, uint32_t T, uint32_t n_tiles, uint32_t n_groups,
                               const unsigned long long* __restrict__ gtable, size_t gtable_stride,
                               unsigned long long* __restrict__ sgtable, size_t sgtable_stride)
{
#ifndef GR4PM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3); // latency-bound, few waves: win the issue arbitration against throughput kernels
#endif
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sg = blockIdx.y, ch = blockIdx.z;
    if (e > T) return;
    const uint32_t g0 = sg * kSuper, g1 = min(g0 + kSuper, n_groups);
    const unsigned long long r0 = static_cast<unsigned long long>(g0) * kGroup * kTileW + e;
    sgtable[ch * sgtable_stride + static_cast<size_t>(sg) * (T + 1) + e] =
        walk_groups(r0, g0, g1, cnt, T, n_tiles, gtable + ch * gtable_stride, nullptr);
}
// one workgroup per channel: thread 0 walks the supergroups (the only truly serial part of a call; sgtable == nullptr:
// the groups), then one thread per supergroup walks its groups, then one thread per group walks its tiles, each from the
// entry just found
__global__ __launch_bounds__(256) void k_scan_entries(ChanState* __restrict__ st, unsigned long long A0,
                                                      uint32_t cnt, uint32_t T, uint32_t n_tiles,
                                                      const uint32_t* __restrict__ table, size_t table_stride,
                                                      const unsigned long long* __restrict__ gtable,
                                                      size_t gtable_stride, unsigned long long* __restrict__ gentry,
                                                      uint32_t n_groups, int32_t* __restrict__ entry,
                                                      const unsigned long long* __restrict__ sgtable,
                                                      size_t sgtable_stride)
{
#ifndef GR4PM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3); // latency-bound, few waves: win the issue arbitration against throughput kernels
#endif
    __shared__ unsigned long long sge[256]; // entry of each supergroup (more than 256 of them: the flat walk)
    const uint32_t ch = blockIdx.x;
    unsigned long long* ge = gentry + static_cast<size_t>(ch) * n_groups;
    const unsigned long long* gtab = gtable + ch * gtable_stride;
    const uint32_t n_super = (n_groups + kSuper - 1) / kSuper;
    const bool three_levels = sgtable != nullptr && n_super <= 256;
    if (threadIdx.x == 0) {
        const unsigned long long rabs = st[ch].r;
        unsigned long long r = rabs > A0 ? rabs - A0 : 0; // local; may exceed cnt
        if (three_levels) {
            for (uint32_t sg = 0; sg < n_super; ++sg) {
                sge[sg] = r;
                const unsigned long long lo = static_cast<unsigned long long>(sg) * kSuper * kGroup * kTileW;
                const uint32_t g1 = min((sg + 1) * kSuper, n_groups);
                const uint32_t t1 = min(g1 * kGroup, n_tiles);
                const unsigned long long hi = min(static_cast<unsigned long long>(t1) * kTileW,
                                                  static_cast<unsigned long long>(cnt));
                if (r >= hi) continue;
                const unsigned long long e = r > lo ? r - lo : 0; // <= T by construction
                r = sgtable[ch * sgtable_stride + static_cast<size_t>(sg) * (T + 1) + e];
            }
        } else {
            r = walk_groups(r, 0, n_groups, cnt, T, n_tiles, gtab, ge);
        }
        const unsigned long long rnew = A0 + (r > cnt ? r : cnt);
        st[ch].r = rnew > rabs ? rnew : rabs;
    }
    __syncthreads();
    if (three_levels) {
        for (uint32_t sg = threadIdx.x; sg < n_super; sg += blockDim.x)
            walk_groups(sge[sg], sg * kSuper, min((sg + 1) * kSuper, n_groups), cnt, T, n_tiles, gtab, ge);
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t grp = threadIdx.x; grp < n_groups; grp += blockDim.x) {
        const uint32_t t0 = grp * kGroup, t1 = min(t0 + kGroup, n_tiles);
        walk_tiles(ge[grp], t0, t1, cnt, T, table + ch * table_stride, entry + static_cast<size_t>(ch) * n_tiles);
    }
}

// Two steps instead of one wave per tile doing scan + median tests serially (175 us per 2^26 items: every test
// waited for the one before it):
//   k_tile_visit    one wave per (tile, channel): redo the scan from the known entry on the bitmap alone and
//                   append every visited candidate (the items whose history the reference tests, hpp:268-272)
//                   to the channel's visit list
//   k_median_tests  the tests of hpp:273-295 for all visited candidates of a call in parallel, one wave each:
//                   2T + 1 powers read with all loads in flight, bandwidth- instead of latency-bound
// zloc points at local position 0 of the channel (and is readable T items before it).
// Round 6: eight tiles per workgroup (one wave each), and the slots of all eight come from ONE atomic on the channel's
// counter: a wave per workgroup made 8 192 atomics on one address per 2^28 samples, one behind the other at the L2 (~ 12 ns
// each) -- 75 of the kernel's 105 us.
constexpr int kVisitWaves = 8;
__global__ __launch_bounds__(64 * kVisitWaves) void k_tile_visit(const unsigned long long* __restrict__ bitmap, size_t bm_stride,
                                                                uint32_t cnt, uint32_t T, uint32_t n_tiles,
                                                                const int32_t* __restrict__ entry, ChanState* __restrict__ st,
                                                                uint32_t* __restrict__ visit, uint32_t visit_cap)
{
#ifndef GR4PM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3); // latency-bound, few waves: win the issue arbitration against throughput kernels
#endif
    __shared__ uint32_t wave_cnt[kVisitWaves];
    __shared__ uint32_t wg_base;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const uint32_t tile = blockIdx.x * kVisitWaves + static_cast<uint32_t>(wave), ch = blockIdx.y;
    const int32_t e = tile < n_tiles ? entry[ch * n_tiles + tile] : -1;
    // lane i keeps the i-th candidate of the current batch of 64; a batch that fills up inside the walk (more than 64
    // visits per tile: a short time_threshold) reserves its slots by itself, the last batch of every wave goes through
    // the workgroup's one atomic
    uint32_t mine = 0, have = 0;
    if (e >= 0) { // (wave-uniform)
        const unsigned long long* bm = bitmap + static_cast<size_t>(ch) * bm_stride;
        const uint32_t lo = tile * kTileW;
        const uint32_t hi = min(lo + kTileW, cnt);
        uint32_t r = lo + static_cast<uint32_t>(e);
        TileBits tb;
        tb.load(bm, lo, hi, lane);
        while (r < hi) {
            const uint32_t p = tb.next(r, hi, lane);
            if (p >= hi) break;
            if (static_cast<uint32_t>(lane) == have) mine = p;
            if (++have == 64) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&st[ch].vis_cnt, 64u);
                base = __shfl(base, 0);
                if (base + lane < visit_cap) visit[static_cast<size_t>(ch) * visit_cap + base + lane] = mine;
                else st[ch].overflow = 1;
                have = 0;
            }
            r = p + T + 1;
        }
    }
    if (lane == 0) wave_cnt[wave] = have;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kVisitWaves; ++w) total += wave_cnt[w];
        wg_base = total ? atomicAdd(&st[ch].vis_cnt, total) : 0u;
    }
    __syncthreads();
    if (have) {
        uint32_t base = wg_base;
        for (int w = 0; w < wave; ++w) base += wave_cnt[w];
        if (static_cast<uint32_t>(lane) < have) {
            if (base + lane < visit_cap) visit[static_cast<size_t>(ch) * visit_cap + base + lane] = mine;
            else st[ch].overflow = 1;
        }
    }
}
// hpp:273-279 for one candidate p (wave-uniform; local position), by the whole wave: is count(history < z[p] /
// power_threshold) at least half of the 2T + 1 powers around p?  Uniform result.
__device__ __forceinline__ bool median_test_passes(const float* z, uint32_t p, uint32_t T, float power_threshold, int lane)
{
    const uint32_t hist = 2 * T + 1;
    // the candidate is the same for the whole wave: a uniform window pointer (SGPR pair) + the lane's 32-bit offset,
    // i.e. one address register per lane instead of a 64-bit address per load.  With that the kernel needs at most
    // 32 VGPRs, which is what a SIMD has left beside two correlator waves (HISTORY.md section 9): its waves run
    // beside them on the memory bandwidth the correlator leaves idle instead of waiting for a free CU.
    const float* zw = z + (static_cast<long long>(p) - static_cast<long long>(T)); // window [p - T, p + T]
    const float best = zw[T];
    const float thr = best / power_threshold; // hpp:275
    // counted per wave, not per lane: ballot + population count (scalar unit; no per-lane counter, no shuffles)
    uint32_t below = 0;
    const char* zb = reinterpret_cast<const char*>(zw);
    const uint32_t off = static_cast<uint32_t>(lane) * 4u; // byte offset of the lane's item inside a row of 64
    const uint32_t end = hist * 4u;
    // the row loop is UNIFORM (every lane takes every pass; a lane beyond the window reads its last item again and
    // does not count), so that the wave-wide count is the same in every lane
#pragma unroll 1
    for (uint32_t row = 0; row < end; row += 8u * 256u) { // eight independent loads in flight per lane
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float*>(zb + min(row + off + 256u * k, end - 4u));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            below += static_cast<uint32_t>(__popcll(__ballot(row + off + 256u * k < end && v[k] < thr)));
    }
    return 2 * below >= hist; // hpp:279
}
// DEFERRED: the list k_resolve_visited compacted (count def_cnt) instead of k_tile_visit's (vis_cnt)
template <bool DEFERRED>
__global__ __launch_bounds__(64) void k_median_tests(const float* __restrict__ zloc, size_t z_stride,
                                                     unsigned long long A0, uint32_t T, float power_threshold,
                                                     ChanState* __restrict__ st, const uint32_t* __restrict__ visit,
                                                     uint32_t visit_cap, unsigned long long* __restrict__ det,
                                                     uint32_t det_cap)
{
    const uint32_t ch = blockIdx.y;
    const int lane = threadIdx.x;
    const float* z = zloc + static_cast<size_t>(ch) * z_stride;
    const uint32_t n = min(DEFERRED ? st[ch].def_cnt : st[ch].vis_cnt, visit_cap);
    for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
        const uint32_t p = __builtin_amdgcn_readfirstlane(visit[static_cast<size_t>(ch) * visit_cap + idx]);
        if (median_test_passes(z, p, T, power_threshold, lane) && lane == 0) {
            const unsigned int slot = atomicAdd(&st[ch].det_cnt, 1u);
            if (slot < det_cap) det[static_cast<size_t>(ch) * det_cap + slot] = A0 + p;
            else st[ch].overflow = 1;
        }
    }
}
// Round 5, behind k_candidates_wave<12, true>: the visited candidates, 256 per wave.  A lane looks four candidates up
// in the (pass, defer) pairs of their bitmap words: passed -> a detection; deferred (beyond the candidate kernel's budget
// per block: constant input) -> onto a second list, which k_median_tests<true> tests from memory with a wave per
// candidate; else nothing.  ONE atomic per wave and list: the channel's counters are single addresses, and an atomic
// per 64 candidates (5 461 per 2^28 samples, one behind the other at the L2) was most of this kernel's first form.
// (Inside k_tile_visit's walk the same lookup cost 56 us per 2^28 samples without ever running.)
// Round 6: four waves per workgroup (one per SIMD, <= 32 registers: the workgroup still starts beside a correlator
// workgroup), one atomic per WORKGROUP and list (1366 atomics per list were 15 of the kernel's 23 us).
constexpr int kResolveWaves = 4;
__global__ __attribute__((amdgpu_flat_work_group_size(64 * kResolveWaves, 64 * kResolveWaves), amdgpu_num_vgpr(16))) // (pairs: 32)
void k_resolve_visited(unsigned long long A0, ChanState* __restrict__ st,
                                                        const uint32_t* __restrict__ visit, uint32_t visit_cap,
                                                        const unsigned long long* __restrict__ passmap, size_t bm_stride,
                                                        unsigned long long* __restrict__ det, uint32_t det_cap,
                                                        uint32_t* __restrict__ deferred)
{
    __shared__ uint32_t cnt_ok[kResolveWaves], cnt_def[kResolveWaves];
    __shared__ uint32_t base_ok, base_def;
    const uint32_t ch = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const ulonglong2* pm = reinterpret_cast<const ulonglong2*>(passmap) + static_cast<size_t>(ch) * bm_stride;
    const uint32_t* vis = visit + static_cast<size_t>(ch) * visit_cap;
    const uint32_t n = min(st[ch].vis_cnt, visit_cap);
    if (blockIdx.x * (256u * kResolveWaves) >= n) return; // (the whole workgroup)
    const uint32_t base = (blockIdx.x * kResolveWaves + static_cast<uint32_t>(wave)) * 256u;
    uint32_t p[4];
    bool ok[4], later[4];
    unsigned long long m[4], md[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // (coalesced: group k is the 64 consecutive entries base + 64 k ..)
        const uint32_t idx = base + 64u * k + lane;
        p[k] = idx < n ? vis[idx] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const ulonglong2 pd = pm[p[k] >> 6]; // (pass word, defer word)
        const bool valid = base + 64u * k + lane < n;
        ok[k] = valid && ((pd.x >> (p[k] & 63u)) & 1ull);
        later[k] = valid && ((pd.y >> (p[k] & 63u)) & 1ull);
        m[k] = __ballot(ok[k]);
        md[k] = __ballot(later[k]);
    }
    const uint32_t n_ok = static_cast<uint32_t>(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
    const uint32_t n_def = static_cast<uint32_t>(__popcll(md[0]) + __popcll(md[1]) + __popcll(md[2]) + __popcll(md[3]));
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
        cnt_ok[wave] = n_ok;
        cnt_def[wave] = n_def;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t_ok = 0, t_def = 0;
#pragma unroll
        for (int w = 0; w < kResolveWaves; ++w) {
            t_ok += cnt_ok[w];
            t_def += cnt_def[w];
        }
        base_ok = t_ok ? atomicAdd(&st[ch].det_cnt, t_ok) : 0u;
        base_def = t_def ? atomicAdd(&st[ch].def_cnt, t_def) : 0u;
    }
    __syncthreads();
    if (n_ok) {
        uint32_t slot = base_ok;
        for (int w = 0; w < wave; ++w) slot += cnt_ok[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ok[k]) {
                const uint32_t at = slot + static_cast<uint32_t>(__popcll(m[k] & below));
                if (at < det_cap) det[static_cast<size_t>(ch) * det_cap + at] = A0 + p[k];
                else st[ch].overflow = 1;
            }
            slot += static_cast<uint32_t>(__popcll(m[k]));
        }
    }
    if (n_def) {
        uint32_t slot = base_def;
        for (int w = 0; w < wave; ++w) slot += cnt_def[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (later[k]) {
                const uint32_t at = slot + static_cast<uint32_t>(__popcll(md[k] & below));
                if (at < visit_cap) deferred[static_cast<size_t>(ch) * visit_cap + at] = p[k];
                else st[ch].overflow = 1;
            }
            slot += static_cast<uint32_t>(__popcll(md[k]));
        }
    }
}

// sample at signed offset `o` relative to the first item of this call
__device__ __forceinline__ cf sample_at(const cf* cur, const cf* carry, uint32_t xc, long long o)
{
    return o >= 0 ? cur[o] : carry[static_cast<long long>(xc) + o];
}

// ------------------------------------------------------------------ k_tags
// one wave per pending detection; emits a raw record when the tag leaves in this call, i.e.
// pos + hist in [E0, E1).  FFT_NOISE: the noise power comes from a forward transform of the block containing pos (same
// arithmetic as k_correlate); otherwise k_correlate_w64 has left it behind the channel's powers (noise_rel) and the
// kernel needs neither the transform's LDS nor its registers.
// Round 6: eight waves per workgroup where the kernel needs no transform (FFT_NOISE = false), and the record slots of a
// workgroup's eight waves come from ONE atomic -- 2048 atomics with return on the channel's one counter were a sixth of
// the kernel's 87 us (the same serialisation k_tile_visit had); and only the bins that exist are accumulated (the groups
// of nine ran all nine for four bins).
template <bool FFT_NOISE>
constexpr int kTagWaves = FFT_NOISE ? 1 : 8;
template <bool FFT_NOISE, int BG = 9>
__global__ __launch_bounds__(64 * kTagWaves<FFT_NOISE>) void k_tags(const cf* __restrict__ in, size_t in_stride,
                                             const cf* __restrict__ carry, size_t carry_stride,
                                             uint32_t xc, unsigned long long E0, unsigned long long E1,
                                             uint32_t hist, uint32_t stride_s, int n_bins,
                                             const float4* __restrict__ tmpl, const cf* __restrict__ tw1a,
                                             const cf* __restrict__ tw1b, const cf* __restrict__ twA,
                                             const cf* __restrict__ twB, const cf* __restrict__ td, uint32_t td_len,
                                             const float* __restrict__ zcur, size_t z_stride,
                                             ChanState* __restrict__ st,
                                             const unsigned long long* __restrict__ det, uint32_t det_cap,
                                             RawTag* __restrict__ rec, uint32_t rec_cap, uint32_t noise_rel,
                                             uint32_t fft_n)
{
#ifndef GR4PM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3); // latency-bound, few waves: win the issue arbitration against throughput kernels
#endif
    constexpr int kW = kTagWaves<FFT_NOISE>;
    __shared__ cf lds[FFT_NOISE ? kExchangeItems : 1];
    __shared__ cf zbin_all[kW][kMaxBins];
    __shared__ uint32_t wave_cnt[2][kW]; // (by the parity of the round: a wave may be one round ahead of the slowest)
    __shared__ uint32_t wg_base;
    const uint32_t ch = blockIdx.y;
    const uint32_t n_det = min(st[ch].det_cnt, det_cap);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    cf* zbin = zbin_all[wave];
    const uint32_t gw = blockIdx.x * kW + static_cast<uint32_t>(wave), n_waves = gridDim.x * kW;
    // The grid strides over the pending detections (a launch of det_cap mostly empty workgroups cost more than the
    // records themselves): lane l of wave g looks at detection g + l * n_waves, the wave then works through the
    // ones that leave in this call one after the other.  The record slots of ALL of a workgroup's round come from ONE
    // atomic (round 4: one atomicAdd on the same word per record was half of the kernel's time at 10 000 records a call;
    // round 6: one per wave still a sixth).
    uint32_t parity = 0;
    for (uint32_t first = 0; first < n_det; first += 64u * n_waves, parity ^= 1u) { // (n_det: the same in every wave)
    const uint32_t my = first + gw + static_cast<uint32_t>(lane) * n_waves;
    unsigned long long my_pos = 0;
    bool leaves = false;
    if (my < n_det) {
        my_pos = det[static_cast<size_t>(ch) * det_cap + my];
        leaves = my_pos + hist >= E0 && my_pos + hist < E1;
    }
    unsigned long long todo = __ballot(leaves);
    if (lane == 0) wave_cnt[parity][wave] = static_cast<uint32_t>(__popcll(todo));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kW; ++w) total += wave_cnt[parity][w];
        wg_base = total ? atomicAdd(&st[ch].rec_cnt, total) : 0u;
    }
    __syncthreads();
    if (todo == 0) continue;
    unsigned int slot_base = wg_base;
    for (int w = 0; w < wave; ++w) slot_base += wave_cnt[parity][w];
    slot_base = __builtin_amdgcn_readfirstlane(slot_base);
    for (unsigned int taken = 0; todo != 0; ++taken, todo &= todo - 1) {
    const int src = __ffsll(static_cast<long long>(todo)) - 1;
    const unsigned long long pos =
        (static_cast<unsigned long long>(static_cast<unsigned int>(__shfl(static_cast<int>(my_pos >> 32), src))) << 32) |
        static_cast<unsigned int>(__shfl(static_cast<int>(my_pos & 0xffffffffu), src));
    wave_lds_sync();
    const unsigned long long blk = pos / stride_s;
    const uint32_t lag = static_cast<uint32_t>(pos - blk * stride_s);
    const long long o = static_cast<long long>(blk * stride_s) - static_cast<long long>(E0);
    const cf* cur = in + static_cast<size_t>(ch) * in_stride;
    const cf* car = carry + static_cast<size_t>(ch) * carry_stride;
    float noise = 0.0f;
    if constexpr (FFT_NOISE) {
        cf r[32];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            r[2 * n1] = sample_at(cur, car, xc, o + 2 * lane + 128 * n1);
            r[2 * n1 + 1] = sample_at(cur, car, xc, o + 2 * lane + 1 + 128 * n1);
        }
        fft1_wave(lane, r, lds, tw1a, tw1b);
        // noise power: bins N/4 .. 3N/4-1 == k3 in {2,3,4,5} (hpp:257-265)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k3 = 2; k3 < 6; ++k3) noise += cnorm(r[8 * q + k3]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) noise += __shfl_xor(noise, d);
    } else {
        (void)tw1a;
        (void)tw1b;
        (void)lds;
        // block b of this call, or the last block of the call before (b = -1): the carried value at [0]
        const long long b = static_cast<long long>(blk) - static_cast<long long>(E0 / stride_s);
        noise = zcur[static_cast<size_t>(ch) * z_stride + noise_rel + 1 + b];
    }
    // (fft_n: 2048, or 4096 behind k_correlate_4096, which leaves the block energies the same way; FFT_NOISE is 2048 only)
    noise /= static_cast<float>(fft_n / 2) * static_cast<float>(fft_n);

    // The correlation at the ONE lag of the detection, for every bin, straight from its definition instead of
    // n_bins more transforms: the value the reference reads at z_idx = (N - lag) mod N of FFT(X .* conj(FFT(s_b)))
    // (hpp:246-252,300) is N * sum_n x[pos + n] conj(s_b[n]) over the L = (64 - 1) sps + ntaps template samples
    // (no wrap: lag < S).  297 complex MACs per bin against 112 kFLOP per transform; accumulated in double
    // (closer to the exact sum than either float FFT).  td: [bin][L] conj(s_b[n]), the float template of hpp:166-182.
    (void)tmpl;
    (void)twA;
    (void)twB;
    // All loads of a group of bins are in flight before the first product (one lane-sample per 64 template samples,
    // the same for every bin; bin by bin every pass of the loop waited for its own two loads: 45 dependent round
    // trips per tag), and the bins' lane sums are reduced side by side.
    constexpr int kU = 5, kBinGroup = BG; // 320 template samples and BG bins per round: one round for the defaults
    constexpr int kV = 2 * BG <= 8 ? 8 : 2 * BG <= 16 ? 16 : 32; // the sums of a round, padded to a power of two
    static_assert(2 * BG <= kV, "bin group");
    for (int bin0 = 0; bin0 < n_bins; bin0 += kBinGroup) {
        double acc[kV]; // [2 b]: real part of bin b's lane sum, [2 b + 1]: imaginary part
#pragma unroll
        for (int v = 0; v < kV; ++v) acc[v] = 0.0;
        for (uint32_t n0 = 0; n0 < td_len; n0 += 64 * kU) {
            cf x[kU], t[kBinGroup][kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const uint32_t n = n0 + 64 * u + lane;
                x[u] = n < td_len ? sample_at(cur, car, xc, o + static_cast<long long>(lag) + n) : mk(0.f, 0.f);
            }
#pragma unroll
            for (int b = 0; b < kBinGroup; ++b) {
                if (bin0 + b < n_bins) { // (uniform)
                    const cf* tb = td + static_cast<size_t>(bin0 + b) * td_len;
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const uint32_t n = n0 + 64 * u + lane;
                        t[b][u] = n < td_len ? tb[n] : mk(0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < kBinGroup; ++b) {
                if (bin0 + b < n_bins) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) { // same order of accumulation per lane as the loop it replaces
                        acc[2 * b] += static_cast<double>(x[u].x) * t[b][u].x - static_cast<double>(x[u].y) * t[b][u].y;
                        acc[2 * b + 1] += static_cast<double>(x[u].x) * t[b][u].y + static_cast<double>(x[u].y) * t[b][u].x;
                    }
                }
            }
        }
        // The lane sums of all kV values across the wave, as the butterfly `v += shfl_xor(v, d)`, d = 32 .. 1, computes
        // them -- but at every step a lane keeps only the half of the values its bit d selects and hands the other half
        // to its partner, so that kV values cost kV - 1 + (6 - log2 kV) exchanges instead of 6 kV (8 values: 10, not 48;
        // the exchanges were 20 of the kernel's 73 us).  Own value + partner's value at every step, like the butterfly:
        // the same additions in the same order, the same bits.  Value v ends up in the lanes whose top log2 kV bits are v.
        int d = 32;
#pragma unroll
        for (int n = kV; n > 1; n >>= 1, d >>= 1) {
            const bool upper = (lane & d) != 0;
#pragma unroll
            for (int k = 0; k < n / 2; ++k) {
                const double send = upper ? acc[k] : acc[k + n / 2];
                const double keep = upper ? acc[k + n / 2] : acc[k];
                acc[k] = keep + __shfl_xor(send, d);
            }
        }
        for (; d > 0; d >>= 1) acc[0] += __shfl_xor(acc[0], d);
        constexpr int kLanesPerValue = 64 / kV;
        const int v = lane / kLanesPerValue;
        if (lane % kLanesPerValue == 0 && v < 2 * kBinGroup && bin0 + v / 2 < n_bins)
            reinterpret_cast<float*>(zbin + bin0 + v / 2)[v & 1] = static_cast<float>(acc[0] * fft_n);
    }
    wave_lds_sync();
    if (lane == 0) {
        int best = 0;
        cf z = mk(0.f, 0.f);
        float zp = -1.0f;
        for (int bin = 0; bin < n_bins; ++bin) { // hpp:305-313
            const float p = cnorm(zbin[bin]);
            if (p > zp) {
                best = bin;
                z = zbin[bin];
                zp = p;
            }
        }
        RawTag t;
        t.pos = pos;
        t.zx = z.x;
        t.zy = z.y;
        t.zpow = zp;
        t.left = best > 0 ? cnorm(zbin[best - 1]) : 0.0f;          // hpp:329-333
        t.right = best < n_bins - 1 ? cnorm(zbin[best + 1]) : 0.0f; // hpp:334-338
        const float* z0 = zcur + static_cast<size_t>(ch) * z_stride; // z0[0] <-> item E0
        const long long rel = static_cast<long long>(pos) - static_cast<long long>(E0);
        t.prev = z0[rel - 1]; // hpp:322 _history[_history_size]
        t.next = z0[rel + 1]; // hpp:323 _history[_history_size - 2]
        t.noise = noise;
        t.bin_idx = best;
        t.pad = 0;
        const unsigned int slot = slot_base + taken;
        if (slot < rec_cap) rec[static_cast<size_t>(ch) * rec_cap + slot] = t;
        else st[ch].overflow = 1;
    }
    } // detections of this wave that leave
    } // rounds over the pending list
}

// =====================================================================================
// Generic power-of-two block size (fft_size != 2048): one 256-thread workgroup per block,
// in-place radix-2 FFT in LDS (bit-reversed load, log2 N stages).  Same arithmetic contract
// as the tuned path (forward, un-normalised), far less tuned: it exists so that `fft_size`
// stays the free setting it is in the reference (syncword_detection.hpp:133).
// LDS: X (N cf) + work (N cf).  Templates natural order [bin][N]; twiddles tw[k] = W_N^k, k < N/2.
// =====================================================================================
__device__ __forceinline__ uint32_t bitrev(uint32_t v, int bits) { return __brev(v) >> (32 - bits); }

// in-place DIT FFT of `a` (already bit-reversed), all threads of the workgroup
__device__ __forceinline__ void wg_fft_inplace(cf* a, const cf* __restrict__ tw, uint32_t N, int log2n)
{
    for (int s = 1; s <= log2n; ++s) {
        const uint32_t half = 1u << (s - 1);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < N / 2; t += blockDim.x) {
            const uint32_t grp = t >> (s - 1), j = t & (half - 1);
            const uint32_t i0 = (grp << s) + j, i1 = i0 + half;
            const cf w = tw[j << (log2n - s)];
            const cf u = a[i0], v = cmulc(a[i1], w);
            a[i0] = u + v;
            a[i1] = u - v;
        }
    }
    __syncthreads();
}

// mode 0: zpow for all lags of block blockIdx.x.  mode 1 (k_tags_generic): see below.
__global__ __launch_bounds__(256) void k_correlate_generic(const cf* __restrict__ in, size_t in_stride,
                                                           uint32_t n_blocks, uint32_t stride_s, uint32_t N,
                                                           int log2n, int n_bins, const cf* __restrict__ tmpl,
                                                           const cf* __restrict__ tw, float* __restrict__ zpow,
                                                           size_t z_stride)
{
    extern __shared__ cf gl[];
    cf* X = gl;
    cf* W = gl + N;
    const uint32_t b = blockIdx.x;
    const cf* x = in + static_cast<size_t>(blockIdx.y) * in_stride + static_cast<size_t>(b) * stride_s;
    float* zo = zpow + static_cast<size_t>(blockIdx.y) * z_stride + static_cast<size_t>(b) * stride_s;
    for (uint32_t i = threadIdx.x; i < N; i += 256) X[bitrev(i, log2n)] = x[i];
    wg_fft_inplace(X, tw, N, log2n);
    constexpr int kMaxPer = 32; // N <= 8192
    float zmax[kMaxPer];
#pragma unroll
    for (int u = 0; u < kMaxPer; ++u) zmax[u] = -1.0f;
    for (int bin = 0; bin < n_bins; ++bin) {
        const cf* t = tmpl + static_cast<size_t>(bin) * N;
        for (uint32_t i = threadIdx.x; i < N; i += 256) W[bitrev(i, log2n)] = cmulc(X[i], t[i]);
        wg_fft_inplace(W, tw, N, log2n);
#pragma unroll
        for (int u = 0; u < kMaxPer; ++u) {
            const uint32_t lag = u * 256 + threadIdx.x;
            if (lag < stride_s) zmax[u] = fmaxf(zmax[u], cnorm(W[(N - lag) & (N - 1)]));
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < kMaxPer; ++u) {
        const uint32_t lag = u * 256 + threadIdx.x;
        if (lag < stride_s) zo[lag] = zmax[u];
    }
    (void)n_blocks;
}

// generic-size counterpart of k_tags: one 256-thread workgroup per pending detection
__global__ __launch_bounds__(256) void k_tags_generic(const cf* __restrict__ in, size_t in_stride,
                                                      const cf* __restrict__ carry, size_t carry_stride,
                                                      uint32_t xc, unsigned long long E0, unsigned long long E1,
                                                      uint32_t hist, uint32_t stride_s, uint32_t N, int log2n,
                                                      int n_bins, const cf* __restrict__ tmpl,
                                                      const cf* __restrict__ tw, const float* __restrict__ zcur,
                                                      size_t z_stride, ChanState* __restrict__ st,
                                                      const unsigned long long* __restrict__ det, uint32_t det_cap,
                                                      RawTag* __restrict__ rec, uint32_t rec_cap)
{
    extern __shared__ cf gl[];
    __shared__ cf zbin[kMaxBins];
    __shared__ float red[256];
    cf* X = gl;
    cf* W = gl + N;
    const uint32_t ch = blockIdx.y;
    const uint32_t n_det = min(st[ch].det_cnt, det_cap);
    for (uint32_t idx = blockIdx.x; idx < n_det; idx += gridDim.x) { // uniform per workgroup
    const unsigned long long pos = det[static_cast<size_t>(ch) * det_cap + idx];
    const unsigned long long c = pos + hist;
    if (c < E0 || c >= E1) continue;
    const unsigned long long blk = pos / stride_s;
    const uint32_t lag = static_cast<uint32_t>(pos - blk * stride_s);
    const long long o = static_cast<long long>(blk * stride_s) - static_cast<long long>(E0);
    const cf* cur = in + static_cast<size_t>(ch) * in_stride;
    const cf* car = carry + static_cast<size_t>(ch) * carry_stride;
    for (uint32_t i = threadIdx.x; i < N; i += 256) X[bitrev(i, log2n)] = sample_at(cur, car, xc, o + i);
    wg_fft_inplace(X, tw, N, log2n);
    float noise = 0.0f; // hpp:257-265
    for (uint32_t k = N / 4 + threadIdx.x; k < 3 * N / 4; k += 256) noise += cnorm(X[k]);
    red[threadIdx.x] = noise;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (static_cast<int>(threadIdx.x) < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    noise = red[0] / (static_cast<float>(N / 2) * static_cast<float>(N));
    for (int bin = 0; bin < n_bins; ++bin) {
        const cf* t = tmpl + static_cast<size_t>(bin) * N;
        for (uint32_t i = threadIdx.x; i < N; i += 256) W[bitrev(i, log2n)] = cmulc(X[i], t[i]);
        wg_fft_inplace(W, tw, N, log2n);
        if (threadIdx.x == 0) zbin[bin] = W[(N - lag) & (N - 1)];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int best = 0;
        cf z = mk(0.f, 0.f);
        float zp = -1.0f;
        for (int bin = 0; bin < n_bins; ++bin) {
            const float p = cnorm(zbin[bin]);
            if (p > zp) {
                best = bin;
                z = zbin[bin];
                zp = p;
            }
        }
        RawTag t;
        t.pos = pos;
        t.zx = z.x;
        t.zy = z.y;
        t.zpow = zp;
        t.left = best > 0 ? cnorm(zbin[best - 1]) : 0.0f;
        t.right = best < n_bins - 1 ? cnorm(zbin[best + 1]) : 0.0f;
        const float* z0 = zcur + static_cast<size_t>(ch) * z_stride;
        const long long rel = static_cast<long long>(pos) - static_cast<long long>(E0);
        t.prev = z0[rel - 1];
        t.next = z0[rel + 1];
        t.noise = noise;
        t.bin_idx = best;
        t.pad = 0;
        const unsigned int slot = atomicAdd(&st[ch].rec_cnt, 1u);
        if (slot < rec_cap) rec[static_cast<size_t>(ch) * rec_cap + slot] = t;
        else st[ch].overflow = 1;
    }
    __syncthreads();
    } // grid-stride loop over the pending detections
}

// drop emitted detections (pos + hist < E1) from the pending list; one workgroup per channel.  Round 6: what stays is
// what lies in the call's last `hist` items -- a handful of ten thousand -- so the list is read once by 256 threads, the
// survivors are collected in LDS and written back to the front behind a barrier (46 -> 5 us per 2^28 samples; the one
// wave that walked the list 64 entries at a time stays as the path for more survivors than the LDS list holds).
// 256 threads, 4 KiB of LDS: four waves of <= 16 registers, one per SIMD -- the workgroup starts BESIDE a correlator
// workgroup (32 VGPRs per SIMD and 11 KiB of LDS are what that leaves: tools/check_occupancy.py), as the one wave of rounds
// 1 - 5 did.
constexpr int kCompactThreads = 256, kCompactKeep = 512;
__global__ __launch_bounds__(kCompactThreads) void k_compact_pending(ChanState* __restrict__ st,
                                                                     ChanState* __restrict__ st_host,
                                                                     unsigned long long* __restrict__ det, uint32_t det_cap,
                                                                     unsigned long long E1, uint32_t hist, int n_channels)
{
    __shared__ unsigned long long kept[kCompactKeep];
    __shared__ uint32_t n_kept;
    const int ch = blockIdx.x;
    if (ch >= n_channels) return;
    const int tid = threadIdx.x, lane = tid & 63;
    // the host's view of this call (record count, overflow flag) goes straight to pinned memory
    if (tid == 0) {
        st_host[ch] = st[ch];
        n_kept = 0;
    }
    unsigned long long* d = det + static_cast<size_t>(ch) * det_cap;
    const uint32_t n = min(st[ch].det_cnt, det_cap);
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kCompactThreads) {
        const unsigned long long p = d[i];
        if (p + hist >= E1) {
            const uint32_t at = atomicAdd(&n_kept, 1u); // (LDS; survivors are rare)
            if (at < kCompactKeep) kept[at] = p;
        }
    }
    __syncthreads(); // every read of the list lies in front of the writes below
    const uint32_t w = n_kept;
    if (w <= kCompactKeep) {
        for (uint32_t i = tid; i < w; i += kCompactThreads) d[i] = kept[i];
    } else if (tid < 64) { // more survivors than the LDS list holds: the in-place walk of rounds 1 - 5, by one wave
        uint32_t ww = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            const unsigned long long p = i < n ? d[i] : 0ull;
            const bool keep = i < n && p + hist >= E1;
            const unsigned long long m = __ballot(keep);
            const uint32_t off = __popcll(m & ((1ull << lane) - 1ull));
            if (keep) d[ww + off] = p; // all reads of this chunk happened above; writes land at indices <= i
            ww += __popcll(m);
        }
    }
    if (tid == 0) {
        st[ch].det_cnt = w;
        st[ch].rec_cnt = 0; // the host copy of ChanState was written above
        st[ch].vis_cnt = 0;
        st[ch].def_cnt = 0;
    }
}

// out[i] = item (i - hist) of the stream: the 2T+1 delay of hpp:318-319,342
__global__ __launch_bounds__(256) void k_delay_copy(const cf* __restrict__ in, size_t in_stride,
                                                    const cf* __restrict__ carry, size_t carry_stride,
                                                    uint32_t xc, uint32_t hist, size_t n,
                                                    cf* __restrict__ out, size_t out_stride)
{
    const cf* cur = in + static_cast<size_t>(blockIdx.y) * in_stride;
    const cf* car = carry + static_cast<size_t>(blockIdx.y) * carry_stride;
    cf* o = out + static_cast<size_t>(blockIdx.y) * out_stride;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        o[i] = sample_at(cur, car, xc, static_cast<long long>(i) - static_cast<long long>(hist));
    }
}

// next carry = last xc items of (carry ++ in[0..n))
__global__ void k_update_carry(const cf* __restrict__ in, size_t in_stride, const cf* __restrict__ carry,
                               cf* __restrict__ carry_next, size_t carry_stride, uint32_t xc, size_t n)
{
    const cf* cur = in + static_cast<size_t>(blockIdx.y) * in_stride;
    const cf* car = carry + static_cast<size_t>(blockIdx.y) * carry_stride;
    cf* nx = carry_next + static_cast<size_t>(blockIdx.y) * carry_stride;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= xc) return;
    nx[i] = sample_at(cur, car, xc, static_cast<long long>(n) - static_cast<long long>(xc) + i);
}

// zcur head (zc items before item E0) = tail of the previous call's z buffer
__global__ void k_update_zcarry(const float* __restrict__ zprev, float* __restrict__ zcur, size_t z_stride,
                                uint32_t zc, size_t n_prev, uint32_t noise_off, uint32_t n_prev_blocks)
{
    const float* p = zprev + static_cast<size_t>(blockIdx.y) * z_stride;
    float* c = zcur + static_cast<size_t>(blockIdx.y) * z_stride;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the noise power of the last block of the call before: a detection in its last 2T + 1 items is emitted by this call
    if (i == 0) c[noise_off] = n_prev_blocks ? p[noise_off + n_prev_blocks] : 0.0f;
    if (i >= zc) return;
    c[i] = p[n_prev + i]; // prev layout: [zc carry][n_prev items]; take its last zc entries
}

// double precision radix-2 FFT for the one-time template build
void fft_double(std::vector<std::complex<double>>& a)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                const double ang = -2.0 * M_PI * static_cast<double>(k) / static_cast<double>(len);
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                const auto u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
        }
    }
}

} // namespace
} // namespace gr4pm

using namespace gr4pm;

struct gr4pm_syncword_detection {
    // settings
    size_t fft_size, sps, n_channels, max_items;
    int min_bin, max_bin, n_bins;
    uint64_t T;
    float power_threshold;
    hipStream_t stream;
    // derived (hpp:148-164,194,236)
    size_t L, S, hist;
    float self_corr;
    // geometry of carried state
    uint32_t xc, zc;
    size_t z_stride, bm_stride, table_stride;
    uint32_t noise_off = 0; // per channel row of a z set: [zc carried powers][items][...] and, from here, one noise power
                            // per block: [0] the last block of the call before, [1 + b] block b (k_correlate_w64 writes them)
    uint32_t max_tiles, det_cap, rec_cap;
    // device
    DevBuf<float4> tmpl;
    DevBuf<cf> tw; // tw1a ++ tw1b ++ twA ++ twB (fft2048_wave.hpp)
    // the one-exchange correlator (fft2048_w64.hpp, k_correlate_w64): templates in its lane order
    // ([bin][16][64] float4), mid-stage twiddle table, lane constants
    bool lds_candidates = false; // GR4PM_CANDIDATES_LDS at creation: k_candidates also for T = 768
    bool separate_median = false; // GR4PM_SD_SEPARATE_MEDIAN at creation: k_candidates_wave<12, false> + k_median_tests
    bool w64_no_prune = false;   // GR4PM_W64_NO_PRUNE at creation: the correlator variant that computes all 32 registers
    int corr_kind = 0; // 0: k_correlate_w64 (default), 1: k_correlate (two exchanges), 2: k_correlate_pair
    DevBuf<float4> tmpl64, tT64;
    DevBuf<cf> td; // [bin][L] conj of the float time-domain templates (hpp:166-182): k_tags' direct correlation
    DevBuf<cf> cc64;
    int n_cus = 256;
    int w64_variant = -1;
    int w64_one = -1; // GR4PM_W64_ONE at creation; -1 (unset): the one-bin kernel unless the handle shares its CUs
    bool coresident = false; // set by the receivers: kernels of other pipeline stages run beside the correlator
    uint32_t w64_blocks_per_wave = 6; // blocks per wave and workgroup (handed out dynamically); 0: persistent waves (GR4PM_W64_BLOCKS_PER_WAVE)
    // raised by a correlator kernel whose bounded hand-off spin ran out ("wave" / "pair" kernels; k_correlate_w64
    // has no spins); checked after the stream synchronisation of process()
    PinnedBuf<unsigned> fault;
    int spin_limit = 1 << 20;
    // the two-waves-per-block correlator (fft2048_pair.hpp): templates in its lane order,
    // tw1p ++ twAp ++ twB
    bool use_pair = false;
    DevBuf<float4> tmplp;
    DevBuf<cf> twp;
    // generic block sizes (fft_size != 2048)
    bool generic = false;
    int log2n = 0;
    DevBuf<cf> g_tmpl, g_tw;
    DevBuf<cf> tw4k;          // fft_size 4096: tw1 ++ tw2 of fft4096_wg.hpp
    int c4096_variant = 1;    // GR4PM_C4096_VARIANT: which form of k_correlate_4096 runs (1: pass-1 twiddles in registers)
    bool force_radix2 = false; // GR4PM_CORRELATOR=radix2: the generic kernel also for 4096 (tests compare the two)
    DevBuf<cf> carry[kCarry]; // sample carry before this call, before the calls ahead, and the one being written
    DevBuf<float> z[kSets];
    // candidate bitmap, tile tables and group tables exist kSets times like z[]: the look-ahead of
    // the next calls fills the other sets while this call's scan still reads its own
    DevBuf<unsigned long long> bitmap[kSets];
    // per bitmap word a pair: candidates that pass the median test | candidates k_candidates_wave<12, true> left untested
    // (beyond its budget per block: k_median_tests if the scan visits them)
    DevBuf<unsigned long long> passmap[kSets];
    bool fused_median[kSets] = {};             // ... and whether the set's front was made by that kernel
    bool last_fused = false;                   // ... of the set the last process() call scanned (scan_counts)
    DevBuf<uint32_t> table[kSets];
    DevBuf<unsigned long long> gtable[kSets], gentry, sgtable[kSets];
    size_t gtable_stride = 0, sgtable_stride = 0;
    bool super_level = true; // (GR4PM_SD_NO_SUPER at creation: the two-level scan of rounds 1 - 5, for A/B)
    uint32_t max_groups = 0;
    DevBuf<int32_t> entry;
    DevBuf<ChanState> st;
    DevBuf<unsigned long long> det;
    DevBuf<uint32_t> visit; // candidates visited by the scan of the current call, per channel
    DevBuf<uint32_t> deferred; // ... of which the fused candidate kernel left untested (k_resolve_visited), per channel
    uint32_t visit_cap = 0;
    PinnedBuf<ChanState> st_host;
    PinnedBuf<RawTag> rec_host; // written by k_tags through the device-visible mapping
    // stream position
    uint64_t items_consumed = 0;
    int cur = 0;          // which of z[] / bitmap[] / table[] / gtable[] is current
    int ci = 0;           // carry[ci] = the last xc items before items_consumed
    size_t last_done = 0; // items of the last call (for the z carry)
    // look-ahead (gr4pm_syncword_detection_announce / _hint_next): the front part of the next
    // kAhead calls (z carry, correlator, candidates, tables) runs on two more streams while this
    // call's scan, tag kernels and read-back are in flight
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_zcarry = nullptr, ev_mid[kSets] = {}, ev_front[kSets] = {};
    bool front_recorded[kSets] = {}; // ev_front[i] has been recorded at least once (never wait for a fresh event)
    struct Ahead {
        const gr4pm_c64* in;
        size_t stride, n;
        uint64_t E0; // first item of that call
        int set;     // buffer set its front writes
        int ev;      // index into ev_announce: recorded on the handle's stream when the input was announced
    };
    // an announced buffer is only read by the look-ahead streams after everything that was queued on the
    // handle's stream at announcement time (the producer of that buffer, in stream-ordered code) has run
    hipEvent_t ev_announce[kAhead + 2] = {};
    int ev_next = 0;
    std::deque<Ahead> launched;  // fronts in flight or done, oldest first
    std::deque<Ahead> announced; // inputs announced but not launched yet
    ~gr4pm_syncword_detection()
    {
        if (stream2) {
            (void)hipStreamSynchronize(stream2);
            (void)hipStreamDestroy(stream2);
        }
        if (stream3) {
            (void)hipStreamSynchronize(stream3);
            (void)hipStreamDestroy(stream3);
        }
        for (auto e : ev_mid)
            if (e) (void)hipEventDestroy(e);
        for (auto e : ev_announce)
            if (e) (void)hipEventDestroy(e);
        for (auto e : ev_front)
            if (e) (void)hipEventDestroy(e);
        if (ev_zcarry) (void)hipEventDestroy(ev_zcarry);
    }
};

namespace {

gr4pm_status sd_reset(gr4pm_syncword_detection* h)
{
    h->items_consumed = 0;
    h->cur = 0;
    h->last_done = 0;
    if (h->stream2) GR4PM_HIP_TRY(hipStreamSynchronize(h->stream2));
    if (h->stream3) GR4PM_HIP_TRY(hipStreamSynchronize(h->stream3));
    h->launched.clear();
    h->announced.clear();
    h->ci = 0;
    for (int i = 0; i < kCarry; ++i) GR4PM_TRY(h->carry[i].zero(h->stream));
    for (int i = 0; i < kSets; ++i) GR4PM_TRY(h->z[i].zero(h->stream));
    GR4PM_TRY(h->st.zero(h->stream));
    GR4PM_TRY(h->det.zero(h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    return GR4PM_OK;
}

// output_tag(), hpp:56-115, evaluated on the host with the C library the reference uses
void finish_tag(const gr4pm_syncword_detection* h, const RawTag& t, uint64_t out_index, gr4pm_tag* o)
{
    const int freq_bin = h->min_bin + t.bin_idx;
    const double bin_spacing = M_PI / static_cast<double>(h->L);
    double syncword_freq = static_cast<double>(freq_bin) * bin_spacing;
    float syncword_phase = std::arg(std::complex<float>(t.zx, t.zy));
    float correlation_power;
    const float pi_f = 3.14159265358979323846f;
    if (freq_bin > h->min_bin && freq_bin < h->max_bin) {
        const double a = t.left, b = t.zpow, c = t.right;
        const double quad = std::clamp((c - a) / (2.0 * (2.0 * b - (a + c))), -0.5, 0.5);
        const double delta_freq = quad * bin_spacing;
        syncword_freq += delta_freq;
        syncword_phase -= static_cast<float>(delta_freq * 0.5 * static_cast<double>(h->L));
        if (syncword_phase >= pi_f) {
            syncword_phase -= 2.0f * pi_f;
        } else if (syncword_phase < -pi_f) {
            syncword_phase += 2.0f * pi_f;
        }
        correlation_power = static_cast<float>(b + (c - a) * (c - a) / (16.0 * (b - 0.5 * (a + c))));
    } else {
        correlation_power = t.zpow;
    }
    const float amplitude =
        std::sqrt(correlation_power) / (static_cast<float>(h->fft_size) * h->self_corr);
    const float syncword_power = amplitude * amplitude * h->self_corr;
    const float esn0_db =
        10.0f * std::log10((syncword_power * static_cast<float>(h->sps)) /
                           (t.noise * static_cast<float>(h->L)));
    const double a = t.prev, b = t.zpow, c = t.next;
    const float time_est =
        static_cast<float>(std::clamp((c - a) / (2.0 * (2.0 * b - (a + c))), -0.5, 0.5));
    o->index = out_index;
    o->amplitude = amplitude;
    o->phase = syncword_phase;
    o->freq = syncword_freq;
    o->freq_bin = freq_bin;
    o->noise_power = t.noise;
    o->esn0_db = esn0_db;
    o->time_est = time_est;
    o->flags = GR4PM_TAG_SYNCWORD;
    o->user = 0;
}

} // namespace
namespace gr4pm {
// (library-internal, common.hpp) the receivers tell their detector that it shares the CUs with the rest of the chain
void sd_set_coresident(gr4pm_syncword_detection* h, bool on)
{
    if (h) h->coresident = on;
}
} // namespace gr4pm
namespace {
gr4pm_status launch_correlate(gr4pm_syncword_detection* h, hipStream_t stream, const gr4pm_c64* in,
                              size_t in_stride, uint32_t n_blocks, float* zout)
{
    if (h->generic && h->fft_size == static_cast<size_t>(kN4k) && !h->force_radix2) {
#define GR4PM_C4096(V)                                                                                                   \
    hipLaunchKernelGGL(k_correlate_4096<V>, dim3(n_blocks, static_cast<unsigned>(h->n_channels)), dim3(kT4k), 0, stream,  \
                       reinterpret_cast<const cf*>(in), in_stride, n_blocks, static_cast<uint32_t>(h->S), h->n_bins,      \
                       h->g_tmpl.p, h->tw4k.p, h->tw4k.p + 16 * 256, zout, h->z_stride, h->noise_off - h->zc)
        switch (h->c4096_variant) { // GR4PM_C4096_VARIANT (bit-identical forms, correlate_4096.hpp)
        case 0: GR4PM_C4096(0); break;
        case 1: GR4PM_C4096(1); break;
        case 2: GR4PM_C4096(2); break;
        case 3: GR4PM_C4096(3); break;
        case 4: GR4PM_C4096(4); break;
        case 5: GR4PM_C4096(5); break;
        case 6: GR4PM_C4096(6); break;
        case 7: GR4PM_C4096(7); break;
        case 17: GR4PM_C4096(17); break; // A/B: variant 1 with round 4's power stores, same results
#ifdef GR4PM_EXPERIMENTS
        case 9: GR4PM_C4096(9); break; // (FMA-form butterflies: not bit-identical; announced at creation)
#endif
        default: GR4PM_C4096(1); break;
        }
#undef GR4PM_C4096
        GR4PM_HIP_TRY(hipGetLastError());
        return GR4PM_OK;
    }
    if (h->generic) {
        const uint32_t N = static_cast<uint32_t>(h->fft_size);
        hipLaunchKernelGGL(k_correlate_generic, dim3(n_blocks, static_cast<unsigned>(h->n_channels)), dim3(256),
                           2 * N * sizeof(cf), stream, reinterpret_cast<const cf*>(in), in_stride, n_blocks,
                           static_cast<uint32_t>(h->S), N, h->log2n, h->n_bins, h->g_tmpl.p, h->g_tw.p, zout,
                           h->z_stride);
        GR4PM_HIP_TRY(hipGetLastError());
        return GR4PM_OK;
    }
    if (h->use_pair) {
        dim3 gridp((n_blocks + kPairBlocksPerWg - 1) / kPairBlocksPerWg, static_cast<unsigned>(h->n_channels));
        hipLaunchKernelGGL(k_correlate_pair, gridp, dim3(kPairThreads), 0, stream, reinterpret_cast<const cf*>(in),
                           in_stride, n_blocks, static_cast<uint32_t>(h->S), h->n_bins, h->tmplp.p, h->twp.p,
                           h->tw.p + kTw1aItems, reinterpret_cast<const float4*>(h->twp.p + kTw1pItems), zout,
                           h->z_stride, h->fault.p, h->spin_limit);
        GR4PM_HIP_TRY(hipGetLastError());
        return GR4PM_OK;
    }
    if (h->corr_kind == 0) {
        // persistent waves: one 8-wave workgroup per CU, every wave walks items wave, wave + W, ...
        const uint32_t total = n_blocks * static_cast<uint32_t>(h->n_channels);
        const uint32_t bpw = h->w64_blocks_per_wave;
        // The grid for workgroups of about `waves` x bpw items each: a whole number of rounds of one workgroup per CU
        // (both kernels take a CU's LDS), so that no CU idles through most of a last, partly filled round -- the
        // 2^28-sample launch was 12.47 rounds at nine bins and 8.31 at one; a launch of less than a round gets one item
        // per wave on as many CUs as it can use.  GR4PM_W64_BALANCED=0: round 5's fixed shares of waves x bpw items.
        static const bool balanced = !(getenv("GR4PM_W64_BALANCED") && getenv("GR4PM_W64_BALANCED")[0] == '0');
        auto grid_for_items = [&](uint32_t waves) -> uint32_t {
            const uint32_t cus = static_cast<uint32_t>(h->n_cus);
            if (!bpw) return std::min<uint32_t>(cus, (total + waves - 1) / waves);
            const uint32_t g0 = (total + waves * bpw - 1) / (waves * bpw);
            if (!balanced) return g0;
            if (g0 <= cus) return std::max<uint32_t>(g0, std::min<uint32_t>(cus, (total + waves - 1) / waves));
            return cus * std::max<uint32_t>(1u, (g0 + cus / 2) / cus);
        };
        const uint32_t wgs = grid_for_items(kW64Waves);
#define GR4PM_W64_LAUNCH(V)                                                                                         \
    hipLaunchKernelGGL(k_correlate_w64<V>, dim3(wgs), dim3(kW64Threads), 0, stream, reinterpret_cast<const cf*>(in), \
                       in_stride, n_blocks, total, static_cast<uint32_t>(h->S), h->n_bins, h->tmpl64.p, h->tT64.p,  \
                       h->cc64.p, zout, h->z_stride, bpw, h->noise_off - h->zc)
        // GR4PM_W64_VARIANT: the timing-only ablations of tools/w64_variants.py
        // registers 1 .. 3 of the output hold lags >= 1793: with a stride of at most that (the default: 1752) the
        // kernel variant that never computes their powers runs (bit 16384)
        const bool prune = h->S <= 1793 && !h->w64_no_prune;
        // one frequency bin: GR4PM_W64_ONE=1 runs the three-waves-per-SIMD kernel of correlate_w64_one.hpp instead of the
        // general one (bit-identical powers, the same launch time: HISTORY.md section 3a; 2, 3: its timing-only ablations)
        // A stand-alone SyncwordDetection with min_freq_bin == max_freq_bin has the chip to itself: the dedicated kernel
        // (12 waves per CU, 4 % faster alone, bit-identical) is the default.  Inside a receiver the general kernel
        // stays: the one-bin kernel leaves 14 VGPRs per SIMD and nothing of the chain fits beside it.
        const int one_mode = h->w64_one;
        const bool one_off = one_mode == 0 || (one_mode < 0 && h->coresident);
        if (h->n_bins == 1 && prune && h->w64_variant < 0 && !one_off) {
            const uint32_t wgs1 = grid_for_items(kW1Waves);
#define GR4PM_W1_LAUNCH(A)                                                                                           \
    hipLaunchKernelGGL(k_correlate_w64_one<A>, dim3(wgs1), dim3(kW1Threads), 0, stream, reinterpret_cast<const cf*>(in), \
                       in_stride, n_blocks, total, static_cast<uint32_t>(h->S), h->tmpl64.p, h->tT64.p, h->cc64.p, zout, \
                       h->z_stride, bpw, h->noise_off - h->zc)
#ifdef GR4PM_EXPERIMENTS
            if (one_mode == 2) GR4PM_W1_LAUNCH(1);
            else if (one_mode == 3) GR4PM_W1_LAUNCH(2);
            else
#endif
                GR4PM_W1_LAUNCH(0);
#undef GR4PM_W1_LAUNCH
            GR4PM_HIP_TRY(hipGetLastError());
            return GR4PM_OK;
        }
        switch (h->w64_variant) {
#ifdef GR4PM_EXPERIMENTS // timing-only ablations (wrong powers): tools/build_variant.sh builds them, build() does not
        case 8: GR4PM_W64_LAUNCH(8); break;
        case 32: GR4PM_W64_LAUNCH(32); break;
        case 232: GR4PM_W64_LAUNCH(232); break;
        case 1256: GR4PM_W64_LAUNCH(1256); break;
        case 2048: GR4PM_W64_LAUNCH(2048); break;
        case 4096: GR4PM_W64_LAUNCH(4096); break;
        case 6144: GR4PM_W64_LAUNCH(6144); break;
        case 100352: GR4PM_W64_LAUNCH(98304 + 16384 + 2048); break; // the default kernel without power stores
        case 102400: GR4PM_W64_LAUNCH(98304 + 16384 + 4096); break; // ... without sample loads after the first block
        case 104448: GR4PM_W64_LAUNCH(98304 + 16384 + 6144); break; // ... without either
#endif
        case 0: // round 2's bin loop: everything (re, im) interleaved, templates by LDS-DMA into the exchange buffer
            if (prune) GR4PM_W64_LAUNCH(16384);
            else GR4PM_W64_LAUNCH(0);
            break;
        case 65536: GR4PM_W64_LAUNCH(65536 + 16384); break; // planar second half, templates still by LDS-DMA
        case 131072: GR4PM_W64_LAUNCH(98304 + 16384 + 131072); break; // A/B: fixed shares of blocks per wave (rounds 1 - 3), same results
        case 262144: GR4PM_W64_LAUNCH(98304 + 16384 + 262144); break; // A/B: round 4's power stores (a lane compare and a branch per store), same results
        default:
            // planar mid stage / pass B / powers in the bin loop (65536), templates from global memory straight
            // into registers (32768)
            if (prune) GR4PM_W64_LAUNCH(98304 + 16384);
            else GR4PM_W64_LAUNCH(98304);
            break;
        }
#undef GR4PM_W64_LAUNCH
        GR4PM_HIP_TRY(hipGetLastError());
        return GR4PM_OK;
    }
    dim3 grid((n_blocks + kWavesPerWg - 1) / kWavesPerWg, static_cast<unsigned>(h->n_channels));
    const cf* tw1a = h->tw.p;
    const cf* tw1b = tw1a + kTw1aItems;
    const cf* twA = tw1b + kTw1bItems;
    hipLaunchKernelGGL(k_correlate, grid, dim3(kCorrThreads), 0, stream, reinterpret_cast<const cf*>(in),
                       in_stride, n_blocks, static_cast<uint32_t>(h->S), h->n_bins, h->tmpl.p, tw1a, tw1b,
                       reinterpret_cast<const float4*>(twA), zout, h->z_stride, h->fault.p, h->spin_limit);
    GR4PM_HIP_TRY(hipGetLastError());
    return GR4PM_OK;
}

// everything of a call that does not depend on the scan state left by the call before it:
// z carry (tail of the other z buffer), correlation powers, candidate bitmap, tile tables and
// group tables, written to buffer set `which`.  `E0` = first item of the call, `n_prev` = items
// of the call before it, carry[ci] = the samples before E0 (carry[ci + 1] is written).
gr4pm_status launch_front(gr4pm_syncword_detection* h, hipStream_t stream, int which, int ci,
                          const gr4pm_c64* in, size_t in_stride, size_t n_in, uint64_t E0, size_t n_prev)
{
    const uint32_t n_blocks = static_cast<uint32_t>((n_in - h->fft_size) / h->S + 1); // hpp:238
    const size_t J = static_cast<size_t>(n_blocks) * h->S;
    const uint64_t E1 = E0 + J;
    const uint32_t T = static_cast<uint32_t>(h->T);
    const unsigned nch = static_cast<unsigned>(h->n_channels);
    float* zw = h->z[which].p;
    // the correlator needs nothing but the input; behind it, on the look-ahead's second stream, the
    // two carries (two tiny kernels that would otherwise sit between consecutive correlator launches)
    // and the candidate / table kernels
    GR4PM_TRY(launch_correlate(h, stream, in, in_stride, n_blocks, zw + h->zc));
    const bool ahead = stream != h->stream;
    static const bool one_front_stream = getenv("GR4PM_SD_FRONT_ONE_STREAM") != nullptr;
    if (ahead && !one_front_stream) {
        GR4PM_HIP_TRY(hipEventRecord(h->ev_mid[which], stream));
        stream = h->stream3;
        GR4PM_HIP_TRY(hipStreamWaitEvent(stream, h->ev_mid[which], 0));
    }
    // z carry: positions E0-zc .. E0-1 (the tail of the call before, complete once this call's
    // correlator -- behind that call's on its stream -- is); sample carry for the call after this
    // one: the last xc items up to E1
    hipLaunchKernelGGL(k_update_zcarry, dim3((h->zc + 255) / 256, nch), dim3(256), 0, stream,
                       h->z[(which + kSets - 1) % kSets].p, zw, h->z_stride, h->zc, n_prev, h->noise_off,
                       static_cast<uint32_t>(n_prev / h->S));
    hipLaunchKernelGGL(k_update_carry, dim3((h->xc + 255) / 256, nch), dim3(256), 0, stream,
                       reinterpret_cast<const cf*>(in), in_stride, h->carry[ci].p, h->carry[(ci + 1) % kCarry].p,
                       static_cast<size_t>(h->xc), h->xc, J);
    const uint64_t A0 = E0 > T ? E0 - T : 0, A1 = E1 > T ? E1 - T : 0;
    const uint32_t cnt = static_cast<uint32_t>(A1 - A0);
    if (cnt == 0) {
        if (ahead) {
            GR4PM_HIP_TRY(hipEventRecord(h->ev_front[which], stream));
            h->front_recorded[which] = true;
        }
        return GR4PM_OK;
    }
    const float* zloc = zw + h->zc - static_cast<ptrdiff_t>(E0 - A0);
    const uint32_t n_wg = (cnt + kCandTile - 1) / kCandTile;
    const size_t smem = (static_cast<size_t>(kCandBlocks + (T >> 6) + 2) * 65 + kCandBlocks) * sizeof(float);
    h->fused_median[which] = false;
    if (T == 768 && !h->lds_candidates) { // the LDS-free form: runs beside the correlator's workgroups
        const uint32_t n_words = n_wg * (kCandTile / 64), n_blk = (n_words + 11) / 12;
        if (!h->separate_median) { // (GR4PM_SD_SEPARATE_MEDIAN at creation: round 4's two passes over the powers, for A/B)
            // blocks per wave: a wave also reads (without scanning it) the block before its first one.  Fewer, longer
            // waves cost less overlap and balance worse (21 845 waves of 16 blocks on 7168 wave slots are 3.05 rounds)
            static const uint32_t kChain = experiment_env_wg("GR4PM_CAND_CHAIN", 8u, 1u, 64u);
            h->fused_median[which] = true;
            hipLaunchKernelGGL((k_candidates_wave<12, true>), dim3((n_blk + kChain - 1) / kChain, nch), dim3(64), 0, stream,
                               zloc, h->z_stride, cnt, n_words, kChain, h->bitmap[which].p, h->bm_stride,
                               h->passmap[which].p, nullptr, h->power_threshold);
        } else {
            constexpr uint32_t kChain = 8;
            hipLaunchKernelGGL((k_candidates_wave<12, false>), dim3((n_blk + kChain - 1) / kChain, nch), dim3(64), 0, stream,
                               zloc, h->z_stride, cnt, n_words, kChain, h->bitmap[which].p, h->bm_stride, nullptr, nullptr, 0.0f);
        }
    } else {
        hipLaunchKernelGGL(k_candidates, dim3(n_wg, nch), dim3(256), smem, stream, zloc, h->z_stride, cnt, T,
                           h->bitmap[which].p, h->bm_stride);
    }
    const uint32_t n_tiles = (cnt + kTileW - 1) / kTileW;
    hipLaunchKernelGGL(k_tile_tables, dim3(n_tiles, nch), dim3(64), 0, stream, h->bitmap[which].p, h->bm_stride,
                       cnt, T, n_tiles, h->table[which].p, h->table_stride);
    const uint32_t n_groups = (n_tiles + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_group_tables, dim3((T + 1 + 127) / 128, n_groups, nch), dim3(128), 0, stream, cnt, T,
                       n_tiles, h->table[which].p, h->table_stride, h->gtable[which].p, h->gtable_stride);
    if (h->super_level && n_groups > kSuper)
        hipLaunchKernelGGL(k_super_tables, dim3((T + 1 + 127) / 128, (n_groups + kSuper - 1) / kSuper, nch), dim3(128), 0,
                           stream, cnt, T, n_tiles, n_groups, h->gtable[which].p, h->gtable_stride, h->sgtable[which].p,
                           h->sgtable_stride);
    GR4PM_HIP_TRY(hipGetLastError());
    if (ahead) {
        GR4PM_HIP_TRY(hipEventRecord(h->ev_front[which], stream));
        h->front_recorded[which] = true;
    }
    return GR4PM_OK;
}

gr4pm_status ensure_ahead_streams(gr4pm_syncword_detection* h)
{
    if (h->stream2) return GR4PM_OK;
    int least = 0, greatest = 0;
    GR4PM_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // correlator: lowest priority: look-ahead work yields to the current call's detector kernels,
    // and a priority of its own also gives the stream a hardware queue of its own (streams of one
    // priority share GPU_MAX_HW_QUEUES = 4 queues; a shared queue serialises its kernels)
    GR4PM_HIP_TRY(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, least));
    // the candidate / table kernels behind it: the lowest priority as well.  Their stream starts every
    // front with a wait for the correlator; at the caller's (usually highest) priority it can end up
    // in the caller's hardware queue, and the caller's scan / tag kernels then sit behind that wait
    // for a whole correlator launch (measured with 64 channels: 7 instead of 2 ms per call).  The
    // default priority, which the receiver's stage streams have, costs 20 % for the same reason.
    GR4PM_HIP_TRY(hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, least));
    (void)greatest;
    GR4PM_HIP_TRY(hipEventCreateWithFlags(&h->ev_zcarry, hipEventDisableTiming));
    for (int i = 0; i < kSets; ++i) {
        GR4PM_HIP_TRY(hipEventCreateWithFlags(&h->ev_mid[i], hipEventDisableTiming));
        GR4PM_HIP_TRY(hipEventCreateWithFlags(&h->ev_front[i], hipEventDisableTiming));
    }
    return GR4PM_OK;
}

} // namespace

extern "C" {

gr4pm_status gr4pm_syncword_detection_create(const gr4pm_syncword_detection_params* p,
                                             gr4pm_syncword_detection** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    if (p->min_freq_bin > p->max_freq_bin) { // hpp:145-147
        set_error("min_freq_bin is greater than max_freq_bin");
        return GR4PM_ERR_INVALID;
    }
    if (!p->rrc_taps || !p->syncword || !p->constellation || p->n_syncword == 0 ||
        p->n_rrc_taps == 0 || p->samples_per_symbol == 0 || p->n_channels == 0) {
        set_error("missing setting");
        return GR4PM_ERR_INVALID;
    }
    const size_t L = (p->n_syncword - 1) * p->samples_per_symbol + p->n_rrc_taps; // hpp:148-149
    if (L > p->fft_size) { // hpp:150-152
        set_error("fft_size too small");
        return GR4PM_ERR_INVALID;
    }
    for (size_t j = 0; j < p->n_syncword; ++j)
        if (p->syncword[j] >= p->n_constellation) {
            set_error("syncword symbol outside constellation");
            return GR4PM_ERR_INVALID;
        }
    const bool pow2 = p->fft_size >= 256 && p->fft_size <= 8192 && (p->fft_size & (p->fft_size - 1)) == 0;
    if (!pow2) {
        set_error("fft_size %zu not built (supported: powers of two 256..8192; 2048 is the tuned path)",
                  p->fft_size);
        return GR4PM_ERR_UNSUPPORTED;
    }
    const int n_bins = p->max_freq_bin - p->min_freq_bin + 1;
    if (n_bins > kMaxBins || p->time_threshold > 8192 || p->time_threshold == 0) {
        set_error("n_bins %d (max %d) or time_threshold %llu (1..8192) not built", n_bins, kMaxBins,
                  static_cast<unsigned long long>(p->time_threshold));
        return GR4PM_ERR_UNSUPPORTED;
    }
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_syncword_detection;
    if (!h) return GR4PM_ERR_NOMEM;
    h->fft_size = p->fft_size;
    h->generic = p->fft_size != static_cast<size_t>(kFftN);
    h->log2n = 0;
    while ((size_t{ 1 } << h->log2n) < p->fft_size) ++h->log2n;
    h->sps = p->samples_per_symbol;
    h->n_channels = p->n_channels;
    h->min_bin = p->min_freq_bin;
    h->max_bin = p->max_freq_bin;
    h->n_bins = n_bins;
    h->T = p->time_threshold;
    h->power_threshold = p->power_threshold;
    h->stream = static_cast<hipStream_t>(p->stream);
    h->L = L;
    h->S = p->fft_size - L + 1; // hpp:236
    h->hist = 2 * h->T + 1;     // hpp:194
    h->max_items = std::max(p->max_items, p->fft_size);

    // start(), hpp:154-189: shaped syncword, self correlation, frequency-shifted templates
    using c64 = std::complex<float>;
    std::vector<c64> sw(L);
    for (size_t j = 0; j < p->n_syncword; ++j) {
        const gr4pm_c64 cc = p->constellation[p->syncword[j]];
        for (size_t k = 0; k < p->n_rrc_taps; ++k) {
            const float t = p->rrc_taps[k];
            sw[j * h->sps + k] += c64(cc.re * t, cc.im * t); // complex * float, hpp:157-158
        }
    }
    float self_corr = 0.0f;
    for (auto x : sw) self_corr += x.real() * x.real() + x.imag() * x.imag(); // hpp:161-164
    h->self_corr = self_corr;
    std::vector<float4> tmpl(h->generic ? 1 : static_cast<size_t>(n_bins) * 1024);
    std::vector<float4> tmplp(tmpl.size()), tmpl64(tmpl.size());
    std::vector<cf> g_tmpl(h->generic ? static_cast<size_t>(n_bins) * p->fft_size : 0);
    std::vector<cf> td(static_cast<size_t>(n_bins) * L);
    for (int b = 0; b < n_bins; ++b) {
        const int freq_bin = p->min_freq_bin + b;
        double phase = 0.0;
        const double phase_incr = static_cast<double>(freq_bin) * M_PI / static_cast<double>(L);
        std::vector<std::complex<double>> a(p->fft_size);
        for (size_t i = 0; i < L; ++i) {
            const float c = static_cast<float>(std::cos(phase)), s = static_cast<float>(std::sin(phase));
            const float xr = sw[i].real(), xi = sw[i].imag();
            a[i] = { static_cast<double>(xr * c - xi * s), static_cast<double>(xr * s + xi * c) };
            td[static_cast<size_t>(b) * L + i] = mk(xr * c - xi * s, -(xr * s + xi * c));
            phase += phase_incr;
            if (phase >= M_PI) { // hpp:177-181, including the `< pi` quirk
                phase -= 2.0 * M_PI;
            } else if (phase < M_PI) {
                phase += 2.0 * M_PI;
            }
        }
        fft_double(a);
        if (h->generic) { // conj (hpp:185-187), natural order
            for (size_t k = 0; k < p->fft_size; ++k)
                g_tmpl[static_cast<size_t>(b) * p->fft_size + k] =
                    mk(static_cast<float>(a[k].real()), static_cast<float>(-a[k].imag()));
            continue;
        }
        // conj (hpp:185-187), rounded to float32, laid out in FFT-1's output distribution
        for (int lane = 0; lane < 64; ++lane)
            for (int jp = 0; jp < 16; ++jp) {
                const auto t0 = std::conj(a[fft1_out_index(lane, 2 * jp)]);
                const auto t1 = std::conj(a[fft1_out_index(lane, 2 * jp + 1)]);
                tmpl[static_cast<size_t>(b) * 1024 + jp * 64 + lane] =
                    make_float4(static_cast<float>(t0.real()), static_cast<float>(t0.imag()),
                                static_cast<float>(t1.real()), static_cast<float>(t1.imag()));
            }
        for (int lane = 0; lane < 64; ++lane) // and in the one-exchange schedule's: index lane + 64 j
            for (int u = 0; u < 16; ++u) {
                const auto t0 = std::conj(a[w64_index(lane, 2 * u)]);
                const auto t1 = std::conj(a[w64_index(lane, 2 * u + 1)]);
                tmpl64[static_cast<size_t>(b) * 1024 + u * 64 + lane] =
                    make_float4(static_cast<float>(t0.real()), static_cast<float>(t0.imag()),
                                static_cast<float>(t1.real()), static_cast<float>(t1.imag()));
            }
        for (int L = 0; L < 128; ++L) // the same values in the pair schedule's lane order
            for (int jp = 0; jp < 8; ++jp) {
                const auto t0 = std::conj(a[fft1p_out_index(L, 2 * jp)]);
                const auto t1 = std::conj(a[fft1p_out_index(L, 2 * jp + 1)]);
                tmplp[static_cast<size_t>(b) * 1024 + jp * 128 + L] =
                    make_float4(static_cast<float>(t0.real()), static_cast<float>(t0.imag()),
                                static_cast<float>(t1.real()), static_cast<float>(t1.imag()));
            }
    }
    std::vector<cf> tw(kTw1aItems + kTw1bItems + kTwAItems + kTwBItems);
    build_twiddle_tables(
        [](int k) {
            const double ang = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        },
        tw.data(), tw.data() + kTw1aItems, tw.data() + kTw1aItems + kTw1bItems,
        tw.data() + kTw1aItems + kTw1bItems + kTwAItems);
    std::vector<cf> twp(kTw1pItems + kTwApItems + kTwBItems);
    build_pair_twiddle_tables(
        [](int k) {
            const double ang = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        },
        twp.data(), twp.data() + kTw1pItems);
    for (int i = 0; i < kTwBItems; ++i) twp[kTw1pItems + kTwApItems + i] = tw[kTw1aItems + kTw1bItems + kTwAItems + i];
    std::vector<float4> tT64(kW64TwFloat4);
    std::vector<cf> cc64(64);
    build_w64_tables(
        [](int k) {
            const double ang = -2.0 * M_PI * k / kFftN;
            return mk(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        },
        tT64.data(), cc64.data());
    {
        // which correlator kernel runs: "w64" (default: one LDS exchange per transform), "wave" (round 1: two
        // exchanges; bit-identical to "pair", two waves per block)
        const char* e = getenv("GR4PM_CORRELATOR");
        const std::string k = e ? e : "w64";
        h->use_pair = !h->generic && k == "pair";
        h->corr_kind = k == "pair" ? 2 : k == "wave" ? 1 : 0;
        // the bit-identical variants (-1 default, 0 = round 2's bin loop, 65536 = planar loop with LDS-DMA templates,
        // GR4PM_W64_ONE=1) select silently; every other value is a timing-only ablation with wrong powers and says so
        // (the library as build() makes it holds the bit-identical forms only: experiment_env(.., true) returns nullptr
        // there and says that the switch was ignored; `make EXPERIMENTS=1` -- tools/build_variant.sh -- has the rest)
        const char* v = getenv("GR4PM_W64_VARIANT");
        h->w64_variant = v ? atoi(v) : -1;
        if (v && h->w64_variant != -1 && h->w64_variant != 0 && h->w64_variant != 65536 && h->w64_variant != 131072 &&
            h->w64_variant != 262144 &&
            !experiment_env("GR4PM_W64_VARIANT", true))
            h->w64_variant = -1;
        if (const char* cv = getenv("GR4PM_C4096_VARIANT")) {
            const int c = atoi(cv);
            if ((c >= 0 && c <= 7) || c == 17) h->c4096_variant = c;
            else if (c == 9 && experiment_env("GR4PM_C4096_VARIANT", true)) h->c4096_variant = 9;
            else fprintf(stderr, "[gr4pm] GR4PM_C4096_VARIANT=%s is not one of 0 .. 7: using the default (1)\n", cv);
        }
        const char* one = getenv("GR4PM_W64_ONE");
        h->w64_one = one ? atoi(one) : -1;
        if (h->w64_one > 1 && !experiment_env("GR4PM_W64_ONE", true)) h->w64_one = 1;
        if (const char* b = getenv("GR4PM_W64_BLOCKS_PER_WAVE")) h->w64_blocks_per_wave = static_cast<uint32_t>(std::max(0, atoi(b)));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            h->n_cus = prop.multiProcessorCount;
        // GR4PM_W64_CUS: persistent workgroups of the correlator (default: one per CU); fewer leave whole CUs
        // to the kernels of the other pipeline stages
        if (const char* c = getenv("GR4PM_W64_CUS")) h->n_cus = std::max(1, std::min(h->n_cus, atoi(c)));
    }

    h->xc = static_cast<uint32_t>(round_up(h->hist + h->S + 2, 64));
    h->zc = static_cast<uint32_t>(round_up(2 * h->T + 2, 64));
    h->noise_off = static_cast<uint32_t>(h->zc + round_up(h->max_items, 64) + 64);
    h->z_stride = h->noise_off + round_up(h->max_items / h->S + 4, 64);
    const size_t max_cnt = h->max_items + h->T;
    h->bm_stride = round_up(max_cnt, kCandTile) / 64 + 16;
    h->max_tiles = static_cast<uint32_t>((max_cnt + kTileW - 1) / kTileW);
    h->table_stride = static_cast<size_t>(h->max_tiles) * (h->T + 1);
    h->det_cap = static_cast<uint32_t>(h->max_items / (h->T + 1) + 16);
    h->rec_cap = h->det_cap;
    gr4pm_status s = GR4PM_OK;
    auto ok = [&](gr4pm_status r) {
        if (s == GR4PM_OK) s = r;
    };
    ok(h->tmpl.alloc(tmpl.size()));
    ok(h->tw.alloc(tw.size()));
    ok(h->tmplp.alloc(tmplp.size()));
    ok(h->fault.alloc(1));
    if (h->fault.p) *h->fault.p = 0;
    if (const char* e = getenv("GR4PM_TEST_SPIN_LIMIT")) h->spin_limit = atoi(e); // tests force a timeout with 0
    h->lds_candidates = getenv("GR4PM_CANDIDATES_LDS") != nullptr;
    h->separate_median = getenv("GR4PM_SD_SEPARATE_MEDIAN") != nullptr;
    h->w64_no_prune = getenv("GR4PM_W64_NO_PRUNE") != nullptr;
    ok(h->tmpl64.alloc(tmpl64.size()));
    ok(h->tT64.alloc(tT64.size()));
    ok(h->td.alloc(td.size()));
    ok(h->cc64.alloc(cc64.size()));
    ok(h->twp.alloc(twp.size()));
    std::vector<cf> g_tw(h->generic ? p->fft_size / 2 : 0);
    for (size_t k = 0; k < g_tw.size(); ++k) {
        const double ang = -2.0 * M_PI * static_cast<double>(k) / static_cast<double>(p->fft_size);
        g_tw[k] = mk(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
    }
    std::vector<cf> tw4k(16 * 256 + 256);
    if (h->generic && p->fft_size == static_cast<size_t>(kN4k)) {
        build_4096_tables(
            [](int k) {
                const double ang = -2.0 * M_PI * k / kN4k;
                return mk(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
            },
            tw4k.data(), tw4k.data() + 16 * 256);
        ok(h->tw4k.alloc(tw4k.size()));
        const char* e = getenv("GR4PM_CORRELATOR");
        h->force_radix2 = e && std::string(e) == "radix2";
    }
    if (h->generic) {
        ok(h->g_tmpl.alloc(g_tmpl.size()));
        ok(h->g_tw.alloc(g_tw.size()));
        const int lds_bytes = static_cast<int>(2 * p->fft_size * sizeof(cf)) + 2048;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_correlate_generic),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_tags_generic),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
            set_error("cannot reserve %d bytes of LDS for fft_size %zu", lds_bytes, p->fft_size);
            ok(GR4PM_ERR_HIP);
        }
    }
    for (int i = 0; i < kCarry; ++i) ok(h->carry[i].alloc(static_cast<size_t>(h->xc) * h->n_channels));
    for (int i = 0; i < kSets; ++i) ok(h->z[i].alloc(h->z_stride * h->n_channels));
    for (int i = 0; i < kSets; ++i) {
        ok(h->bitmap[i].alloc(h->bm_stride * h->n_channels));
        ok(h->passmap[i].alloc(2 * h->bm_stride * h->n_channels)); // (pass word, defer word) pairs
        ok(h->table[i].alloc(h->table_stride * h->n_channels));
    }
    ok(h->entry.alloc(static_cast<size_t>(h->max_tiles) * h->n_channels));
    h->max_groups = (h->max_tiles + kGroup - 1) / kGroup;
    h->gtable_stride = static_cast<size_t>(h->max_groups) * (h->T + 1);
    for (int i = 0; i < kSets; ++i) ok(h->gtable[i].alloc(h->gtable_stride * h->n_channels));
    ok(h->gentry.alloc(static_cast<size_t>(h->max_groups) * h->n_channels));
    h->super_level = getenv("GR4PM_SD_NO_SUPER") == nullptr;
    h->sgtable_stride = static_cast<size_t>((h->max_groups + kSuper - 1) / kSuper) * (h->T + 1);
    for (int i = 0; i < kSets; ++i) ok(h->sgtable[i].alloc(h->sgtable_stride * h->n_channels));
    ok(h->st.alloc(h->n_channels));
    ok(h->det.alloc(static_cast<size_t>(h->det_cap) * h->n_channels));
    h->visit_cap = static_cast<uint32_t>((h->max_items + h->T) / (h->T + 1) + h->max_tiles + 16);
    ok(h->visit.alloc(static_cast<size_t>(h->visit_cap) * h->n_channels));
    ok(h->deferred.alloc(static_cast<size_t>(h->visit_cap) * h->n_channels));
    ok(h->st_host.alloc(h->n_channels));
    ok(h->rec_host.alloc(static_cast<size_t>(h->rec_cap) * h->n_channels));
    if (s == GR4PM_OK && h->generic) s = h->g_tmpl.upload(g_tmpl.data(), g_tmpl.size(), h->stream);
    if (s == GR4PM_OK && h->generic) s = h->g_tw.upload(g_tw.data(), g_tw.size(), h->stream);
    if (s == GR4PM_OK && h->tw4k.p) s = h->tw4k.upload(tw4k.data(), tw4k.size(), h->stream);
    if (s == GR4PM_OK) s = h->tmpl.upload(tmpl.data(), tmpl.size(), h->stream);
    if (s == GR4PM_OK) s = h->tw.upload(tw.data(), tw.size(), h->stream);
    if (s == GR4PM_OK) s = h->tmplp.upload(tmplp.data(), tmplp.size(), h->stream);
    if (s == GR4PM_OK && !h->generic) s = h->tmpl64.upload(tmpl64.data(), tmpl64.size(), h->stream);
    if (s == GR4PM_OK) s = h->tT64.upload(tT64.data(), tT64.size(), h->stream);
    if (s == GR4PM_OK) s = h->td.upload(td.data(), td.size(), h->stream);
    if (s == GR4PM_OK) s = h->cc64.upload(cc64.data(), cc64.size(), h->stream);
    if (s == GR4PM_OK) s = h->twp.upload(twp.data(), twp.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s == GR4PM_OK) s = sd_reset(h);
    if (s == GR4PM_OK) s = ensure_ahead_streams(h); // now, not at the first announcement in the middle of a stream
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

void gr4pm_syncword_detection_destroy(gr4pm_syncword_detection* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID

gr4pm_status gr4pm_syncword_detection_reset(gr4pm_syncword_detection* h)
try {
    return h ? sd_reset(h) : GR4PM_ERR_INVALID;
}
GR4PM_ABI_CATCH
size_t gr4pm_syncword_detection_syncword_samples_size(const gr4pm_syncword_detection* h) { return h->L; }
float gr4pm_syncword_detection_self_corr(const gr4pm_syncword_detection* h) { return h->self_corr; }
uint64_t gr4pm_syncword_detection_items_consumed(const gr4pm_syncword_detection* h)
try {
    return h->items_consumed;
}
GR4PM_ABI_CATCH_RET(0)

void gr4pm_syncword_detection_scan_counts(const gr4pm_syncword_detection* h, size_t channel, uint64_t* visited,
                                          uint64_t* tested_from_memory)
try {
    // (the host copy of the channel's counters, written by the last process() call before they were reset)
    if (!h || channel >= h->n_channels) return;
    if (visited) *visited = h->st_host.p[channel].vis_cnt;
    if (tested_from_memory) *tested_from_memory = h->last_fused ? h->st_host.p[channel].def_cnt : h->st_host.p[channel].vis_cnt;
}
GR4PM_ABI_CATCH_VOID

gr4pm_status gr4pm_syncword_detection_correlate_only(gr4pm_syncword_detection* h, const gr4pm_c64* in,
                                                     size_t in_stride, size_t n_in)
try {
    if (!h || !in) return GR4PM_ERR_INVALID;
    if (n_in < h->fft_size) return GR4PM_INSUFFICIENT_INPUT_ITEMS;
    if (n_in > h->max_items) {
        set_error("n_in %zu exceeds max_items %zu", n_in, h->max_items);
        return GR4PM_ERR_INVALID;
    }
    const uint32_t n_blocks = static_cast<uint32_t>((n_in - h->fft_size) / h->S + 1);
    return launch_correlate(h, h->stream, in, in_stride, n_blocks, h->z[h->cur].p + h->zc);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_detection_announce(gr4pm_syncword_detection* h, const gr4pm_c64* in,
                                               size_t in_stride, size_t n_in)
try {
    if (!h || !in) return GR4PM_ERR_INVALID;
    GR4PM_TRY(ensure_ahead_streams(h));
    // one launched front is consumed by the next call before a new one is launched
    if (h->launched.size() + h->announced.size() < static_cast<size_t>(kAhead) + 1) {
        const int e = h->ev_next;
        h->ev_next = (h->ev_next + 1) % (kAhead + 2);
        if (!h->ev_announce[e]) GR4PM_HIP_TRY(hipEventCreateWithFlags(&h->ev_announce[e], hipEventDisableTiming));
        GR4PM_HIP_TRY(hipEventRecord(h->ev_announce[e], h->stream));
        h->announced.push_back({ in, in_stride, n_in, 0, 0, e });
    }
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_detection_hint_next(gr4pm_syncword_detection* h, const gr4pm_c64* in_next,
                                                size_t in_stride, size_t n_next)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->announced.clear();
    if (!in_next) return GR4PM_OK;
    return gr4pm_syncword_detection_announce(h, in_next, in_stride, n_next);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_detection_process(gr4pm_syncword_detection* h, const gr4pm_c64* in,
                                              size_t in_stride, size_t n_in, gr4pm_c64* out,
                                              size_t out_stride, size_t* n_done, gr4pm_tag* tags,
                                              size_t tags_cap, size_t* n_tags)
try {
    if (!h || !in || !n_done) return GR4PM_ERR_INVALID;
    *n_done = 0;
    if (n_tags)
        for (size_t c = 0; c < h->n_channels; ++c) n_tags[c] = 0;
    if (n_in < h->fft_size) return GR4PM_INSUFFICIENT_INPUT_ITEMS; // hpp:215-227
    if (n_in > h->max_items) {
        set_error("n_in %zu exceeds max_items %zu", n_in, h->max_items);
        return GR4PM_ERR_INVALID;
    }
#ifdef GR4PM_TIMING
    static double t_acc[4] = { 0, 0, 0, 0 };
    static int t_calls = 0;
    const auto tp0 = std::chrono::steady_clock::now();
#endif
    const uint32_t n_blocks = static_cast<uint32_t>((n_in - h->fft_size) / h->S + 1); // hpp:238
    const size_t J = static_cast<size_t>(n_blocks) * h->S;
    const uint64_t E0 = h->items_consumed, E1 = E0 + J;
    const uint32_t T = static_cast<uint32_t>(h->T);
    const int cur = (h->cur + 1) % kSets; // the buffer set of this call
    const unsigned nch = static_cast<unsigned>(h->n_channels);
    hipStream_t s = h->stream;
    float* zcur = h->z[cur].p;          // [zc carry][J items] per channel
    const cf* carry = h->carry[h->ci].p; // last xc items before E0

    // the front part of this call (z carry, correlation powers, candidate bitmap, tile and group
    // tables) may already be there, or on its way, from the look-ahead of an earlier call; fronts
    // that do not match this call are waited for and dropped
    bool have_front = false;
    if (!h->launched.empty()) {
        const auto& a = h->launched.front();
        if (a.in == in && a.stride == in_stride && a.n == n_in && a.E0 == E0 && a.set == cur) {
            GR4PM_HIP_TRY(hipStreamWaitEvent(s, h->ev_front[cur], 0));
            h->launched.pop_front();
            have_front = true;
        } else {
            for (const auto& l : h->launched) GR4PM_HIP_TRY(hipStreamWaitEvent(s, h->ev_front[l.set], 0));
            h->launched.clear();
        }
    }
    if (!have_front) GR4PM_TRY(launch_front(h, s, cur, h->ci, in, in_stride, n_in, E0, h->last_done));
    // fronts launched below read this call's z / sample carry: behind this call's front on the
    // look-ahead stream already, or ordered by an event when the front was launched here
    const bool order_by_event = !have_front && !h->announced.empty();
    if (order_by_event) GR4PM_HIP_TRY(hipEventRecord(h->ev_zcarry, s));

    // the serial part of the detector over candidate range [A0, A1)
    const uint64_t A0 = E0 > T ? E0 - T : 0, A1 = E1 > T ? E1 - T : 0;
    const uint32_t cnt = static_cast<uint32_t>(A1 - A0);
    // local position 0 <-> absolute A0 <-> zcur[zc + (A0 - E0)]
    const float* zloc = zcur + h->zc - static_cast<ptrdiff_t>(E0 - A0);
    if (cnt > 0) {
        const uint32_t n_tiles = (cnt + kTileW - 1) / kTileW;
        const uint32_t n_groups = (n_tiles + kGroup - 1) / kGroup;
        hipLaunchKernelGGL(k_scan_entries, dim3(nch), dim3(256), 0, s, h->st.p, static_cast<unsigned long long>(A0),
                           cnt, T, n_tiles, h->table[cur].p, h->table_stride, h->gtable[cur].p, h->gtable_stride,
                           h->gentry.p, n_groups, h->entry.p,
                           h->super_level && n_groups > kSuper ? h->sgtable[cur].p : nullptr, h->sgtable_stride);
        hipLaunchKernelGGL(k_tile_visit, dim3((n_tiles + kVisitWaves - 1) / kVisitWaves, nch), dim3(64 * kVisitWaves), 0, s,
                           h->bitmap[cur].p, h->bm_stride, cnt, T,
                           n_tiles, h->entry.p, h->st.p, h->visit.p, h->visit_cap);
        // at most one visited candidate per T + 1 items (+ one per tile): the grids stride over the real count
        const uint32_t n_vis = cnt / (T + 1) + n_tiles + 1;
        // 16 384 workgroups over ALL channels (round 6: that many per channel were 358 000 workgroups with nothing to do
        // per step of config 2, 105 us)
        const uint32_t median_grid = std::max<uint32_t>(64u, 16384u / nch);
        h->last_fused = h->fused_median[cur];
        if (h->fused_median[cur]) {
            // the front's candidate kernel has tested (nearly) every candidate: look the visited ones up; the untested ones
            // (constant input) onto a second list and through the test from memory (the grid strides over their real
            // count and leaves at once when there is none)
            hipLaunchKernelGGL(k_resolve_visited, dim3((n_vis + 256 * kResolveWaves - 1) / (256 * kResolveWaves), nch),
                               dim3(64 * kResolveWaves), 0, s,
                               static_cast<unsigned long long>(A0), h->st.p, h->visit.p, h->visit_cap, h->passmap[cur].p,
                               h->bm_stride, h->det.p, h->det_cap, h->deferred.p);
            hipLaunchKernelGGL(k_median_tests<true>, dim3(std::min<uint32_t>(n_vis, median_grid), nch), dim3(64), 0, s, zloc,
                               h->z_stride, static_cast<unsigned long long>(A0), T, h->power_threshold, h->st.p,
                               h->deferred.p, h->visit_cap, h->det.p, h->det_cap);
        } else {
            hipLaunchKernelGGL(k_median_tests<false>, dim3(std::min<uint32_t>(n_vis, median_grid), nch), dim3(64), 0, s, zloc,
                               h->z_stride, static_cast<unsigned long long>(A0), T, h->power_threshold, h->st.p, h->visit.p,
                               h->visit_cap, h->det.p, h->det_cap);
        }
    }
    // tags leaving in this call
    if (h->generic && h->fft_size == static_cast<size_t>(kN4k) && !h->force_radix2 && h->hist <= h->S) {
        // k_correlate_4096 has left every block's noise power behind the powers: the tags come from k_tags<false>
        // (correlation at the detection's lag from its definition), as behind k_correlate_w64
        hipLaunchKernelGGL((k_tags<false, 9>), dim3(std::min<uint32_t>(h->det_cap, 4096u) / kTagWaves<false>, nch),
                           dim3(64 * kTagWaves<false>), 0, s,
                           reinterpret_cast<const cf*>(in), in_stride, carry, static_cast<size_t>(h->xc), h->xc,
                           static_cast<unsigned long long>(E0), static_cast<unsigned long long>(E1),
                           static_cast<uint32_t>(h->hist), static_cast<uint32_t>(h->S), h->n_bins, h->tmpl.p, h->tw.p, h->tw.p,
                           h->tw.p, h->tw.p, h->td.p, static_cast<uint32_t>(h->L), zcur + h->zc, h->z_stride, h->st.p,
                           h->det.p, h->det_cap, h->rec_host.p, h->rec_cap, h->noise_off - h->zc,
                           static_cast<uint32_t>(kN4k));
    } else if (h->generic) {
        const uint32_t N = static_cast<uint32_t>(h->fft_size);
        hipLaunchKernelGGL(k_tags_generic, dim3(std::min<uint32_t>(h->det_cap, 1024u), nch), dim3(256),
                           2 * N * sizeof(cf), s,
                           reinterpret_cast<const cf*>(in), in_stride, carry, static_cast<size_t>(h->xc), h->xc,
                           static_cast<unsigned long long>(E0), static_cast<unsigned long long>(E1),
                           static_cast<uint32_t>(h->hist), static_cast<uint32_t>(h->S), N, h->log2n, h->n_bins,
                           h->g_tmpl.p, h->g_tw.p, zcur + h->zc, h->z_stride, h->st.p, h->det.p, h->det_cap,
                           h->rec_host.p, h->rec_cap);
    } else {
        // 2048 waves over ALL channels: what the chip holds of the 9-bin form at once (two per SIMD); more only queue
        // (measured: 4096 and 8192 per channel are slower on one channel, 2048 per channel took 179 us on 64 channels, 102
        // with 256)
        static const uint32_t tags_waves_env = getenv("GR4PM_TAGS_WAVES") ? static_cast<uint32_t>(std::max(8, atoi(getenv("GR4PM_TAGS_WAVES")))) : 0u;
        const uint32_t tags_waves = tags_waves_env ? tags_waves_env : std::max<uint32_t>(8u, (2048u / nch) & ~7u);
        auto launch_tags = [&](auto kernel, uint32_t waves_per_wg) {
            hipLaunchKernelGGL(kernel, dim3(std::max<uint32_t>(1u, std::min<uint32_t>(h->det_cap, tags_waves) / waves_per_wg), nch),
                               dim3(64 * waves_per_wg), 0, s,
                               reinterpret_cast<const cf*>(in), in_stride, carry, static_cast<size_t>(h->xc), h->xc,
                               static_cast<unsigned long long>(E0), static_cast<unsigned long long>(E1),
                               static_cast<uint32_t>(h->hist), static_cast<uint32_t>(h->S), h->n_bins, h->tmpl.p, h->tw.p,
                               h->tw.p + kTw1aItems, h->tw.p + kTw1aItems + kTw1bItems,
                               h->tw.p + kTw1aItems + kTw1bItems + kTwAItems, h->td.p, static_cast<uint32_t>(h->L),
                               zcur + h->zc, h->z_stride, h->st.p, h->det.p, h->det_cap, h->rec_host.p, h->rec_cap,
                               h->noise_off - h->zc, static_cast<uint32_t>(kFftN));
        };
        // k_correlate_w64 leaves every block's noise power behind the powers; the round-1 correlators do not.  Only ONE
        // block of the call before is carried (k_update_zcarry): enough while a detection that leaves in this call
        // cannot lie more than one block before E0, i.e. hist = 2T + 1 <= S.  A longer history (T > 875 at the
        // default S = 1752) puts detections two or more blocks back; their block is then transformed again from the
        // sample carry (xc >= hist + S + 2 covers it).
        if (h->corr_kind == 0 && !h->use_pair && h->hist <= h->S) {
            if (h->n_bins <= 4) launch_tags(k_tags<false, 4>, kTagWaves<false>); // (fewer registers: twice the waves)
            else launch_tags(k_tags<false, 9>, kTagWaves<false>);
        } else {
            launch_tags(k_tags<true, 9>, kTagWaves<true>);
        }
    }
    hipLaunchKernelGGL(k_compact_pending, dim3(nch), dim3(kCompactThreads), 0, s, h->st.p, h->st_host.p, h->det.p,
                       h->det_cap, static_cast<unsigned long long>(E1), static_cast<uint32_t>(h->hist),
                       static_cast<int>(nch));
    if (out) {
        const unsigned gx = static_cast<unsigned>(std::min<size_t>((J + 255) / 256, 4096));
        hipLaunchKernelGGL(k_delay_copy, dim3(gx, nch), dim3(256), 0, s, reinterpret_cast<const cf*>(in),
                           in_stride, carry, static_cast<size_t>(h->xc), h->xc,
                           static_cast<uint32_t>(h->hist), J, reinterpret_cast<cf*>(out), out_stride);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    // fronts of the announced calls, queued after this call's own kernels (they are the critical
    // path).  A front writes the buffer set of a call that has completed (every call ends with a
    // host-side wait) and reads the set / sample carry of the call before it, which is ahead of it
    // on the same stream.
    {
        uint64_t e0 = E1;
        size_t j_prev = J;
        int set = cur, c = h->ci;
        for (const auto& l : h->launched) {
            const size_t nb = (l.n - h->fft_size) / h->S + 1;
            e0 = l.E0 + nb * h->S;
            j_prev = nb * h->S;
            set = l.set;
            c = (c + 1) % kCarry;
        }
        bool first = true;
        while (!h->announced.empty() && h->launched.size() < static_cast<size_t>(kAhead)) {
            auto a = h->announced.front();
            h->announced.pop_front();
            if (a.n < h->fft_size || a.n > h->max_items) { // not a call this handle can run: no look-ahead
                h->announced.clear();
                break;
            }
            if (first && order_by_event) GR4PM_HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_zcarry, 0));
            first = false;
            set = (set + 1) % kSets;
            c = (c + 1) % kCarry;
            a.E0 = e0;
            a.set = set;
            // this front's correlator overwrites the z buffer whose tail the z carry of the front two
            // calls before it reads on the other look-ahead stream (set + 1): long done in practice
            // (a whole correlator launch lies in between), ordered by its event all the same
            if (h->front_recorded[(set + 1) % kSets])
                GR4PM_HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_front[(set + 1) % kSets], 0));
            GR4PM_HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_announce[a.ev], 0)); // the producer of a.in has run
            GR4PM_TRY(launch_front(h, h->stream2, set, c, a.in, a.stride, a.n, e0, j_prev));
            h->launched.push_back(a);
            const size_t nb = (a.n - h->fft_size) / h->S + 1;
            e0 += nb * h->S;
            j_prev = nb * h->S;
        }
    }
#ifdef GR4PM_TIMING
    const auto tp1 = std::chrono::steady_clock::now();
#endif
    GR4PM_HIP_TRY(hipStreamSynchronize(s));
#ifdef GR4PM_TIMING
    const auto tp2 = std::chrono::steady_clock::now();
#endif
    if (*h->fault.p != 0) {
        set_error("correlator template hand-off timed out (fault word %u): the correlation powers of this call are "
                  "not valid", *h->fault.p);
        *h->fault.p = 0;
        return GR4PM_ERR_HIP;
    }

    // the raw records were written by k_tags straight into pinned host memory (a handful of
    // 48-byte records per call: no second copy, no second synchronisation); the tag arithmetic
    // is finished on the host below
    bool overflow = false, any = false;
    for (unsigned c = 0; c < nch; ++c) {
        overflow |= h->st_host.p[c].overflow != 0;
        any |= h->st_host.p[c].rec_cnt != 0;
    }
#ifdef GR4PM_TIMING
    const auto tp3 = std::chrono::steady_clock::now();
#endif
    h->items_consumed = E1;
    h->cur = cur;
    h->ci = (h->ci + 1) % kCarry;
    h->last_done = J;
    *n_done = J;
    if (overflow) {
        set_error("detection list overflow");
        return GR4PM_ERR_OVERFLOW;
    }
    gr4pm_status ret = GR4PM_OK;
    for (unsigned c = 0; c < nch && any; ++c) {
        const uint32_t n = std::min(h->st_host.p[c].rec_cnt, h->rec_cap);
        RawTag* r = h->rec_host.p + static_cast<size_t>(c) * h->rec_cap;
        std::sort(r, r + n, [](const RawTag& a, const RawTag& b) { return a.pos < b.pos; });
        if (n_tags) n_tags[c] = n;
        for (uint32_t i = 0; i < n; ++i) {
            if (!tags || i >= tags_cap) {
                ret = GR4PM_ERR_OVERFLOW;
                break;
            }
            finish_tag(h, r[i], r[i].pos + h->hist - E0, &tags[static_cast<size_t>(c) * tags_cap + i]);
        }
    }
    if (ret == GR4PM_ERR_OVERFLOW) set_error("tags_cap too small");
#ifdef GR4PM_TIMING
    {
        const auto tp4 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        t_acc[0] += us(tp0, tp1);
        t_acc[1] += us(tp1, tp2);
        t_acc[2] += us(tp2, tp3);
        t_acc[3] += us(tp3, tp4);
        if (++t_calls % 8 == 0) {
            fprintf(stderr, "[gr4pm timing] launch %.0f us, sync %.0f us, records %.0f us, finish %.0f us (mean of the last 8)\n",
                    t_acc[0] / 8, t_acc[1] / 8, t_acc[2] / 8, t_acc[3] / 8);
            t_acc[0] = t_acc[1] = t_acc[2] = t_acc[3] = 0;
        }
    }
#endif
    return ret;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_detection_last_zpow(gr4pm_syncword_detection* h, float* zpow, size_t stride)
try {
    if (!h || !zpow) return GR4PM_ERR_INVALID;
    for (size_t c = 0; c < h->n_channels; ++c) {
        GR4PM_HIP_TRY(hipMemcpyAsync(zpow + c * stride, h->z[h->cur].p + c * h->z_stride + h->zc,
                                     h->last_done * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"
