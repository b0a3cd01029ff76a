// correlate_w64_one.hpp -- k_correlate_w64_one: the ONE-bin form of k_correlate_w64 (correlate_w64.hpp), three waves
// per SIMD instead of two.  Replaces syncword_detection.hpp:238-252,300-313 for min_freq_bin == max_freq_bin (the
// reference's own benchmark publishes that row: benchmarks/results.md).
//
// With one frequency bin the block costs two transforms and moves 13.4 bytes per sample through HBM (8 x 2048 / 1752
// read, 4 written): at two waves per SIMD the vector ALU is 58 % busy and a quarter of the wave cycles wait for memory
// (profiles/r3_k_correlate_1bin_pmc.json) -- neither roof.  What two waves per SIMD need the registers for does not
// exist here: no spectrum kept across bins (the product X .* T happens in place), no running maximum (the power of
// the only bin is the result), the template (16 KiB) stays in LDS for the whole launch.  What is left fits 168 VGPRs,
// and the LDS fits twelve waves because the exchange goes through the buffer in two halves:
//
//   pass A output (32 complex per lane) -> 32 stores of the RE parts (ds_write_addtid_b32, rows of 64 + 4 dwords)
//   -> the lane's sixteen 16-byte row reads of the re image (they stay in 64 registers)
//   -> 32 stores of the IM parts over the same rows (the LDS queue of a wave is in order: the reads above have been
//      served) -> per group of four points: two row reads of the im image + two twiddle reads, then the same
//      arithmetic as w64_mid_dev / w64_mid_dev_p on (re rows from registers, im rows from LDS).
//
// LDS: 16 KiB twiddle table + 16 KiB template + 12 x 8.5 KiB exchange buffers = 134 KiB.  Same instruction count per
// transform as the nine-bin kernel, bit-identical powers, 166 VGPRs, no spills.
//
// MEASURED (MI355X, 2^26 samples, tools/one_bin_check.py): 0.243 ms median / 0.225 min against 0.250 / 0.216 for the
// general kernel at two waves per SIMD -- no gain, so the kernel is OPT-IN (GR4PM_W64_ONE=1) and the general kernel
// stays the one-bin path.  Its own ablations say why: the memory side alone (loads, one FMA per sample, stores) takes
// 0.180 ms (5.0 TB/s for 8-byte-per-lane loads and 4-byte-per-lane stores), the arithmetic alone 0.172 ms with three
// waves per SIMD and 0.178 ms with two: the arithmetic is not waiting for anything a third wave could hide, and the
// two halves overlap to 0.24, not to 0.18, at either occupancy.
#pragma once
#include "correlate_w64.hpp"

namespace gr4pm {
namespace {

constexpr int kW1Waves = 12, kW1Threads = kW1Waves * 64;
constexpr int kW1Row = 68;                      // dwords per row of the half image: 64 + 4 pad
constexpr int kW1BufF4 = 32 * kW1Row / 4;       // 544 float4 = 8704 B per wave
constexpr int kW1TmplF4 = 16 * 64;              // the one template: [u = 16][lane = 64] float4
constexpr int kW1LdsF4 = kW64TwFloat4 + kW1TmplF4 + kW1Waves * kW1BufF4;
static_assert(kW1LdsF4 * 16 <= 160 * 1024, "LDS budget of one workgroup per CU");

// 32 stores of one plane (X = 0: re, 1: im): lane l writes r[k] to row k, column l
#define GR4PM_W1_ST8(k, M)                                                                                            \
    asm volatile("ds_write_addtid_b32 %0 offset:%c8\n\tds_write_addtid_b32 %1 offset:%c9\n\t"                        \
                 "ds_write_addtid_b32 %2 offset:%c10\n\tds_write_addtid_b32 %3 offset:%c11\n\t"                      \
                 "ds_write_addtid_b32 %4 offset:%c12\n\tds_write_addtid_b32 %5 offset:%c13\n\t"                      \
                 "ds_write_addtid_b32 %6 offset:%c14\n\tds_write_addtid_b32 %7 offset:%c15"                          \
                 :                                                                                                    \
                 : "v"(r[(k)].M), "v"(r[(k) + 1].M), "v"(r[(k) + 2].M), "v"(r[(k) + 3].M), "v"(r[(k) + 4].M),       \
                   "v"(r[(k) + 5].M), "v"(r[(k) + 6].M), "v"(r[(k) + 7].M), "i"((k) * kW1Row * 4),                   \
                   "i"(((k) + 1) * kW1Row * 4), "i"(((k) + 2) * kW1Row * 4), "i"(((k) + 3) * kW1Row * 4),            \
                   "i"(((k) + 4) * kW1Row * 4), "i"(((k) + 5) * kW1Row * 4), "i"(((k) + 6) * kW1Row * 4),            \
                   "i"(((k) + 7) * kW1Row * 4)                                                                        \
                 : "memory")
__device__ __forceinline__ void w1_store_re(const cf* r, uint32_t base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base) : "memory"); // (see w64_store: M0 belongs to this asm)
    GR4PM_W1_ST8(0, x);
    GR4PM_W1_ST8(8, x);
    GR4PM_W1_ST8(16, x);
    GR4PM_W1_ST8(24, x);
}
__device__ __forceinline__ void w1_store_im(const cf* r, uint32_t base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base) : "memory");
    GR4PM_W1_ST8(0, y);
    GR4PM_W1_ST8(8, y);
    GR4PM_W1_ST8(16, y);
    GR4PM_W1_ST8(24, y);
}
#undef GR4PM_W1_ST8

// the exchange and the mid stage of one transform.  PLANAR: output as sixteen pc (pairs of points, re / im apart)
// for dft32p, else 32 interleaved cf for dft32.  `between` runs after the im stores have been issued (the caller's
// registers of pass A are dead from there on).
template <bool PLANAR, typename Between>
__device__ __forceinline__ void w1_exchange(const cf* r, uint32_t base, int lane, const float4* row, const float4* tT, cf c,
                                            cf* b, pc* bp, Between between)
{
    w1_store_re(r, base);
    float4 re[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) re[i] = row[i];
    // the reads above are in the wave's LDS queue before the stores below (asm volatile + "memory" on both sides)
    asm volatile("" ::: "memory");
    w1_store_im(r, base);
    between();
    float4 q[2][4]; // im rows g, 8 + g and the twiddles of group g; group g + 1 is in flight while g is consumed
    auto issue = [&](int g) {
        float4* d = q[g & 1];
        d[0] = row[g], d[1] = row[8 + g];
        d[2] = tT[(g * 2 + 0) * 64 + lane], d[3] = tT[(g * 2 + 1) * 64 + lane];
    };
    issue(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) issue(g + 1);
        const float4* s = q[g & 1];
        const float4 a0r = re[g], a1r = re[8 + g];
        if (PLANAR) {
            bp[2 * g] = w64_mid_pair_asm(f2{ a0r.x, a0r.y }, f2{ a1r.x, a1r.y }, f2{ s[0].x, s[0].y }, f2{ s[1].x, s[1].y },
                                         f2{ s[2].x, s[2].y }, f2{ s[3].x, s[3].y }, c);
            bp[2 * g + 1] = w64_mid_pair_asm(f2{ a0r.z, a0r.w }, f2{ a1r.z, a1r.w }, f2{ s[0].z, s[0].w },
                                             f2{ s[1].z, s[1].w }, f2{ s[2].z, s[2].w }, f2{ s[3].z, s[3].w }, c);
            asm volatile("" : "+v"(bp[2 * g].r), "+v"(bp[2 * g].i), "+v"(bp[2 * g + 1].r), "+v"(bp[2 * g + 1].i)::"memory");
        } else {
            w64_mid_pair(f2{ a0r.x, a0r.y }, f2{ a1r.x, a1r.y }, f2{ s[0].x, s[0].y }, f2{ s[1].x, s[1].y },
                         f2{ s[2].x, s[2].y }, f2{ s[3].x, s[3].y }, c, b + 4 * g);
            w64_mid_pair(f2{ a0r.z, a0r.w }, f2{ a1r.z, a1r.w }, f2{ s[0].z, s[0].w }, f2{ s[1].z, s[1].w },
                         f2{ s[2].z, s[2].w }, f2{ s[3].z, s[3].w }, c, b + 4 * g + 2);
            w64_pin4(b + 4 * g);
        }
    }
}

// tmpl: [u = 16][lane = 64] float4 of the one bin (the layout of k_correlate_w64's templates); everything else as there.
// Requires stride_s <= 1793 (the registers whose lags are never stored are not computed).
// ABL (timing only, wrong results): 1 = the memory side alone (loads, one power per sample, stores), 2 = the arithmetic
// alone (no sample loads after the first block, no power stores)
template <int ABL>
__global__ __launch_bounds__(kW1Threads) void k_correlate_w64_one(const cf* __restrict__ in, size_t in_stride,
                                                                  uint32_t n_blocks, uint32_t total, uint32_t stride_s,
                                                                  const float4* __restrict__ tmpl,
                                                                  const float4* __restrict__ tT,
                                                                  const cf* __restrict__ cc, float* __restrict__ zpow,
                                                                  size_t z_stride, uint32_t blocks_per_wave,
                                                                  uint32_t noise_rel)
{
    __shared__ float4 lds4[kW1LdsF4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kW64TwFloat4; i += kW1Threads) lds4[i] = tT[i];
    for (int i = tid; i < kW1TmplF4; i += kW1Threads) lds4[kW64TwFloat4 + i] = tmpl[i];
    __syncthreads(); // the only workgroup-wide synchronisation of the kernel
    const float4* ldsT = lds4;
    const float4* ldsTmpl = lds4 + kW64TwFloat4 + lane;
    float4* xb4 = lds4 + kW64TwFloat4 + kW1TmplF4 + wave * kW1BufF4;
    const uint32_t base = __builtin_amdgcn_readfirstlane(
        static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)xb4)));
    const float4* row = xb4 + (lane & 31) * (kW1Row / 4);
    const cf c = cc[lane];
    const uint32_t n_waves = blocks_per_wave ? kW1Waves : gridDim.x * kW1Waves;
    // (blocks_per_wave != 0: an even split of the items over the grid, as in k_correlate_w64)
    const uint32_t wg_begin = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * total / gridDim.x);
    const uint32_t wg_end = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * total / gridDim.x);
    uint32_t item = blocks_per_wave ? wg_begin + wave : blockIdx.x * kW1Waves + wave;
    const uint32_t item_end = blocks_per_wave ? wg_end : total;
    if (item >= item_end) return;

    const bool one_channel = total <= n_blocks;
    // half h of a block's samples (registers 16 h .. 16 h + 15)
    auto load_half = [&](cf* dst, uint32_t it, int h) {
        const uint32_t ch = one_channel ? 0u : it / n_blocks, b = it - ch * n_blocks;
        int ln = lane;
        asm volatile("" : "+v"(ln)); // addresses are formed here, not hoisted out of the block loop
        const cf* x = in + static_cast<size_t>(ch) * in_stride + static_cast<size_t>(b) * stride_s + ln;
#pragma unroll
        for (int j = 16 * h; j < 16 * h + 16; ++j) dst[j] = x[64 * j];
    };
    cf X[32];
    load_half(X, item, 0);
    load_half(X, item, 1);
    for (;;) {
        const uint32_t ch = one_channel ? 0u : item / n_blocks, blk = item - ch * n_blocks;
        float* zo = zpow + static_cast<size_t>(ch) * z_stride + static_cast<size_t>(blk) * stride_s;
        const uint32_t next = item + n_waves;
        const bool has_next = next < item_end;
        if (ABL == 1) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            float* zl = zo + (kFftN - ln);
            float pw[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) pw[j] = fmaf(X[j].y, X[j].y, X[j].x * X[j].x);
            if (has_next) {
                load_half(X, next, 0);
                load_half(X, next, 1);
            }
#pragma unroll
            for (int j = 4; j < 32; ++j)
                if (static_cast<uint32_t>(kFftN - (ln + 64 * j)) < stride_s) zl[-64 * j] = pw[j];
            if (!has_next) break;
            item = next;
            continue;
        }
        // ---- forward transform of the block (hpp:239-241)
        {
            cf bq[32];
            dft32(X);
            w1_exchange<false>(X, base, lane, row, ldsT, c, bq, nullptr, [] {});
            dft32(bq);
#pragma unroll
            for (int j = 0; j < 32; ++j) X[j] = bq[j];
        }
        if (noise_rel) { // hpp:257-265, as in k_correlate_w64
            float e = 0.0f;
#pragma unroll
            for (int j = 8; j < 24; ++j) e = fmaf(X[j].y, X[j].y, fmaf(X[j].x, X[j].x, e));
            asm volatile("s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                         : "+v"(e));
            if (lane == 63) zpow[static_cast<size_t>(ch) * z_stride + noise_rel + 1 + blk] = e;
        }
        // ---- X .* template in place (hpp:247-249), inverse-direction transform (hpp:250-251), powers (hpp:307-308)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float4 t = ldsTmpl[u * 64];
            X[2 * u] = cmul(X[2 * u], mk(t.x, t.y));
            X[2 * u + 1] = cmul(X[2 * u + 1], mk(t.z, t.w));
            if ((u & 1) == 1 && u >= 3) w64_pin4(X + 2 * u - 6); // at most four template reads ahead of their use
        }
        dft32(X);
        pc bp[16];
        w1_exchange<true>(X, base, lane, row, ldsT, c, nullptr, bp, [] {});
        // pass A's registers are dead: they take the samples of this wave's next block -- one half now (it arrives while
        // pass B runs; both halves in flight beside pass B's 64 + ~30 registers would spill), the other after pass B
        if (has_next && ABL != 2) load_half(X, next, 0);
        dft32p(bp);
        // pass B is complete HERE (its temporaries are dead) before the second half is requested: hipcc otherwise hoists
        // those loads above the arithmetic (197 live registers)
#pragma unroll
        for (int k = 0; k < 16; k += 4)
            asm volatile("" : "+v"(bp[k].r), "+v"(bp[k].i), "+v"(bp[k + 1].r), "+v"(bp[k + 1].i), "+v"(bp[k + 2].r),
                         "+v"(bp[k + 2].i), "+v"(bp[k + 3].r), "+v"(bp[k + 3].i));
        if (has_next && ABL != 2) load_half(X, next, 1);
        {
            int ln = lane;
            asm volatile("" : "+v"(ln)); // no hoisted lag registers
            float* zl = zo + (kFftN - ln); // lag of register j: 2048 - lane - 64 j (j = 0, lane = 0: lag 0)
            // (round 5, as in k_correlate_w64) with 1728 < stride_s registers 5 .. 31 are stored by every lane: one uniform
            // branch per block instead of a lane compare, an exec mask and a branch around each of the 29 stores
            const bool whole_rows = stride_s > static_cast<uint32_t>(kFftN - 64 * 5);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                // outputs k (lo) and k + 16 (hi) of the pair; registers 1 .. 3 hold lags >= 1793 > stride_s: never stored
                const cf pw = vfma(bp[k].i, bp[k].i, bp[k].r * bp[k].r);
                if (ABL == 2) {
                    asm volatile("" ::"v"(pw));
                    continue;
                }
                if (whole_rows) {
                    if (k == 0) {
                        if (ln == 0) zo[0] = pw.x;
                    } else if (k == 4) {
                        if (static_cast<uint32_t>(kFftN - 256 - ln) < stride_s) zl[-256] = pw.x;
                    } else if (k > 4) {
                        zl[-64 * k] = pw.x;
                    }
                    zl[-64 * (k + 16)] = pw.y;
                    continue;
                }
                if (k == 0) {
                    if (static_cast<uint32_t>((kFftN - ln) & (kFftN - 1)) < stride_s) zo[(kFftN - ln) & (kFftN - 1)] = pw.x;
                } else if (k > 3) {
                    if (static_cast<uint32_t>(kFftN - (ln + 64 * k)) < stride_s) zl[-64 * k] = pw.x;
                }
                if (static_cast<uint32_t>(kFftN - (ln + 64 * (k + 16))) < stride_s) zl[-64 * (k + 16)] = pw.y;
            }
        }
        if (!has_next) break;
        item = next;
    }
}

} // namespace
} // namespace gr4pm
