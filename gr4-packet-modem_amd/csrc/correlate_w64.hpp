// correlate_w64.hpp -- k_correlate_w64: the overlap-save correlator on the one-exchange wave FFT
// of fft2048_w64.hpp (device code, included by syncword_detection.hip).
// Replaces syncword_detection.hpp:238-252,300-313 (forward FFT of the block, x B templates, FFT,
// |.|^2, best bin per lag).
//
// One 512-thread workgroup per CU, eight PERSISTENT waves: every wave walks its own sequence of
// overlap-save blocks (item = channel * n_blocks + block), so the waves of a CU drift apart and
// one wave's HBM prologue / LDS exchange is covered by the arithmetic of the others.
//
// LDS (151.5 KiB of 160): [0, 16 KiB) the mid-stage twiddle table T (shared by the 8 waves) |
// eight private 16.5 KiB exchange buffers.  There is no room for a shared template buffer and
// none is needed: between the last mid-stage read of a transform and the first exchange store
// of the next one a wave's exchange buffer is idle, and during exactly that window the wave
// has the NEXT template copied global -> LDS into it by the LDS-DMA path (global_load_lds, no
// VGPRs).  So no wave ever waits for another one: the kernel has no barrier after its
// prologue, no spin loop and no hand-off words.
//
// Per transform and wave the LDS pipe now sees 64 ds_write_addtid_b32 (128 cycles) + 32 + 16 + 16
// ds_read_b128 (exchange rows, T, template: 256 cycles) where k_correlate needed ~800 cycles
// (two exchanges through ds_write2_b64 at 79 B/clk).
#pragma once
#include "fft2048_w64.hpp"

namespace gr4pm {
namespace {

constexpr int kW64Waves = 8, kW64Threads = kW64Waves * 64;
constexpr int kW64BufF4 = kW64BufDwords / 4;                         // 1056 float4 per wave
constexpr int kW64LdsF4 = kW64TwFloat4 + kW64Waves * kW64BufF4;      // 9472 float4 = 151552 B
static_assert(kW64LdsF4 * 16 <= 160 * 1024, "LDS budget of one workgroup per CU");

// exchange store: lane l writes r[k1] to row k1, column l (re plane, im plane 256 B further)
// with ds_write_addtid_b32: LDS address = M0 + offset + 4 * lane.  M0 is written ONCE, in front of the
// 64 stores (one wait state before the first DS use): rewriting it per group of stores made every
// group wait for the previous group's stores to leave the LDS queue (0.19 ms of 0.79 per 2^26 samples).
// hipcc itself never touches M0 in this kernel (no LDS-DMA builtin, no GWS / sendmsg / movrel): the
// only other writer is w64_dma_template, which sets it for itself.
#define GR4PM_ADDTID4(k)                                                                                              \
    asm volatile("ds_write_addtid_b32 %0 offset:%c8\n\tds_write_addtid_b32 %1 offset:%c9\n\t"                        \
                 "ds_write_addtid_b32 %2 offset:%c10\n\tds_write_addtid_b32 %3 offset:%c11\n\t"                      \
                 "ds_write_addtid_b32 %4 offset:%c12\n\tds_write_addtid_b32 %5 offset:%c13\n\t"                      \
                 "ds_write_addtid_b32 %6 offset:%c14\n\tds_write_addtid_b32 %7 offset:%c15"                          \
                 :                                                                                                    \
                 : "v"(r[(k)].x), "v"(r[(k)].y), "v"(r[(k) + 1].x), "v"(r[(k) + 1].y), "v"(r[(k) + 2].x),           \
                   "v"(r[(k) + 2].y), "v"(r[(k) + 3].x), "v"(r[(k) + 3].y), "i"((k) * kW64Row * 4),                  \
                   "i"((k) * kW64Row * 4 + 256), "i"(((k) + 1) * kW64Row * 4), "i"(((k) + 1) * kW64Row * 4 + 256),   \
                   "i"(((k) + 2) * kW64Row * 4), "i"(((k) + 2) * kW64Row * 4 + 256), "i"(((k) + 3) * kW64Row * 4),   \
                   "i"(((k) + 3) * kW64Row * 4 + 256)                                                                 \
                 : "memory")

__device__ __forceinline__ void w64_store(const cf* r, uint32_t base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base) : "memory");
    GR4PM_ADDTID4(0);
    GR4PM_ADDTID4(4);
    GR4PM_ADDTID4(8);
    GR4PM_ADDTID4(12);
    GR4PM_ADDTID4(16);
    GR4PM_ADDTID4(20);
    GR4PM_ADDTID4(24);
    GR4PM_ADDTID4(28);
}
#undef GR4PM_ADDTID4

// 16 KiB template, global -> this wave's exchange buffer, linear ([u][lane] float4), by the
// LDS-DMA path: 16 pieces of 1 KiB, LDS address of a piece = M0 + 16 * lane.  All LDS reads of
// the wave have returned before the first piece is issued (s_waitcnt lgkmcnt(0)): the DMA is
// not ordered against them by the LDS queue.
__device__ __forceinline__ void w64_dma_template(const float4* tg, uint32_t base, uint32_t voff)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // opaque copies: otherwise the 16 offsets / 16 LDS addresses are hoisted out of the block loop
    // and live (or spill) across the whole kernel
    asm volatile("" : "+v"(voff), "+s"(base));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        // global address = tg (SGPR pair) + per-lane byte offset (lane * 16 + u * 1024)
        const uint32_t vo = voff + u * 1024u;
        const uint32_t b0 = base + u * 1024u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(tg), "s"(b0) : "memory");
    }
}

// Issue order: the reads of group g + 1, then the arithmetic of group g, and nothing further ahead.
// A "memory" barrier alone is not enough -- hipcc keeps the reads in order but sinks all the
// arithmetic below them (48 x 16 bytes in flight = 192 VGPRs, spills): the barrier also takes the
// group's results as operands, so they exist before the next reads are issued.
__device__ __forceinline__ void w64_pin4(cf* b)
{
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
}
// the same arithmetic on two consecutive m at a time with packed FP32 (the planar reads put re[m], re[m+1]
// in one register pair): 8 packed instructions per pair + 2 v_pk_mov_b32 to interleave (re, im) for pass B,
// instead of 16 scalar ones -- fewer issue slots per wave (a lone wave issues one VALU op per 4 cycles,
// packed or not)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void w64_mid_pair(f2 a0r, f2 a1r, f2 a0i, f2 a1i, f2 tr, f2 ti, cf c, cf* b)
{
    f2 ur, ui, br, bi, t0, t1;
    // one statement (see cmul): ur = a0r + c.x a1r - c.y a1i ; ui = a0i + c.x a1i + c.y a1r (c.x / c.y broadcast with
    // op_sel) ; br = tr ur - ti ui ; bi = tr ui + ti ur ; then (br.lo, bi.lo), (br.hi, bi.hi)
    asm("v_pk_fma_f32 %4, %9, %14, %8 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %5, %11, %14, %10 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %11, %14, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %9, %14, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_mul_f32 %4, %12, %0\n\t"
        "v_pk_mul_f32 %5, %12, %1\n\t"
        "v_pk_fma_f32 %2, %13, %1, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %13, %0, %5\n\t"
        "v_pk_mov_b32 %6, %2, %3 op_sel:[0,0]\n\t"
        "v_pk_mov_b32 %7, %2, %3 op_sel:[1,1]"
        : "=&v"(ur), "=&v"(ui), "=&v"(br), "=&v"(bi), "=&v"(t0), "=&v"(t1), "=&v"(b[0]), "=&v"(b[1])
        : "v"(a0r), "v"(a1r), "v"(a0i), "v"(a1i), "v"(tr), "v"(ti), "v"(c));
}

template <int DEPTH, bool PACKED, bool FAKE = false>
__device__ __forceinline__ void w64_mid_dev(int lane, const float4* row, const float4* tT, cf c, cf* b)
{
    if (FAKE) { // timing-only ablation: no LDS reads, same arithmetic on register values
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 s0 = make_float4(c.x, c.y, c.x + g, c.y), s1 = make_float4(c.y, c.x, g, c.y);
            asm volatile("" : "+v"(s0.x), "+v"(s0.y), "+v"(s0.z), "+v"(s0.w), "+v"(s1.x), "+v"(s1.y), "+v"(s1.z), "+v"(s1.w));
            w64_mid_group(s0, s1, s1, s0, s0, s1, c, b + 4 * g);
            w64_pin4(b + 4 * g);
        }
        return;
    }
    // q[g % (DEPTH + 1)] holds group g's six reads; groups g + 1 .. g + DEPTH are in flight while g is consumed
    float4 q[DEPTH + 1][6];
    auto issue = [&](int g) {
        float4* d = q[g % (DEPTH + 1)];
        d[0] = row[g], d[1] = row[8 + g], d[2] = row[16 + g], d[3] = row[24 + g];
        d[4] = tT[(g * 2 + 0) * 64 + lane], d[5] = tT[(g * 2 + 1) * 64 + lane];
    };
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) issue(g);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g + DEPTH < 8) issue(g + DEPTH);
        const float4* s = q[g % (DEPTH + 1)];
        if (PACKED) {
            w64_mid_pair(f2{ s[0].x, s[0].y }, f2{ s[1].x, s[1].y }, f2{ s[2].x, s[2].y }, f2{ s[3].x, s[3].y },
                         f2{ s[4].x, s[4].y }, f2{ s[5].x, s[5].y }, c, b + 4 * g);
            w64_mid_pair(f2{ s[0].z, s[0].w }, f2{ s[1].z, s[1].w }, f2{ s[2].z, s[2].w }, f2{ s[3].z, s[3].w },
                         f2{ s[4].z, s[4].w }, f2{ s[5].z, s[5].w }, c, b + 4 * g + 2);
        } else {
            w64_mid_group(s[0], s[1], s[2], s[3], s[4], s[5], c, b + 4 * g);
        }
        w64_pin4(b + 4 * g);
    }
}

// the planar form of the mid stage (fft2048_w64.hpp: pc): same reads, same prefetch distance, outputs
// b[2g], b[2g + 1] = items (4g, 4g + 1), (4g + 2, 4g + 3) as (re, re) / (im, im) pairs: eight packed instructions
// per pair and no v_pk_mov_b32.  One asm statement per pair: hipcc pads every pair of ADJACENT DEPENDENT packed
// instructions of its own with an s_nop (it takes op_sel_hi of a VOP3P source for a dst_sel: "forwarding hazard";
// the hardware interlocks by itself, the interleaved kernel has run such pairs inside asm statements since round 1),
// and between two asm statements it pads every register overlap -- so the statement has no scratch outputs: u is
// formed in the registers of the row reads it consumes (dead afterwards).
__device__ __forceinline__ pc w64_mid_pair_asm(f2 a0r, f2 a1r, f2 a0i, f2 a1i, f2 tr, f2 ti, cf c)
{
    pc b;
    asm("v_pk_fma_f32 %2, %4, %8, %2 op_sel_hi:[1,0,1]\n\t"                                       // ur  = a0r + c.x a1r
        "v_pk_fma_f32 %3, %5, %8, %3 op_sel_hi:[1,0,1]\n\t"                                       // ui  = a0i + c.x a1i
        "v_pk_fma_f32 %2, %5, %8, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t" // ur -= c.y a1i
        "v_pk_fma_f32 %3, %4, %8, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                        // ui += c.y a1r
        "v_pk_mul_f32 %0, %6, %2\n\t"                                                             // br  = tr ur
        "v_pk_mul_f32 %1, %6, %3\n\t"                                                             // bi  = tr ui
        "v_pk_fma_f32 %0, %7, %3, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"                           // br -= ti ui
        "v_pk_fma_f32 %1, %7, %2, %1"                                                              // bi += ti ur
        : "=&v"(b.r), "=&v"(b.i), "+v"(a0r), "+v"(a0i)
        : "v"(a1r), "v"(a1i), "v"(tr), "v"(ti), "v"(c));
    return b;
}
template <int DEPTH>
__device__ __forceinline__ void w64_mid_dev_p(int lane, const float4* row, const float4* tT, cf c, pc* b)
{
    float4 q[DEPTH + 1][6];
    auto issue = [&](int g) {
        float4* d = q[g % (DEPTH + 1)];
        d[0] = row[g], d[1] = row[8 + g], d[2] = row[16 + g], d[3] = row[24 + g];
        d[4] = tT[(g * 2 + 0) * 64 + lane], d[5] = tT[(g * 2 + 1) * 64 + lane];
    };
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) issue(g);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g + DEPTH < 8) issue(g + DEPTH);
        const float4* s = q[g % (DEPTH + 1)];
        b[2 * g] = w64_mid_pair_asm(f2{ s[0].x, s[0].y }, f2{ s[1].x, s[1].y }, f2{ s[2].x, s[2].y }, f2{ s[3].x, s[3].y },
                                    f2{ s[4].x, s[4].y }, f2{ s[5].x, s[5].y }, c);
        b[2 * g + 1] = w64_mid_pair_asm(f2{ s[0].z, s[0].w }, f2{ s[1].z, s[1].w }, f2{ s[2].z, s[2].w },
                                        f2{ s[3].z, s[3].w }, f2{ s[4].z, s[4].w }, f2{ s[5].z, s[5].w }, c);
        // issue order: the reads of group g + 1, then the arithmetic of group g, and nothing further ahead (w64_pin4)
        asm volatile("" : "+v"(b[2 * g].r), "+v"(b[2 * g].i), "+v"(b[2 * g + 1].r), "+v"(b[2 * g + 1].i)::"memory");
    }
}

// tmpl: [bin][u = 16][lane = 64] float4 = (T[lane + 128 u], T[lane + 128 u + 64]) (conjugated
// template spectra, hpp:166-189); tT / cc: build_w64_tables.  total = n_channels * n_blocks items.
//
// Measured, not adopted (tools/w64_variants.py, MI355X, 2^26 samples): the samples of the next block requested a
// whole block ahead into a register set of their own when n_bins == 1 (0.232 against 0.227 ms: a block's fixed
// cost is its two transforms, not HBM latency -- without any sample load or power store the one-bin launch still
// takes 0.176 ms); waves started at 8 .. 64 different offsets so that they do not ask HBM for their next block at
// the same moment (0.846 .. 0.884 against 0.831 ms at nine bins: they are not in step to begin with); exchange stores
// issued from inside the last stage of pass A (no change); (re, im) interleaved exchange rows written with
// ds_write_b64 (0.82 against 0.78 ms); mid-stage and template reads further ahead (no change); round 5: the samples of
// the next block requested eight at a time between the last bin's products instead of 32 in a row behind them (2.756
// against 2.750 ms per 2^28 at nine bins, same box).
// Timing-only ablations (wrong results; tools/w64_variants.py, HISTORY.md section 3): 8 no template DMA, 32 no
// exchange stores, 64 no exchange / twiddle reads, 128 no template reads, 1024 no power / maximum, 2048 no power
// stores, 4096 no sample loads after the first block.
template <int VAR>
__global__ __launch_bounds__(kW64Threads, 2) void k_correlate_w64(const cf* __restrict__ in, size_t in_stride,
                                                                  uint32_t n_blocks, uint32_t total,
                                                                  uint32_t stride_s, int n_bins,
                                                                  const float4* __restrict__ tmpl,
                                                                  const float4* __restrict__ tT,
                                                                  const cf* __restrict__ cc,
                                                                  float* __restrict__ zpow, size_t z_stride,
                                                                  uint32_t blocks_per_wave, uint32_t noise_rel)
{
    __shared__ float4 lds4[kW64LdsF4];
    const int tid = threadIdx.x;
    // at least one frequency bin (the create call checks min <= max): without this the compiler keeps a path around the
    // bin loop on which the template loads requested during the forward transform are never awaited (tools/check_m0.py)
    __builtin_assume(n_bins >= 1);
#ifdef GR4PM_W64_PRIO
    __builtin_amdgcn_s_setprio(GR4PM_W64_PRIO); // A/B: make ABL=-DGR4PM_W64_PRIO=2
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kW64TwFloat4; i += kW64Threads) lds4[i] = tT[i];
    // the next item of this workgroup that no wave has taken yet (see `dynamic` below) lives in a pad word of the
    // exchange rows (row 0 of wave 0; only the LDS-DMA template variants ever write there, and they keep fixed shares):
    // one more byte of LDS would cost the kernels that run beside this one their place on the CU
    constexpr int kWgNextDword = 128;
    static_assert(kWgNextDword >= 2 * 64 && kWgNextDword < kW64Row,
                  "the hand-out counter must sit in the pad behind re[64] | im[64] of exchange row 0");
    uint32_t* const wg_next = reinterpret_cast<uint32_t*>(lds4 + kW64TwFloat4) + kWgNextDword;
    constexpr bool kDynamic = (VAR & 32768) != 0 && (VAR & 131072) == 0;
    // blocks_per_wave != 0: the workgroup's share of the items is [wg_begin, wg_end), an even split of `total` over the grid
    // (round 6: the host makes the grid a multiple of the CU count, so that the last round of workgroups is as full
    // as the others -- with shares of exactly 8 K items the 2^28-sample launch was 12.47 rounds of 256 workgroups, the
    // last one half empty: tools/overlap_save_pattern.hip measures the same loss on the access pattern alone)
    const uint32_t wg_begin = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * total / gridDim.x);
    const uint32_t wg_end = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * total / gridDim.x);
    if (kDynamic && tid == 0) *wg_next = wg_begin + kW64Waves;
    __syncthreads(); // the only workgroup-wide synchronisation of the kernel
    float4* xb4 = lds4 + kW64TwFloat4 + wave * kW64BufF4;
    // LDS byte address of the buffer (what DS instructions and M0 take)
    const uint32_t base = __builtin_amdgcn_readfirstlane(
        static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)xb4)));
    const float4* row = xb4 + (lane & 31) * (kW64Row / 4);
    const float4* ldsT = lds4;
    const cf c = cc[lane];
    const uint32_t voff = static_cast<uint32_t>(lane) * 16u;
    // blocks_per_wave == 0: persistent waves, wave w of the grid walks items w, w + W, w + 2W, ... (a stand-alone
    // launch).  blocks_per_wave == K: a workgroup owns about 8 K consecutive items (its even share of the grid's) and retires after them, so that the
    // dispatcher can place the workgroups of other streams' kernels on the CU in between (a pipelined receiver:
    // a persistent launch keeps every CU's LDS for its whole duration and everything else waits for it).
    // The 8 K items are handed out through a counter in LDS, one at a time and one block ahead (the samples of the
    // next block are requested during the last transform of the current one): a wave that shares its SIMD with a wave
    // of another kernel -- the serial kernels of the receiver fit beside two correlator waves and then take most of
    // that SIMD's issue slots for as long as they live -- takes fewer blocks instead of keeping the other six waves'
    // CU waiting for its fixed share (VAR & 131072: the fixed shares of rounds 1 - 3, for A/B).
    const bool dynamic = kDynamic && blocks_per_wave != 0;
    const uint32_t n_waves = blocks_per_wave ? kW64Waves : gridDim.x * kW64Waves;
    uint32_t item = blocks_per_wave ? wg_begin + wave : blockIdx.x * kW64Waves + wave;
    const uint32_t item_end = blocks_per_wave ? wg_end : total;
    if (item >= item_end) return;

    const bool one_channel = total <= n_blocks; // no division per block (two 32-bit divisions: ~50 instructions)
    auto load_block = [&](cf* dst, uint32_t it) {
        const uint32_t ch = one_channel ? 0u : it / n_blocks, b = it - ch * n_blocks;
        int ln = lane;
        asm volatile("" : "+v"(ln)); // addresses are formed here, not hoisted out of the block loop
        const cf* x = in + static_cast<size_t>(ch) * in_stride + static_cast<size_t>(b) * stride_s + ln;
#pragma unroll
        for (int j = 0; j < 32; ++j) dst[j] = x[64 * j];
    };
    auto exchange = [&](const cf* r, cf* bq) { // pass-A output -> pass-B input
        if (!(VAR & 32)) w64_store(r, base);
        if (VAR & 32) { // ablation: the stores are gone, pass A stays alive
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("" ::"v"(r[j]));
        }
        w64_mid_dev<1, true, (VAR & 64) != 0>(lane, row, ldsT, c, bq);
    };

    // VAR & 32768 (with the planar bin loop): the templates come straight from global memory (L2: B x 16 KiB, shared
    // by every wave of the chip) into registers, two halves of eight global_load_dwordx4: half A of the next bin is
    // requested when the mid stage's row reads have left their registers, a whole pass B ahead of its use; half B when
    // pass B's registers have died, in front of the product, which starts on half A.  No LDS-DMA (sixteen M0 writes and
    // their wait states per bin, issue behind the wave's last exchange read), no s_waitcnt vmcnt(0) per bin.
    constexpr bool TG = (VAR & 32768) != 0;
    f4v tqa[TG ? 8 : 1], tqb[TG ? 8 : 1];
    // eight global_load_dwordx4 in the saddr form: uniform base (SGPR pair, 4 KiB into the half so that the eight
    // 1-KiB steps fit the 13-bit immediate) + the lane's byte offset in ONE VGPR.  Written as asm: from C++ hipcc forms
    // 64-bit per-lane addresses for them (38 more vector instructions per bin).  The compiler does not know these are
    // loads, so the waits are explicit (tq_wait): s_waitcnt vmcnt(n) with n = the requests issued after the ones wanted.
    auto load_template_half = [&](f4v* dst, int bin, int half) {
        const float4* tb = tmpl + static_cast<size_t>(bin) * 1024 + half * 512 + 256;
        asm volatile("global_load_dwordx4 %0, %8, %9 offset:-4096\n\t"
                     "global_load_dwordx4 %1, %8, %9 offset:-3072\n\t"
                     "global_load_dwordx4 %2, %8, %9 offset:-2048\n\t"
                     "global_load_dwordx4 %3, %8, %9 offset:-1024\n\t"
                     "global_load_dwordx4 %4, %8, %9\n\t"
                     "global_load_dwordx4 %5, %8, %9 offset:1024\n\t"
                     "global_load_dwordx4 %6, %8, %9 offset:2048\n\t"
                     "global_load_dwordx4 %7, %8, %9 offset:3072"
                     : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]),
                       "=&v"(dst[6]), "=&v"(dst[7])
                     : "v"(voff), "s"(tb)
                     : "memory");
    };
    cf X[32];
    load_block(X, item);
    for (;;) {
        const uint32_t ch = one_channel ? 0u : item / n_blocks, blk = item - ch * n_blocks;
        float* zo = zpow + static_cast<size_t>(ch) * z_stride + static_cast<size_t>(blk) * stride_s;
        uint32_t next = item + n_waves; // dynamic: taken from the workgroup's counter in front of the last transform
        bool has_next = next < item_end;
        // ---- forward transform of the block (hpp:239-241)
        {
            cf bq[32];
            dft32(X);
            exchange(X, bq);
            if (TG) load_template_half(tqa, 0, 0);
            else if (!(VAR & 8)) w64_dma_template(tmpl, base, voff); // template 0 while pass B runs
            dft32(bq);
#pragma unroll
            for (int j = 0; j < 32; ++j) X[j] = bq[j];
        }
        if (noise_rel) {
            // hpp:257-265: the block's noise power is the energy of the spectrum's middle half, bins N/4 .. 3N/4 - 1 =
            // registers 8 .. 23 of every lane.  One float per block, behind the channel's powers (k_tags reads it for
            // the few blocks that hold a detection instead of transforming them again).  Unnormalised.
            float e = 0.0f;
#pragma unroll
            for (int j = 8; j < 24; ++j) e = fmaf(X[j].y, X[j].y, fmaf(X[j].x, X[j].x, e));
            // wave sum into lane 63 with DPP adds (an inclusive scan; no lane-index registers that LICM would keep
            // alive across the bin loop, unlike __shfl_xor: 218 instead of 210 VGPRs)
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
        float zmax[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) zmax[j] = -1.0f; // hpp:303
        for (int bin = 0; bin < n_bins; ++bin) {
            cf p[32], bq[32];
            if (!TG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the template has landed in the exchange buffer
            if (TG) {
                load_template_half(tqb, bin, 1);
                // half A (requested a pass B ago) has landed once only the eight requests just made are outstanding
                asm volatile("s_waitcnt vmcnt(8)"
                             : "+v"(tqa[0]), "+v"(tqa[1]), "+v"(tqa[2]), "+v"(tqa[3]), "+v"(tqa[4]), "+v"(tqa[5]),
                               "+v"(tqa[6]), "+v"(tqa[7])::"memory");
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { // hpp:247-249
                float4 t;
                if (TG) {
                    if (u == 8)
                        asm volatile("s_waitcnt vmcnt(0)"
                                     : "+v"(tqb[0]), "+v"(tqb[1]), "+v"(tqb[2]), "+v"(tqb[3]), "+v"(tqb[4]), "+v"(tqb[5]),
                                       "+v"(tqb[6]), "+v"(tqb[7])::"memory");
                    const f4v tv = u < 8 ? tqa[u] : tqb[u - 8];
                    t = make_float4(tv.x, tv.y, tv.z, tv.w);
                } else if (VAR & 128) {
                    t = make_float4(c.x, c.y, c.y, c.x);
                    asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
                } else {
                    t = xb4[u * 64 + lane];
                }
                p[2 * u] = cmul(X[2 * u], mk(t.x, t.y));
                p[2 * u + 1] = cmul(X[2 * u + 1], mk(t.z, t.w));
                if ((u & 1) == 1 && u >= 3) w64_pin4(p + 2 * u - 6); // at most four template reads ahead of their use
            }
            if (bin == n_bins - 1 && dynamic) {
                // as late as the prefetch below allows: a wave that is being held up has committed itself to the block
                // it is working on and to nothing else
                uint32_t taken = 0;
                if (lane == 0) taken = atomicAdd(wg_next, 1u);
                next = __builtin_amdgcn_readfirstlane(taken);
                has_next = next < item_end;
            }
            if (!(VAR & 4096) && bin == n_bins - 1 && has_next) {
                // the spectrum is dead: its registers take the samples of this wave's next block,
                // which arrive while the last transform of this one runs
                load_block(X, next);
            }
            dft32(p); // hpp:250-251
            if (VAR & 65536) {
                // planar second half: exchange stores as before, mid stage and pass B on (re, re) / (im, im) pairs
                pc bp[16];
                w64_store(p, base);
                w64_mid_dev_p<1>(lane, row, ldsT, c, bp);
                if (TG) {
                    if (bin + 1 < n_bins) load_template_half(tqa, bin + 1, 0);
                } else if (!(VAR & 8) && bin + 1 < n_bins) {
                    w64_dma_template(tmpl + static_cast<size_t>(bin + 1) * 1024, base, voff);
                }
                dft32p(bp);
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4) {
                    // hpp:307-308: powers of outputs k (lo) and k + 16 (hi) in two packed instructions
                    cf pw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pw[u] = vfma(bp[k0 + u].i, bp[k0 + u].i, bp[k0 + u].r * bp[k0 + u].r);
                    if ((VAR & 16384) && k0 == 0) { // registers 1 .. 3 are never stored
                        asm("v_max_f32 %0, %0, %5\n\tv_max_f32 %1, %1, %6\n\tv_max_f32 %2, %2, %7\n\tv_max_f32 %3, %3, %8\n\t"
                            "v_max_f32 %4, %4, %9"
                            : "+v"(zmax[0]), "+v"(zmax[16]), "+v"(zmax[17]), "+v"(zmax[18]), "+v"(zmax[19])
                            : "v"(pw[0].x), "v"(pw[0].y), "v"(pw[1].y), "v"(pw[2].y), "v"(pw[3].y));
                        continue;
                    }
                    asm("v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %10\n\tv_max_f32 %3, %3, %11\n\t"
                        "v_max_f32 %4, %4, %12\n\tv_max_f32 %5, %5, %13\n\tv_max_f32 %6, %6, %14\n\tv_max_f32 %7, %7, %15"
                        : "+v"(zmax[k0]), "+v"(zmax[k0 + 1]), "+v"(zmax[k0 + 2]), "+v"(zmax[k0 + 3]), "+v"(zmax[k0 + 16]),
                          "+v"(zmax[k0 + 17]), "+v"(zmax[k0 + 18]), "+v"(zmax[k0 + 19])
                        : "v"(pw[0].x), "v"(pw[1].x), "v"(pw[2].x), "v"(pw[3].x), "v"(pw[0].y), "v"(pw[1].y), "v"(pw[2].y),
                          "v"(pw[3].y));
                }
                continue;
            }
            exchange(p, bq);
            if (!(VAR & 8) && bin + 1 < n_bins) w64_dma_template(tmpl + static_cast<size_t>(bin + 1) * 1024, base, voff);
            dft32(bq);
            if (VAR & 1024) { // ablation: no power / maximum
#pragma unroll
                for (int j = 0; j < 32; ++j) asm volatile("" ::"v"(bq[j]));
            } else {
#pragma unroll
                for (int j0 = 0; j0 < 32; j0 += 8) {
                    // hpp:307-308: the best bin's power (max() == the strict-> scan for the VALUE).  v_max_f32 in asm
                    // (fmaxf adds a canonicalising second one), eight per statement (one boundary pad, not eight)
                    float pw[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pw[u] = fmaf(bq[j0 + u].y, bq[j0 + u].y, bq[j0 + u].x * bq[j0 + u].x);
                    if ((VAR & 16384) && j0 == 0) {
                        // lags of registers 1 .. 3 are 2048 - 64 j - lane >= 1793: never stored when stride_s <= 1793
                        // (the launch checks it), so neither their powers nor the last additions of pass B for them exist
                        asm("v_max_f32 %0, %0, %5\n\tv_max_f32 %1, %1, %6\n\tv_max_f32 %2, %2, %7\n\tv_max_f32 %3, %3, %8\n\t"
                            "v_max_f32 %4, %4, %9"
                            : "+v"(zmax[0]), "+v"(zmax[4]), "+v"(zmax[5]), "+v"(zmax[6]), "+v"(zmax[7])
                            : "v"(pw[0]), "v"(pw[4]), "v"(pw[5]), "v"(pw[6]), "v"(pw[7]));
                        continue;
                    }
                    asm("v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %10\n\tv_max_f32 %3, %3, %11\n\t"
                        "v_max_f32 %4, %4, %12\n\tv_max_f32 %5, %5, %13\n\tv_max_f32 %6, %6, %14\n\tv_max_f32 %7, %7, %15"
                        : "+v"(zmax[j0]), "+v"(zmax[j0 + 1]), "+v"(zmax[j0 + 2]), "+v"(zmax[j0 + 3]), "+v"(zmax[j0 + 4]),
                          "+v"(zmax[j0 + 5]), "+v"(zmax[j0 + 6]), "+v"(zmax[j0 + 7])
                        : "v"(pw[0]), "v"(pw[1]), "v"(pw[2]), "v"(pw[3]), "v"(pw[4]), "v"(pw[5]), "v"(pw[6]), "v"(pw[7]));
                }
            }
        }
        // lag k <-> correlation index (N - k) mod N (hpp:300); register j of lane l holds index l + 64 j
        if (VAR & 2048) { // ablation: no power stores
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("" ::"v"(zmax[j]));
        } else {
            int ln = lane;
            asm volatile("" : "+v"(ln)); // no 32 hoisted lag registers
            float* zl = zo + (kFftN - ln); // lag of register j: 2048 - lane - 64 j (j = 0, lane = 0: lag 0)
            // Round 5.  Register j >= 1 holds the lags 2048 - 64 j - lane <= 2048 - 64 j: with 1728 < stride_s (the
            // receiver's 1752; the pruned variants run for stride_s <= 1793 only) registers 5 .. 31 are stored by every
            // lane, register 4 by the lanes whose lag is below the stride and register 0 by lane 0 (lag 0).  One uniform
            // branch per block instead of a lane compare, an exec mask, a branch around the store and a 64-bit address
            // adjustment for each of the 29 stores (230 instructions and 29 branches per block: a third of a bin).
            if ((VAR & 16384) && !(VAR & 262144) && stride_s > static_cast<uint32_t>(kFftN - 64 * 5)) {
                if (ln == 0) zo[0] = zmax[0];
                if (static_cast<uint32_t>(kFftN - 256 - ln) < stride_s) zl[-256] = zmax[4];
#pragma unroll
                for (int j = 5; j < 32; ++j) zl[-64 * j] = zmax[j];
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if ((VAR & 16384) && j >= 1 && j <= 3) continue; // lags >= 1793 > stride_s
                    const uint32_t lag = static_cast<uint32_t>((kFftN - (ln + 64 * j)) & (kFftN - 1));
                    if (j == 0) {
                        if (lag < stride_s) zo[lag] = zmax[0];
                    } else if (lag < stride_s) {
                        zl[-64 * j] = zmax[j];
                    }
                }
            }
        }
        if (!has_next) break;
        item = next;
    }
}

} // namespace
} // namespace gr4pm
