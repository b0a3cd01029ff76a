// stream_blocks.hip -- the sample-rate / symbol-rate blocks around the correlator:
//   Rotator, CoarseFrequencyCorrection, CostasLoop, SyncwordWipeoff, SyncwordDetectionFilter,
//   InterpolatingFirFilter, SymbolFilter, PfbArbResampler
// (reference headers cited at each entry point; paths relative to
//  /root/reference/blocks/include/gnuradio-4.0/packet-modem/).
//
// This translation unit is compiled with -ffp-contract=off: the reference evaluates every
// product and sum separately (baseline x86-64, std::inner_product / std::complex), and the
// FIR outputs here are bit-exact with that order.
//
// Pattern shared by all blocks: tags are sparse, so the tag-driven control flow of the
// reference (which is per-chunk C++ on the CPU) is replayed on the host over the TAG LIST
// only -- never over samples -- and turned into a small table of segments/runs; the kernels
// then process every sample / symbol of the call in parallel from that table.  Recurrences
// whose float rounding makes them order dependent (rotator phasor, Costas PLL, resampler
// phase accumulator) run serially per independent segment (one lane each).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <deque>
#include <vector>

#include "common.hpp"
#include "hostlogic/sdf_gate.hpp"
#include "hostlogic/symbol_filter_replay.hpp"
#include "hostlogic/packet_control.hpp"

namespace gr4pm {
// GR4PM_TIMING_SKIP=name[,name]: timing experiments only -- the named kernels are not launched (their outputs are
// garbage); tells what a kernel costs the pipelined chain, which its duration alone does not
#ifndef GR4PM_SERIAL_PRIO
#define GR4PM_SERIAL_PRIO 3 // s_setprio of the Costas kernels (A/B: make EXTRA=-DGR4PM_SERIAL_PRIO=0)
#endif
#ifndef GR4PM_ROT_PRIO
#define GR4PM_ROT_PRIO GR4PM_SERIAL_PRIO // ... of k_rot_checkpoints, the one serial kernel that runs BESIDE correlator waves
#endif
#ifndef GR4PM_EXPERIMENTS
static constexpr bool timing_skip(const char*) { return false; } // the shipped library leaves no kernel out
#else
static inline bool timing_skip(const char* name)
{
    // comma-separated list, whole names ("symf" does not match "symf_fake"); read once, announced on stderr
    static const char* e = gr4pm::experiment_env("GR4PM_TIMING_SKIP", true);
    if (!e) return false;
    const size_t n = strlen(name);
    for (const char* p = e; (p = strstr(p, name)) != nullptr; p += n)
        if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
#endif
namespace {

struct cf {
    float x, y;
};
__host__ __device__ __forceinline__ cf cmul(cf a, cf b)
{
    return { a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}
__host__ __device__ __forceinline__ cf cadd(cf a, cf b) { return { a.x + b.x, a.y + b.y }; }
__host__ __device__ __forceinline__ cf fmulc(float t, cf z) { return { t * z.x, t * z.y }; }

// std::abs(std::complex<float>) == hypotf; glibc evaluates it as
// (float)sqrt((double)x*x + (double)y*y), reproduced here with IEEE double ops.
__device__ __forceinline__ float hypot_like_glibc(float x, float y)
{
    const double dx = x, dy = y;
    return static_cast<float>(sqrt(dx * dx + dy * dy));
}

size_t bit_ceil_sz(size_t v)
{
    size_t c = 1;
    while (c < v) c <<= 1;
    return c;
}

// =====================================================================================
// Rotator (rotator.hpp:44-65) and CoarseFrequencyCorrection
// (coarse_frequency_correction.hpp:50-98): y = x * e; e *= e_incr; renormalise every 512.
// =====================================================================================
struct RotState {
    cf exp, incr;
    unsigned counter;
    unsigned pad;
};
struct RotSeg {
    unsigned long long start; // offset inside the channel row
    unsigned long long len;
    unsigned channel;
    unsigned ck0;  // first checkpoint slot of this segment
    int mode;      // 0 continue from state[channel]; 1 set_freq: (exp0, incr)
    int last;      // writes state[channel] back
    cf exp0, incr;
};
constexpr unsigned kRotChunk = 8; // samples per checkpoint

// one step of the phasor recurrence (rotator.hpp:58-63): e *= inc; renormalise when the
// incremented counter is a multiple of 512
__device__ __forceinline__ void rot_step(cf& e, cf inc, unsigned& counter)
{
    e = cmul(e, inc);
    if ((++counter & 511u) == 0) {
        const float r = hypot_like_glibc(e.x, e.y);
        e = { e.x / r, e.y / r };
    }
}

// cmul(a, b) as three packed instructions: (a.x b.x, a.x b.y), (a.y b.y, a.y b.x), then
// (t.x - u.x, t.y + u.y) -- the same four products and two sums, each rounded once
__device__ __forceinline__ __attribute__((unused)) cf cmul_pk(cf a, cf b)
{
    cf t, u, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(u) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(t), "v"(u));
    return r;
}

// kRotChunk steps e *= inc without renormalisation, three packed instructions a step:
//   a = (e.x * inc.x, e.x * inc.y)   b = (e.y * inc.y, e.y * inc.x)   e = (a.x - b.x, a.y + b.y)
// which are exactly the four products and two sums of cmul(), each rounded once (the sign of
// b.x is an input modifier).  No s_nop between a packed result and its packed consumer: the hardware
// interlocks (hipcc pads such pairs because it takes op_sel_hi of a VOP3P source for a dst_sel; the
// correlator's cmul has run them unpadded, bit-exact, since round 1).  hipcc itself spends seven instructions
// and four dependent levels a step on the same arithmetic (it builds both a + b and a - b and moves halves around).
__device__ __forceinline__ cf rot_chunk_pk(cf e, cf inc)
{
    static_assert(kRotChunk == 8, "eight unrolled steps below");
    cf a, b;
#define GR4PM_ROT_STEP                                                  \
    "v_pk_mul_f32 %[a], %[e], %[i] op_sel_hi:[0,1]\n"                   \
    "v_pk_mul_f32 %[b], %[e], %[i] op_sel:[1,1] op_sel_hi:[1,0]\n"      \
    "v_pk_add_f32 %[e], %[a], %[b] neg_lo:[0,1] neg_hi:[0,0]\n"
    asm volatile(GR4PM_ROT_STEP GR4PM_ROT_STEP GR4PM_ROT_STEP GR4PM_ROT_STEP GR4PM_ROT_STEP GR4PM_ROT_STEP
                     GR4PM_ROT_STEP GR4PM_ROT_STEP
                 : [e] "+v"(e), [a] "=&v"(a), [b] "=&v"(b)
                 : [i] "v"(inc));
#undef GR4PM_ROT_STEP
    return e;
}

// Measured, not adopted (round 3): four chains per lane, interleaved (k_rot_checkpoints4: a quarter of the waves).
// A chain's step is NOT latency-bound on this chip: the four-chain kernel took 3.7 x the time of the one-chain kernel
// (1.49 against 0.40 ms for 10 004 packet segments, with or without the checkpoint stores) -- each packed instruction
// costs the same ~5 ns whether its neighbours depend on it or not, and the pipelined receiver lost 6 % (stage latency).
// serial: one lane per segment, phasor checkpoints every kRotChunk samples.  The chain of
// dependent complex multiplies is the whole cost, so the loop body is kept to exactly that.
// Register budget: at most 32 VGPRs, on purpose.  These waves live for half a millisecond; a SIMD that runs two
// correlator waves (2 x 240 registers) has exactly 32 left, so a wave of this kernel fits BESIDE them instead of keeping
// the next correlator workgroup off its CU (HISTORY.md section 9).  Hence: segment fields are re-read where they are
// needed instead of kept, chunk counts are 32 bit (a segment is shorter than 2^35 items), one running pointer.
__device__ __forceinline__ void rot_checkpoints_generic(unsigned lane_seg, const RotSeg* __restrict__ segs, unsigned n_segs,
                                                        const RotState* __restrict__ state,
                                                        RotState* __restrict__ state_next, cf* __restrict__ ck,
                                                        cf* __restrict__ seg_incr, unsigned* __restrict__ seg_counter0,
                                                        const unsigned* __restrict__ order)
{
    if (lane_seg >= n_segs) return;
    // order[]: the segments by descending length, so that the long ones (a stream with missed detections) share
    // waves -- a wave lives as long as its longest lane (see costas_process_impl); the array itself stays sorted by
    // position (k_rot_apply and the fused symbol filter search it)
    const unsigned s = order[lane_seg];
    const RotSeg* gp = segs + s;
    cf e, inc;
    unsigned counter;
    if (gp->mode == 2) {
        // a fixed point of the recurrence (see gr4pm_rotator::fixed): no chain; k_rot_const_fill writes the checkpoints
        seg_incr[s] = gp->incr;
        seg_counter0[s] = 0;
        if (gp->last) {
            RotState st;
            st.exp = gp->exp0;
            st.incr = gp->incr;
            st.counter = 0; // (irrelevant while the phasor is fixed; the next set_freq() resets it)
            st.pad = 0;
            state_next[gp->channel] = st;
        }
        return;
    }
    if (gp->mode == 0) {
        const RotState* st = state + gp->channel;
        e = st->exp;
        inc = st->incr;
        counter = st->counter;
    } else {
        e = gp->exp0;
        inc = gp->incr;
        counter = 0;
    }
    seg_incr[s] = inc;
    seg_counter0[s] = counter;
    cf* ckp = ck + gp->ck0;
    unsigned left = static_cast<unsigned>(gp->len / kRotChunk);
    // Round 6 (see k_rot_checkpoints_fresh): whole periods of 64 chunks in a loop without per-chunk decisions.  The
    // renormalisation falls into every 64th chunk, always the same one: the chunk loop below runs up to it (`lead` chunks),
    // then every period is that chunk item by item + 63 plain chunks under a scalar counter; what is left (less than a
    // period) goes through the chunk loop again.  Checkpoints leave 8 bytes at a time here: a continuation's slot has any
    // alignment, and there is one such segment per channel.
    const unsigned lead = ((512u - (counter & 511u)) - 1u) / kRotChunk; // plain chunks in front of the renormalising one
    const bool periods = left > lead && left - lead >= 64u;
    for (int phase = 0; phase < 2; ++phase) {
    const unsigned stop = phase == 0 && periods ? left - lead : 0u; // chunks still to do when this phase ends
    while (left != stop) {
        // (rounds 3 - 5 took four chunks per pass here while no renormalisation fell into them, with 16-byte stores where the
        // slot was aligned -- decided per lane; the periods below have taken that over, this loop sees less than 64 chunks)
        *ckp++ = e;
        --left;
        if ((counter & 511u) < 512u - kRotChunk) { // no renormalisation inside this chunk
            e = rot_chunk_pk(e, inc);
            counter += kRotChunk;
        } else { // one chunk in 64: rolled, ONE instance of the renormalisation's double-precision square root
#pragma unroll 1
            for (unsigned j = 0; j < kRotChunk; ++j) rot_step(e, inc, counter);
        }
    }
    if (phase == 0 && periods) {
        for (; left >= 64u; left -= 64u) {
            *ckp++ = e;
#pragma unroll 1
            for (unsigned j = 0; j < kRotChunk; ++j) rot_step(e, inc, counter); // the period's renormalising chunk
#pragma unroll 1
            for (int k = 0; k < 63; ++k) {
                *ckp++ = e;
                e = rot_chunk_pk(e, inc);
            }
            counter += 63 * kRotChunk;
        }
    }
    } // phases
    gp = segs + s; // (recomputed: one register kept across the loop instead of two)
    const unsigned rem = static_cast<unsigned>(gp->len) & (kRotChunk - 1);
    if (rem) {
        *ckp = e;
        for (unsigned j = 0; j < rem; ++j) rot_step(e, inc, counter);
    }
    if (gp->last) {
        RotState st;
        st.exp = e;
        st.incr = inc;
        st.counter = counter;
        st.pad = 0;
        state_next[gp->channel] = st; // (another row of the ring: another lane may still have to read `state`)
    }
}
__global__ __launch_bounds__(64) void k_rot_checkpoints(const RotSeg* __restrict__ segs, unsigned n_segs,
                                                        const RotState* __restrict__ state,
                                                        RotState* __restrict__ state_next, cf* __restrict__ ck,
                                                        cf* __restrict__ seg_incr, unsigned* __restrict__ seg_counter0,
                                                        const unsigned* __restrict__ order)
{
    __builtin_amdgcn_s_setprio(GR4PM_ROT_PRIO); // latency-bound, few waves
    rot_checkpoints_generic(blockIdx.x * blockDim.x + threadIdx.x, segs, n_segs, state, state_next, ck, seg_incr, seg_counter0,
                            order);
}

// Round 6: the segments that START at a set_freq event (mode 1; the host gives them an even checkpoint slot) in a loop
// without per-chunk decisions.  tools/lone_wave_issue.hip: the chain's three packed instructions cost a lone wave 5.2 core
// clocks each, 6.5 ns a step -- the kernel above takes 12 - 16: every chunk of eight steps it tests the renormalisation
// counter and the slot's alignment per LANE (the lanes of a wave disagree, so both paths run), and the renormalisation's
// chunk goes through a rolled loop.  A fresh segment's counter starts at 0: a period of 512 steps is 31 pairs of chunks
// (one 16-byte store each), one more chunk, seven plain steps and the step that renormalises -- the same operations in
// the same order, under a SCALAR loop counter.  The continuations (mode 0: any counter, any alignment) stay above.
__device__ __forceinline__ void rot_checkpoints_fresh(unsigned lane_seg, const RotSeg* __restrict__ segs, unsigned n_segs,
                                                      RotState* __restrict__ state_next, cf* __restrict__ ck,
                                                      cf* __restrict__ seg_incr, unsigned* __restrict__ seg_counter0,
                                                      const unsigned* __restrict__ order)
{
    if (lane_seg >= n_segs) return;
    const unsigned s = order[lane_seg];
    const RotSeg* gp = segs + s;
    if (gp->mode == 2) { // a fixed point of the recurrence: no chain (k_rot_checkpoints)
        seg_incr[s] = gp->incr;
        seg_counter0[s] = 0;
        if (gp->last) {
            RotState st;
            st.exp = gp->exp0;
            st.incr = gp->incr;
            st.counter = 0;
            st.pad = 0;
            state_next[gp->channel] = st;
        }
        return;
    }
    cf e = gp->exp0;
    const cf inc = gp->incr;
    seg_incr[s] = inc;
    seg_counter0[s] = 0;
    float4* ckp = reinterpret_cast<float4*>(ck + gp->ck0); // (an even slot)
    unsigned chunks = static_cast<unsigned>(gp->len / kRotChunk);
    unsigned counter = 0;
    constexpr unsigned kPeriodChunks = 512 / kRotChunk;
    for (; chunks >= kPeriodChunks; chunks -= kPeriodChunks) {
#pragma unroll 1
        for (int pair = 0; pair < static_cast<int>(kPeriodChunks / 2) - 1; ++pair) {
            const cf e0 = e;
            e = rot_chunk_pk(e, inc);
            *ckp++ = make_float4(e0.x, e0.y, e.x, e.y);
            e = rot_chunk_pk(e, inc);
        }
        const cf e0 = e;
        e = rot_chunk_pk(e, inc);
        *ckp++ = make_float4(e0.x, e0.y, e.x, e.y);
        counter += 512 - kRotChunk;
#pragma unroll 1
        for (unsigned j = 0; j < kRotChunk; ++j) rot_step(e, inc, counter); // (its last step renormalises)
    }
    // less than a period is left: no renormalisation any more
    for (; chunks >= 2; chunks -= 2) {
        const cf e0 = e;
        e = rot_chunk_pk(e, inc);
        *ckp++ = make_float4(e0.x, e0.y, e.x, e.y);
        e = rot_chunk_pk(e, inc);
        counter += 2 * kRotChunk;
    }
    cf* ck1 = reinterpret_cast<cf*>(ckp);
    if (chunks) {
        *ck1++ = e;
        e = rot_chunk_pk(e, inc);
        counter += kRotChunk;
    }
    gp = segs + s; // (recomputed: one register kept across the loops instead of two)
    const unsigned rem = static_cast<unsigned>(gp->len) & (kRotChunk - 1);
    if (rem) {
        *ck1 = e;
        for (unsigned j = 0; j < rem; ++j) rot_step(e, inc, counter);
    }
    if (gp->last) {
        RotState st;
        st.exp = e;
        st.incr = inc;
        st.counter = counter;
        st.pad = 0;
        state_next[gp->channel] = st;
    }
}

__global__ __launch_bounds__(64) void k_rot_checkpoints_fresh(const RotSeg* __restrict__ segs, unsigned n_segs,
                                                              RotState* __restrict__ state_next, cf* __restrict__ ck,
                                                              cf* __restrict__ seg_incr, unsigned* __restrict__ seg_counter0,
                                                              const unsigned* __restrict__ order)
{
    __builtin_amdgcn_s_setprio(GR4PM_ROT_PRIO); // latency-bound, few waves
    rot_checkpoints_fresh(blockIdx.x * blockDim.x + threadIdx.x, segs, n_segs, state_next, ck, seg_incr, seg_counter0, order);
}
// one launch for a whole plan on one stream: the first `fresh_blocks` workgroups take the n_fresh event-started entries of
// order[], the others the continuations behind them -- side by side, as in the one kernel of rounds 1 - 5 (two launches on
// one stream would run the continuation's chain BEHIND the others: 240 + 390 us per 2^28 samples where one kernel took 420)
__global__ __launch_bounds__(64) void k_rot_checkpoints_both(const RotSeg* __restrict__ segs, unsigned n_fresh,
                                                             unsigned fresh_blocks, unsigned n_rest,
                                                             const RotState* __restrict__ state,
                                                             RotState* __restrict__ state_next, cf* __restrict__ ck,
                                                             cf* __restrict__ seg_incr, unsigned* __restrict__ seg_counter0,
                                                             const unsigned* __restrict__ order)
{
    __builtin_amdgcn_s_setprio(GR4PM_ROT_PRIO); // latency-bound, few waves
    if (blockIdx.x < fresh_blocks) // (uniform)
        rot_checkpoints_fresh(blockIdx.x * blockDim.x + threadIdx.x, segs, n_fresh, state_next, ck, seg_incr, seg_counter0, order);
    else
        rot_checkpoints_generic((blockIdx.x - fresh_blocks) * blockDim.x + threadIdx.x, segs, n_rest, state, state_next, ck,
                                seg_incr, seg_counter0, order + n_fresh);
}

// (tests only: GR4PM_TEST_ROT_DELAY_US) keeps a stream busy for `us` microseconds
__global__ void k_test_delay(unsigned us)
{
    const unsigned long long t0 = wall_clock64(); // 100 MHz
    while (wall_clock64() - t0 < 100ull * us) __builtin_amdgcn_s_sleep(64);
}

// the checkpoints of the segments whose phasor is a fixed point of the recurrence (RotSeg::mode == 2): the constant
__global__ __launch_bounds__(256) void k_rot_const_fill(const RotSeg* __restrict__ segs, const unsigned* __restrict__ list,
                                                        cf* __restrict__ ck)
{
    const RotSeg* g = segs + list[blockIdx.y];
    const unsigned long long n = (g->len + kRotChunk - 1) / kRotChunk;
    const cf e = g->exp0;
    cf* dst = ck + g->ck0;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<unsigned long long>(gridDim.x) * blockDim.x)
        dst[i] = e;
}

// parallel: one lane per SAMPLE (coalesced 8-byte accesses); the lane replays at most
// kRotChunk-1 steps of the recurrence from its chunk's checkpoint, in the reference's order
__global__ __launch_bounds__(256) void k_rot_apply(const RotSeg* __restrict__ segs, unsigned n_segs,
                                                   const cf* __restrict__ ck, const cf* __restrict__ seg_incr,
                                                   const unsigned* __restrict__ seg_counter0,
                                                   const cf* __restrict__ in, cf* __restrict__ out,
                                                   size_t stride)
{
    // blockIdx.y = segment; grid-stride over the segment's samples
    const RotSeg g = segs[blockIdx.y];
    const cf inc = seg_incr[blockIdx.y];
    const unsigned c0 = seg_counter0[blockIdx.y];
    const size_t base = static_cast<size_t>(g.channel) * stride + g.start;
    for (unsigned long long j = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; j < g.len;
         j += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
        const unsigned long long c = j / kRotChunk;
        const unsigned steps = static_cast<unsigned>(j - c * kRotChunk);
        cf e = ck[g.ck0 + c];
        unsigned counter = c0 + static_cast<unsigned>(c * kRotChunk);
        for (unsigned t = 0; t < steps; ++t) rot_step(e, inc, counter);
        out[base + j] = cmul(in[base + j], e);
    }
    (void)n_segs;
}

// x[t] *= e_t for the eight items of a checkpoint chunk, e_(t+1) = e_t * inc in between (coarse_frequency_correction.hpp:
// 87, rotator.hpp:58-59): cmul_pk's three packed instructions per product, all 45 in ONE statement -- between separate
// asm statements hipcc pads every dependent packed pair with an s_nop (21 of them in this chain).
__device__ __forceinline__ void rot8_pk(cf (&x)[kRotChunk], cf e, cf inc)
{
    static_assert(kRotChunk == 8, "eight items below");
    cf a, b;
#define GR4PM_X(n)                                                          \
    "v_pk_mul_f32 %[a], %[x" #n "], %[e] op_sel_hi:[0,1]\n"                 \
    "v_pk_mul_f32 %[b], %[x" #n "], %[e] op_sel:[1,1] op_sel_hi:[1,0]\n"    \
    "v_pk_add_f32 %[x" #n "], %[a], %[b] neg_lo:[0,1] neg_hi:[0,0]\n"
#define GR4PM_E                                                             \
    "v_pk_mul_f32 %[a], %[e], %[i] op_sel_hi:[0,1]\n"                       \
    "v_pk_mul_f32 %[b], %[e], %[i] op_sel:[1,1] op_sel_hi:[1,0]\n"          \
    "v_pk_add_f32 %[e], %[a], %[b] neg_lo:[0,1] neg_hi:[0,0]\n"
    asm(GR4PM_X(0) GR4PM_E GR4PM_X(1) GR4PM_E GR4PM_X(2) GR4PM_E GR4PM_X(3) GR4PM_E GR4PM_X(4) GR4PM_E GR4PM_X(5) GR4PM_E
            GR4PM_X(6) GR4PM_E GR4PM_X(7)
        : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]),
          [x6] "+v"(x[6]), [x7] "+v"(x[7]), [e] "+v"(e), [a] "=&v"(a), [b] "=&v"(b)
        : [i] "v"(inc));
#undef GR4PM_X
#undef GR4PM_E
}

// =====================================================================================
// CostasLoop (costas_loop.hpp:92-148): serial per segment (state fully reset by a
// syncword_phase tag, :35-42), one lane per segment.
// =====================================================================================
struct CostasState {
    float phase, freq;
};
struct CostasSeg {
    unsigned long long start;
    unsigned len;
    unsigned channel;
    int mode; // 0 continue, 1 set_phase(phase0)
    int last;
    float phase0;
    float pad;
};

// cos/sin of the loop phase, BIT-EXACT with glibc's cosf / sinf / sincosf (what the reference's
// std::cos(float) / std::sin(float) call, costas_loop.hpp:113-115).  glibc >= 2.28 evaluates them in
// double (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h: Szabolcs Nagy's routines): the
// quadrant n from x * (2/pi * 2^24) by an integer shift, x - n * (pi/2 as a double), one sine and one
// cosine polynomial in x^2, ONE rounding to float.  MI355X has the FP64 rate to do the same, so the
// loop's local oscillator carries the reference's bits instead of "< 1 ULP" ones.  Pinned against the
// host libm for every float of |x| <= 3.2 (tests/sincosf_glibc_check.c: 0 mismatches, with and
// without FMA contraction) and on the device by test_device_sincosf_is_glibc_bit_exact.
// Valid for |x| < 120 (glibc's reduce_fast range; the loop phase is wrapped to [-pi, pi)).
__device__ __forceinline__ void sincosf_glibc(float y, float* s_out, float* c_out)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0; // 2/pi * 2^24, pi/2
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                 S3 = -0x1.994eb3774cf24p-13;
    const double x0 = static_cast<double>(y);
    // reduce_fast (for |y| < pi/4 this yields n = 0 and x = x0: the same as glibc's short path)
    const int n = (static_cast<int>(x0 * hpi_inv) + 0x800000) >> 24;
    const double x = fma(-static_cast<double>(n), hpi, x0);
    const double x2 = x * x;
    // sinf_poly, even n
    const double x3 = x * x2;
    const double s1 = fma(x2, S3, S2);
    const double x7 = x3 * x2;
    const double sp = fma(x7, s1, fma(x3, S1, x));
    // sinf_poly, odd n
    const double x4 = x2 * x2;
    const double c2 = fma(x2, C4, C3);
    const double c1 = fma(x2, C1, C0);
    const double x6 = x4 * x2;
    const double cp = fma(x6, c2, fma(x4, C2, c1));
    float sn = static_cast<float>(sp), cs = static_cast<float>(cp);
    // Tiny arguments (|y| < 2^-12): glibc returns y and 1.0f without evaluating anything.  The polynomials give the same
    // bits by themselves -- cos: x^2 |C1| < 2^-25, so cp > 1 - 2^-25 rounds to 1.0f; sin: sp = x (1 - d) with d < 2^-26
    // rounds back to y -- except for y = -0.0f, where the odd polynomial's sums produce +0.0.  The sign of the sine
    // polynomial IS the sign of the reduced argument whenever the result is not zero, so copying x's sign bit into sn
    // (one v_bfi_b32, no compare, no select) leaves every other value alone and restores the signed zero
    // (tests/sincosf_glibc_check.c and test_device_sincosf_is_glibc_bit_exact sweep it).
    // v_bitop3_b32 (gfx950): any function of three words in one instruction; 0xca = (a & b) | (~a & c), 0x78 = a ^ (b & c)
    const unsigned sb = __builtin_amdgcn_bitop3_b32(0x7fffffffu, __float_as_uint(sn), static_cast<unsigned>(__double2hiint(x)), 0xca);
    const unsigned cb = __float_as_uint(cs);
    // quadrant: sign[n & 3] on the sine argument (an odd polynomial: exact negation), second table =
    // cosine polynomial negated when n & 2; odd n swaps the two.  Bit arithmetic on a 0 / ~0 mask instead of
    // compare + select: a v_cmp result needs wait states before the v_cndmask that reads it, and this chain has
    // nothing to fill them with.  Eight instructions for tiny / swap / signs together (round 2: fifteen + 4 s_nop).
    const unsigned swap = static_cast<unsigned>(__builtin_amdgcn_sbfe(n, 0, 1)); // 0 or ~0
    const unsigned s0 = __builtin_amdgcn_bitop3_b32(swap, cb, sb, 0xca);
    const unsigned c0 = __builtin_amdgcn_bitop3_b32(swap, sb, cb, 0xca);
    const unsigned q = static_cast<unsigned>(n) << 30; // bit 31 = n & 2, bit 30 = n & 1
    *s_out = __uint_as_float(__builtin_amdgcn_bitop3_b32(s0, q, 0x80000000u, 0x78));
    *c_out = __uint_as_float(__builtin_amdgcn_bitop3_b32(c0, q + 0x40000000u, 0x80000000u, 0x78));
}

// costas_loop.hpp:141-145: phase >= pi ? phase - 2 pi : (phase < -pi ? phase + 2 pi : phase), without compares:
//   up   = clamp(phase * K - below(pi) * K)   1.0f for phase > below(pi) <=> phase >= pi (below = the next float down),
//   down = clamp(-phase * K - pi * K)         1.0f for phase < -pi, else 0.0f (K = 2^100: the smallest positive
//                                             difference, one ulp of pi = 2^-22, still scales past 1)
//   phase = fma(up - down, -2 pi, phase)
// up - down is 1, -1 or +0 (at most one of the two is 1): fma(+-1, -2 pi, phase) is the reference's single rounding
// of phase -+ 2 pi, and fma(+0, -2 pi, phase) = -0 + phase is phase bit for bit, the signed zeros included (a product
// of +0 with a POSITIVE constant would turn a phase of -0.0 into +0.0).  Four instructions, no VCC round trip (was
// six + wait states).
__device__ __forceinline__ float costas_wrap(float phase, float pi_f)
{
    const float K = 0x1p100f, two_pi = 2.0f * pi_f;
    const float pi_below = __uint_as_float(__float_as_uint(pi_f) - 1u);
    float up, down;
    asm("v_fma_f32 %0, %2, %3, -%4 clamp\n\t"
        "v_fma_f32 %1, -%2, %3, -%5 clamp"
        : "=&v"(up), "=&v"(down)
        : "v"(phase), "v"(K), "v"(pi_below * K), "v"(pi_f * K));
    return __builtin_fmaf(up - down, -two_pi, phase);
}

// one PLL iteration, costas_loop.hpp:112-146.  Everything is straight-line code without exec-mask branches; the phase
// wrap and sincosf's quadrant logic also without compares (round 3; the QPSK error term keeps its two selects): the
// chain of dependent operations of one iteration is the whole cost of the block.
template <int CONSTELLATION>
__device__ __forceinline__ cf costas_step(cf x, float& phase, float& freq, float k1, float k2)
{
    const float pi_f = 3.14159265358979323846f;
    float sn, cs;
#ifdef GR4PM_COSTAS_HW_SINCOS
    sn = __sinf(phase);
    cs = __cosf(phase);
#else
    sincosf_glibc(phase, &sn, &cs);
#endif
    const cf lo = { cs, -sn }; // :114-115
    const cf z = cmul(x, lo);
    float error;
    if constexpr (CONSTELLATION == 0) error = z.y;
    else if constexpr (CONSTELLATION == 1) error = z.x * z.y;
    else error = (z.x > 0 ? z.y : -z.y) + (z.y > 0 ? -z.x : z.x);
    freq += k2 * error;
    phase += k1 * error + freq;
    phase = costas_wrap(phase, pi_f);
    return z;
}

// The PLL over `len` items starting at item `base` (one lane).  Lanes walk different
// segments, so every load instruction touches 64 different cache lines: whole 128-byte
// lines are loaded with 16-byte instructions, one chunk (16 symbols) ahead of the PLL.
// KV: float4 per prefetched chunk.  8 = whole 128-byte lines, the fastest loop by itself; 2 keeps
// k_costas under 48 VGPRs, which is what a SIMD has left beside two correlator waves
// (gr4pm_costas_loop_set_small_footprint; the pipelined receiver asks for it): the kernel alone then
// takes 1.06 instead of 0.63 ms per 2^26 samples, but the pipelined front end gains 2.7 % (the
// stage has the time, the correlator gets its slots back).
template <int CONSTELLATION, int KV = 8>
__device__ __forceinline__ void costas_run(const cf* __restrict__ in, cf* __restrict__ out, size_t base,
                                           unsigned len, float& phase, float& freq, float k1, float k2)
{
    auto step = [&](cf x) -> cf { return costas_step<CONSTELLATION>(x, phase, freq, k1, k2); };
    constexpr int kV = KV;           // float4 per chunk
    constexpr unsigned kC = 2 * kV;  // symbols per chunk
    unsigned j = 0;
    if (((base + j) & 1) && j < len) { // align to 16 bytes
        out[base + j] = step(in[base + j]);
        ++j;
    }
    // two register sets: while the PLL walks one chunk, the loads of the next one are in flight
    const unsigned n_chunks = (len - j) / kC;
    if (n_chunks > 0) {
        const float4* ip = reinterpret_cast<const float4*>(in + base + j);
        float4* op = reinterpret_cast<float4*>(out + base + j);
        float4 a[kV], b[kV];
        auto load = [&](float4(&v)[kV], unsigned c) {
            const float4* p = ip + static_cast<size_t>(min(c, n_chunks - 1)) * kV; // clamped: no branch
#pragma unroll
            for (int u = 0; u < kV; ++u) v[u] = p[u];
            // keep the loads up here: hipcc otherwise sinks them to their first use, or lets the
            // PLL arithmetic overtake them.  The empty asm orders the loads (memory clobber) and
            // makes the PLL state, where every chain of arithmetic starts, depend on it.
            asm volatile("" : "+v"(phase), "+v"(freq) : : "memory");
        };
        auto run = [&](float4(&v)[kV], unsigned c) {
#pragma unroll
            for (int u = 0; u < kV; ++u) {
                const cf z0 = step(cf{ v[u].x, v[u].y });
                const cf z1 = step(cf{ v[u].z, v[u].w });
                v[u] = make_float4(z0.x, z0.y, z1.x, z1.y);
            }
            float4* q = op + static_cast<size_t>(c) * kV;
#pragma unroll
            for (int u = 0; u < kV; ++u) q[u] = v[u];
        };
        load(a, 0);
        unsigned c = 0;
        for (; c + 2 <= n_chunks; c += 2) {
            load(b, c + 1);
            run(a, c);
            load(a, c + 2);
            run(b, c + 1);
        }
        if (c < n_chunks) run(a, c);
        j += n_chunks * kC;
    }
    for (; j < len; ++j) out[base + j] = step(in[base + j]);
}

// One lane per segment (the PLL is serial inside a segment).
template <int CONSTELLATION, int KV = 8>
__global__ void k_costas(const CostasSeg* __restrict__ segs, unsigned n_segs,
                         const CostasState* __restrict__ state, CostasState* __restrict__ state_next,
                         float k1, float k2,
                         const cf* __restrict__ in, cf* __restrict__ out, size_t stride)
{
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    __builtin_amdgcn_s_setprio(GR4PM_SERIAL_PRIO); // a few latency-bound waves among throughput kernels
    const CostasSeg g = segs[s];
    float phase, freq;
    if (g.mode == 0) {
        phase = state[g.channel].phase;
        freq = state[g.channel].freq;
    } else {
        phase = g.phase0;
        freq = 0.0f;
    }
    costas_run<CONSTELLATION, KV>(in, out, static_cast<size_t>(g.channel) * stride + g.start, g.len, phase, freq, k1,
                              k2);
    if (g.last) { // ping-pong: another lane may still have to read `state`
        state_next[g.channel].phase = phase;
        state_next[g.channel].freq = freq;
    }
}

// The same kernel held to 32 VGPRs (amdgpu_num_vgpr counts register PAIRS on gfx90a and later: 16 -> 32; hipcc spills
// 26 - 30 dwords, six scratch accesses per four symbols in the loop), which is what a SIMD has left beside two 240-VGPR
// correlator waves: its waves start beside a correlator workgroup instead of waiting for -- and then keeping -- a CU
// of their own.  Slower by itself, +2.5 % for the pipelined receiver (gr4pm_costas_loop_set_small_footprint(h, 2)).
template <int CONSTELLATION, int KV>
__global__ __attribute__((amdgpu_num_vgpr(16))) void k_costas_cap(const CostasSeg* __restrict__ segs, unsigned n_segs,
                                                               const CostasState* __restrict__ state,
                                                               CostasState* __restrict__ state_next, float k1, float k2,
                                                               const cf* __restrict__ in, cf* __restrict__ out,
                                                               size_t stride)
{
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    __builtin_amdgcn_s_setprio(GR4PM_SERIAL_PRIO);
    const CostasSeg g = segs[s];
    float phase, freq;
    if (g.mode == 0) {
        phase = state[g.channel].phase;
        freq = state[g.channel].freq;
    } else {
        phase = g.phase0;
        freq = 0.0f;
    }
    costas_run<CONSTELLATION, KV>(in, out, static_cast<size_t>(g.channel) * stride + g.start, g.len, phase, freq, k1, k2);
    if (g.last) {
        state_next[g.channel].phase = phase;
        state_next[g.channel].freq = freq;
    }
}

// Tag-driven settings (gr4pm_costas_loop_process_packets): a chain is the run of items between
// two set_phase events; it consists of pieces with their own constellation and loop
// coefficients (syncword: PILOT, header and payload: QPSK with different bandwidths); phase
// and frequency flow from piece to piece.  One lane per chain.
struct CostasPiece {
    unsigned long long start; // first item of the piece in the loop's OUTPUT (= its input stream's index)
    long long in_off;         // its input items are in[start + in_off ...]: 0, or the gather of the block in front folded in
    unsigned len;
    int constellation;
    float k1, k2;
    unsigned pad;
};
struct CostasChain {
    unsigned piece0, n_pieces;
    int mode; // 0 continue from the carried state, 1 set_phase(phase0)
    int last;
    float phase0;
    unsigned pad;
};
template <int KV>
__device__ __forceinline__ void costas_chains_body(const CostasChain* __restrict__ chains, unsigned n_chains,
                                                   const CostasPiece* __restrict__ pieces,
                                                   const CostasState* __restrict__ state,
                                                   CostasState* __restrict__ state_next, const cf* __restrict__ in,
                                                   cf* __restrict__ out)
{
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_chains) return;
    __builtin_amdgcn_s_setprio(GR4PM_SERIAL_PRIO);
    const CostasChain ch = chains[s];
    float phase, freq;
    if (ch.mode == 0) {
        phase = state[0].phase;
        freq = state[0].freq;
    } else {
        phase = ch.phase0;
        freq = 0.0f;
    }
    for (unsigned q = 0; q < ch.n_pieces; ++q) {
        const CostasPiece pc = pieces[ch.piece0 + q];
        const cf* src = in + pc.in_off;
        if (pc.constellation == 0) costas_run<0, KV>(src, out, pc.start, pc.len, phase, freq, pc.k1, pc.k2);
        else if (pc.constellation == 1) costas_run<1, KV>(src, out, pc.start, pc.len, phase, freq, pc.k1, pc.k2);
        else costas_run<2, KV>(src, out, pc.start, pc.len, phase, freq, pc.k1, pc.k2);
    }
    if (ch.last) {
        state_next[0].phase = phase;
        state_next[0].freq = freq;
    }
}
template <int KV>
__global__ void k_costas_chains(const CostasChain* __restrict__ chains, unsigned n_chains,
                                const CostasPiece* __restrict__ pieces, const CostasState* __restrict__ state,
                                CostasState* __restrict__ state_next, const cf* __restrict__ in,
                                cf* __restrict__ out)
{
    costas_chains_body<KV>(chains, n_chains, pieces, state, state_next, in, out);
}
// The same chains held to 32 VGPRs, as k_costas_cap is (round 6): the decode_headers / soft_bits receivers' PLL -- 121
// VGPRs in the form above -- could not start beside a correlator workgroup (2 x 240 of a SIMD's 512 registers): each of
// its one-wave workgroups (one per 64 packets) waited for a compute unit and then kept a correlator workgroup off it for
// as long as a packet's chain takes.
__global__ __attribute__((amdgpu_num_vgpr(16))) void k_costas_chains_cap(const CostasChain* __restrict__ chains, unsigned n_chains,
                                                                      const CostasPiece* __restrict__ pieces,
                                                                      const CostasState* __restrict__ state,
                                                                      CostasState* __restrict__ state_next,
                                                                      const cf* __restrict__ in, cf* __restrict__ out)
{
    costas_chains_body<2>(chains, n_chains, pieces, state, state_next, in, out);
}

// =====================================================================================
// SyncwordWipeoff (syncword_wipeoff.hpp:66-82): copy, then x[pos] *= syncword[pos] on spans
// =====================================================================================
struct WipeSpan {
    unsigned long long start; // first item of the span inside this call
    unsigned first;           // first syncword position
    unsigned len;
};
__global__ void k_wipe(const WipeSpan* __restrict__ spans, const float* __restrict__ syncword,
                       const cf* in, cf* out) // may be the same buffer
{
    const WipeSpan w = spans[blockIdx.x];
    for (unsigned i = threadIdx.x; i < w.len; i += blockDim.x)
        out[w.start + i] = fmulc(syncword[w.first + i], in[w.start + i]);
}

__global__ void k_sincosf(const float* __restrict__ x, size_t n, float* __restrict__ sn, float* __restrict__ cs)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) sincosf_glibc(x[i], sn + i, cs + i);
}

__global__ void k_costas_wrap(const float* __restrict__ x, size_t n, float* __restrict__ out)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = costas_wrap(x[i], 3.14159265358979323846f);
}

template <typename T>
__global__ void k_copy(const T* __restrict__ in, T* __restrict__ out, size_t n)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        out[i] = in[i];
}

// =====================================================================================
// FIR family.  x(i) for i < 0 comes from the carried history (last `cap` items of the
// previous calls, zero at start: the reference pre-fills its HistoryBuffer with zeros).
// =====================================================================================
template <typename T>
__device__ __forceinline__ T item_at(const T* cur, const T* carry, unsigned cap, long long i)
{
    return i >= 0 ? cur[i] : carry[static_cast<long long>(cap) + i];
}
__device__ __forceinline__ cf mac(cf acc, float t, cf x) { return cadd(acc, fmulc(t, x)); }
__device__ __forceinline__ float mac(float acc, float t, float x) { return acc + t * x; }
__device__ __forceinline__ cf scale_item(float s, cf v) { return fmulc(s, v); }
__device__ __forceinline__ float scale_item(float s, float v) { return s * v; }
__device__ __forceinline__ cf zero_item(cf) { return { 0.f, 0.f }; }
__device__ __forceinline__ float zero_item(float) { return 0.f; }

template <typename T>
__global__ void k_update_hist(const T* __restrict__ in, const T* __restrict__ carry,
                              T* __restrict__ carry_next, unsigned cap, size_t n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    carry_next[i] = item_at(in, carry, cap, static_cast<long long>(n) - cap + i);
}

// InterpolatingFirFilter::processBulk (interpolating_fir_filter.hpp:93-99): taps laid out
// [arm][arm_stride]; out[n*L + j] = sum_m arm_j[m] * x[n - m], m ascending, acc from 0.
// One thread per INPUT item: a workgroup stages its 256 items and the arm_stride - 1 before them in LDS once
// (coalesced), every thread then forms its L outputs from LDS (neighbouring lanes read neighbouring items, the taps
// are broadcast reads) and writes them as one contiguous run of L items.  Round 1 had one thread per OUTPUT: every
// item was fetched L * arm length times through L1 and every output paid two 64-bit divisions (542 us per 2^24
// symbols in, 2^26 samples out; this form: see HISTORY.md section 5).
constexpr unsigned kFirItems = 256;
constexpr size_t kFirMaxSmem = 160 * 1024;
inline size_t interp_fir_smem(size_t L, size_t arm_stride, size_t item_size)
{
    return ((kFirItems + arm_stride) * item_size + 15) / 16 * 16 + L * arm_stride * sizeof(float) + L * sizeof(unsigned);
}
template <typename T>
__global__ __launch_bounds__(kFirItems) void k_interp_fir(const T* __restrict__ in, const T* __restrict__ carry,
                                                          unsigned cap, const float* __restrict__ taps,
                                                          const unsigned* __restrict__ arm_len, unsigned arm_stride,
                                                          unsigned L, size_t n_in, T* __restrict__ out)
{
    // LDS: the item tile first (16-byte aligned whatever L * arm_stride is: complex items are read and written as
    // 64-bit words), then the taps, then the arm lengths; interp_fir_smem() is the host's copy of this layout
    extern __shared__ float4 s_fir[];
    T* tile = reinterpret_cast<T*>(s_fir); // tile[i] = x[n0 - (arm_stride - 1) + i]
    float* s_taps = reinterpret_cast<float*>(s_fir + ((kFirItems + arm_stride) * sizeof(T) + 15u) / 16u);
    unsigned* s_len = reinterpret_cast<unsigned*>(s_taps + L * arm_stride);
    for (unsigned i = threadIdx.x; i < L * arm_stride; i += kFirItems) s_taps[i] = taps[i];
    for (unsigned i = threadIdx.x; i < L; i += kFirItems) s_len[i] = arm_len[i];
    const unsigned hist = arm_stride - 1;
    for (size_t n0 = static_cast<size_t>(blockIdx.x) * kFirItems; n0 < n_in; n0 += static_cast<size_t>(gridDim.x) * kFirItems) {
        __syncthreads(); // taps staged / the tile of the round before is no longer read
        const unsigned count = static_cast<unsigned>(min(static_cast<size_t>(kFirItems), n_in - n0));
        for (unsigned i = threadIdx.x; i < count + hist; i += kFirItems)
            tile[i] = item_at(in, carry, cap, static_cast<long long>(n0) + i - hist);
        __syncthreads();
        if (threadIdx.x < count && L == 4) {
            // the usual interpolation: every item is read from LDS once for the four arms, the four outputs leave as
            // one 32-byte (complex) or 16-byte (float) run.  Per arm the sum still runs over m ascending.
            const T* x = tile + hist + threadIdx.x;
            const unsigned l0 = s_len[0], l1 = s_len[1], l2 = s_len[2], l3 = s_len[3];
            T a0 = zero_item(T{}), a1 = a0, a2 = a0, a3 = a0;
            for (unsigned m = 0; m < arm_stride; ++m) {
                const T v = *(x - m);
                if (m < l0) a0 = mac(a0, s_taps[m], v);
                if (m < l1) a1 = mac(a1, s_taps[arm_stride + m], v);
                if (m < l2) a2 = mac(a2, s_taps[2 * arm_stride + m], v);
                if (m < l3) a3 = mac(a3, s_taps[3 * arm_stride + m], v);
            }
            T* o = out + (n0 + threadIdx.x) * 4;
            if constexpr (sizeof(T) == sizeof(cf)) {
                if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) { // two 16-byte stores per lane
                    float4* o4 = reinterpret_cast<float4*>(o);
                    o4[0] = make_float4(a0.x, a0.y, a1.x, a1.y);
                    o4[1] = make_float4(a2.x, a2.y, a3.x, a3.y);
                } else {
                    o[0] = a0, o[1] = a1, o[2] = a2, o[3] = a3;
                }
            } else {
                o[0] = a0, o[1] = a1, o[2] = a2, o[3] = a3;
            }
        } else if (threadIdx.x < count) {
            const T* x = tile + hist + threadIdx.x; // x[-m] = item n - m
            T* o = out + (n0 + threadIdx.x) * L;
            for (unsigned j = 0; j < L; ++j) {
                const float* arm = s_taps + j * arm_stride;
                const unsigned len = s_len[j];
                T acc = zero_item(T{});
                for (unsigned m = 0; m < len; ++m) acc = mac(acc, arm[m], *(x - m));
                o[j] = acc;
            }
        }
    }
}

// SymbolFilter (symbol_filter.hpp:208-214): y = scale * sum_m arm[m] * x[idx - m]
using hostlogic::SymRun; // hostlogic/symbol_filter_replay.hpp
#ifndef GR4PM_SYM_PER_WG
#define GR4PM_SYM_PER_WG 256
#endif
constexpr unsigned kSymPerWg = GR4PM_SYM_PER_WG; // output symbols (= threads) per workgroup
// One workgroup = up to kSymPerWg consecutive output symbols of ONE run.  The inputs of those symbols
// are one contiguous span (255*sps + arm_size items): it is staged into LDS with coalesced
// loads and every thread then reads its arm_size items from LDS (the MAC order of the
// reference, std::inner_product, m ascending, is kept: bit-exact).
// Everything a workgroup needs is resolved once by k_symf_wg_plan (a binary search inside the
// filter kernel would put ~12 dependent L2 round trips in front of each workgroup).
struct SymWg {
    long long lo_item; // oldest input item of the span (negative: inside the carried history)
    unsigned o0;       // first output symbol
    unsigned count;    // symbols (<= 256)
    unsigned arm;
    float scale;
    unsigned seg;      // fused CFC: segment of max(lo_item, 0)
    unsigned chan;     // index into the SymChan table (launches that span channels)
    // what the fused CFC needs of segment `seg` (RotSeg start / len / ck0, the increment and the counter that
    // k_rot_checkpoints left for it): one scalar load of this entry instead of three dependent ones in front of every
    // workgroup's first vector load.  Spans that run into further segments read those from the tables.
    unsigned long long seg_start, seg_len;
    unsigned seg_ck0, seg_counter0;
    cf seg_incr;
};
static_assert(sizeof(SymWg) == 64, "one 64-byte scalar load per workgroup");
// Fused CoarseFrequencyCorrection: when the symbol filter is fed by a CFC block, the rotation
// is applied while the filter stages its input (the rotated stream is never written to HBM).
struct CfcDev {
    const RotSeg* segs;   // this call's segments of channel 0, sorted by start, tiling [0, n_in)
    const cf* ck;         // phasor checkpoints every kRotChunk items
    const cf* seg_incr;
    const unsigned* seg_counter0;
    unsigned n_segs;
};
// One launch for every channel of a batch (gr4pm_cfc_symbol_filter_run_channels): what the single-channel launch
// passes as kernel arguments comes from a table, indexed by the channel of the workgroup's run.
struct SymChan {
    const cf* in;
    const cf* carry; // history of the channel's SymbolFilter (rotated items)
    cf* carry_next;
    cf* out;
    const RotSeg* segs; // the channel's share of the CFC plan
    const cf* seg_incr;
    const unsigned* seg_counter0;
    unsigned n_segs;
    unsigned pad;
    unsigned long long n; // items consumed by the channel in this call (history update)
    // two-piece input: items [0, n_head) of the call live at head[], item i >= n_head at in[i - n_head] (the
    // detector's delayed stream read in place: the tail of the batch before + this batch's input)
    const cf* head;
    unsigned long long n_head;
};
__device__ __forceinline__ CfcDev chan_cfc(const SymChan& c, const cf* ck)
{
    CfcDev f;
    f.segs = c.segs;
    f.ck = ck;
    f.seg_incr = c.seg_incr;
    f.seg_counter0 = c.seg_counter0;
    f.n_segs = c.n_segs;
    return f;
}
__device__ __forceinline__ unsigned cfc_find_seg(const CfcDev& f, long long idx)
{
    unsigned lo = 0, hi = f.n_segs - 1;
    const unsigned long long u = idx < 0 ? 0ull : static_cast<unsigned long long>(idx);
    while (lo < hi) {
        const unsigned mid = (lo + hi + 1) >> 1;
        if (f.segs[mid].start <= u) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// rotated item idx (>= 0) of the current call: the lane replays at most kRotChunk-1 steps of
// the recurrence from its chunk's checkpoint
__device__ __forceinline__ cf cfc_phasor(const CfcDev& f, long long idx, unsigned seg)
{
    const unsigned long long j = static_cast<unsigned long long>(idx) - f.segs[seg].start;
    const unsigned long long c = j / kRotChunk;
    cf e = f.ck[f.segs[seg].ck0 + c];
    const cf inc = f.seg_incr[seg];
    unsigned counter = f.seg_counter0[seg] + static_cast<unsigned>(c * kRotChunk);
    const unsigned steps = static_cast<unsigned>(j - c * kRotChunk);
    for (unsigned t = 0; t < steps; ++t) rot_step(e, inc, counter);
    return e;
}
__device__ __forceinline__ cf cfc_item(const CfcDev& f, const cf* in, long long idx, unsigned seg)
{
    return cmul(in[idx], cfc_phasor(f, idx, seg)); // coarse_frequency_correction.hpp:87
}
// history after a fused call: last cap ROTATED items
__global__ void k_update_hist_cfc(const cf* __restrict__ in, const cf* __restrict__ carry, cf* __restrict__ carry_next,
                                  unsigned cap, size_t n, CfcDev f)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const long long idx = static_cast<long long>(n) - cap + i;
    carry_next[i] = idx < 0 ? carry[static_cast<long long>(cap) + idx] // history is stored rotated
                            : cfc_item(f, in, idx, cfc_find_seg(f, idx));
}

// the same for every channel of a batched launch (blockIdx.y = channel)
__global__ void k_update_hist_cfc_channels(const SymChan* __restrict__ chans, const cf* __restrict__ ck, unsigned cap)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const SymChan c = chans[blockIdx.y];
    if (c.n == 0) return;
    const long long idx = static_cast<long long>(c.n) - cap + i;
    const CfcDev f = chan_cfc(c, ck);
    if (idx < 0) {
        c.carry_next[i] = c.carry[static_cast<long long>(cap) + idx];
        return;
    }
    const cf x = static_cast<unsigned long long>(idx) < c.n_head ? c.head[idx] : c.in[idx - static_cast<long long>(c.n_head)];
    c.carry_next[i] = cmul(x, cfc_phasor(f, idx, cfc_find_seg(f, idx)));
}

__global__ void k_symf_wg_plan(const SymRun* __restrict__ runs, unsigned n_runs, unsigned n_wg, unsigned sps,
                               unsigned arm_size, CfcDev cfc, SymWg* __restrict__ plan,
                               const SymChan* __restrict__ chans, unsigned sym_per_wg)
{
    const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_wg) return;
    unsigned lo = 0, hi = n_runs - 1;
    while (lo < hi) {
        const unsigned mid = (lo + hi + 1) >> 1;
        if (runs[mid].wg0 <= w) lo = mid;
        else hi = mid - 1;
    }
    const SymRun r = runs[lo];
    const unsigned first = (w - r.wg0) * sym_per_wg;
    SymWg p;
    p.o0 = r.out0 + first;
    p.count = min(sym_per_wg, r.count - first);
    p.lo_item = r.in0 + static_cast<long long>(first) * sps - (arm_size - 1);
    p.arm = r.arm;
    p.scale = r.scale;
    if (chans) cfc = chan_cfc(chans[r.chan], cfc.ck);
    p.seg = cfc.n_segs ? cfc_find_seg(cfc, p.lo_item) : 0u;
    p.chan = r.chan;
    p.seg_start = p.seg_len = 0;
    p.seg_ck0 = p.seg_counter0 = 0;
    p.seg_incr = cf{ 0.f, 0.f };
    if (cfc.n_segs) {
        const RotSeg g = cfc.segs[p.seg];
        p.seg_start = g.start, p.seg_len = g.len, p.seg_ck0 = g.ck0;
        p.seg_incr = cfc.seg_incr[p.seg];
        p.seg_counter0 = cfc.seg_counter0[p.seg];
    }
    plan[w] = p;
}

// Fused CFC: fills the workgroup's tile (phase-major: item i at (i % sps) * pitch + i / sps) with the ROTATED items of
// the span [p.lo_item, p.lo_item + span).
//  1. history items are stored rotated: straight to the tile
//  2. one lane per checkpoint chunk: the lane reads its kRotChunk raw items (64 contiguous bytes) straight from global
//     memory, replays the phasor recurrence of the chunk once (kRotChunk-1 steps for kRotChunk items) and writes the
//     rotated items to the tile; segment by segment (usually one).  The raw items are not staged in LDS (round 1 did):
//     half the LDS footprint, one barrier and one global-memory latency less per workgroup.
// ORIGIN (a multiple of sps and of kRotChunk): item i of the span lives at tile index i + ORIGIN, and a chunk that
// only PARTLY overlaps the span is written whole, its other items into the ORIGIN entries in front of the span or the
// kRotChunk behind it -- as long as the whole chunk lies inside its segment (then its raw items exist).  With
// ORIGIN = 0 a partly overlapping chunk goes item by item through the path at the bottom (about 220 instructions for
// its whole wave): the first and the last chunk of almost every span, i.e. both waves of every workgroup of the
// receiver's filter.
template <unsigned THREADS, unsigned ORIGIN = 0>
__device__ __forceinline__ void cfc_fill_tile(const SymWg& p, unsigned span, unsigned sps, unsigned pitch, cf* tile,
                                              const cf* __restrict__ in, const cf* __restrict__ carry, unsigned cap,
                                              const CfcDev& cfc, const cf* __restrict__ head = nullptr,
                                              long long n_head = 0)
{
    if (p.lo_item < 0)
        for (unsigned i = threadIdx.x; i < span && p.lo_item + i < 0; i += THREADS)
            tile[((i + ORIGIN) % sps) * pitch + (i + ORIGIN) / sps] = carry[static_cast<long long>(cap) + p.lo_item + i];
    const long long lo = p.lo_item < 0 ? 0 : p.lo_item;
    const long long hi = p.lo_item + span;
    for (unsigned sg = p.seg; sg < cfc.n_segs; ++sg) {
        RotSeg g;
        cf inc;
        unsigned c0;
        if (sg == p.seg) { // (uniform) the usual case, and the only segment of most spans: everything is in the plan entry
            g.start = p.seg_start, g.len = p.seg_len, g.ck0 = p.seg_ck0;
            inc = p.seg_incr, c0 = p.seg_counter0;
        } else {
            g = cfc.segs[sg];
            inc = cfc.seg_incr[sg], c0 = cfc.seg_counter0[sg];
        }
        const long long a = max(static_cast<long long>(g.start), lo);
        const long long b = min(static_cast<long long>(g.start + g.len), hi);
        if (a < b) {
            const unsigned long long c_first = static_cast<unsigned long long>(a - g.start) / kRotChunk;
            const unsigned n_chunks =
                static_cast<unsigned>(static_cast<unsigned long long>(b - 1 - g.start) / kRotChunk - c_first) + 1;
            for (unsigned ch = threadIdx.x; ch < n_chunks; ch += THREADS) {
                const unsigned long long c = c_first + ch;
                cf e = cfc.ck[g.ck0 + c];
                unsigned counter = c0 + static_cast<unsigned>(c * kRotChunk);
                const long long idx0 = static_cast<long long>(g.start + c * kRotChunk);
                const bool whole = ORIGIN ? idx0 + static_cast<long long>(kRotChunk) <= static_cast<long long>(g.start + g.len)
                                          : idx0 >= a && idx0 + static_cast<long long>(kRotChunk) <= b;
                if (whole && (counter & 511u) <= 512u - kRotChunk &&
                    (idx0 >= n_head || idx0 + static_cast<long long>(kRotChunk) <= n_head)) {
                    // whole chunk inside the span or (ORIGIN) at least inside the segment, on one side of a two-piece
                    // input, and no renormalisation among its 7 steps (the usual case): straight-line packed arithmetic,
                    // same operations as below.  (With ORIGIN the index can start up to kRotChunk - 1 in front of the span.)
                    const unsigned i0 = static_cast<unsigned>(idx0 - p.lo_item + static_cast<long long>(ORIGIN));
                    // (explicitly a global-memory pointer: in the multi-channel launch `in` / `head` are loaded from the
                    // channel table, which makes them generic pointers and these loads flat_load -- both address paths,
                    // both wait counters -- instead of global_load)
                    typedef const float __attribute__((address_space(1))) * gflt;
                    const gflt src = (gflt)(idx0 >= n_head ? in + (idx0 - n_head) : head + idx0);
                    cf x[kRotChunk];
#pragma unroll
                    for (unsigned t = 0; t < kRotChunk; ++t) x[t] = cf{ src[2 * t], src[2 * t + 1] };
                    rot8_pk(x, e, inc); // hpp:87
                    if (sps == 4) { // items t and t + 4 share their phase row and sit side by side: four addresses, not eight
#pragma unroll
                        for (unsigned t = 0; t < 4; ++t) {
                            const unsigned i = i0 + t;
                            cf* q = tile + (i % 4) * pitch + i / 4;
                            q[0] = x[t];
                            q[1] = x[t + 4];
                        }
                    } else {
#pragma unroll
                        for (unsigned t = 0; t < kRotChunk; ++t) {
                            const unsigned i = i0 + t;
                            tile[(i % sps) * pitch + i / sps] = x[t];
                        }
                    }
                    continue;
                }
#pragma unroll
                for (unsigned t = 0; t < kRotChunk; ++t) {
                    const long long idx = idx0 + t;
                    if (idx >= a && idx < b) {
                        const unsigned i = static_cast<unsigned>(idx - p.lo_item) + ORIGIN;
                        const cf x = idx < n_head ? head[idx] : in[idx - n_head];
                        tile[(i % sps) * pitch + i / sps] = cmul(x, e); // hpp:87
                    }
                    if (t + 1 < kRotChunk) rot_step(e, inc, counter);
                }
            }
        }
        if (static_cast<long long>(g.start + g.len) >= hi) break;
    }
}

// SPS > 0: samples_per_symbol known at compile time (divisions become shifts / constants);
// SPS == 0: run-time value.
template <typename T, int SPS, bool CFC>
__global__ __launch_bounds__(kSymPerWg) void k_symbol_filter(const T* __restrict__ in, const T* __restrict__ carry,
                                                       unsigned cap, const float* __restrict__ taps,
                                                       unsigned arm_size, unsigned sps_rt,
                                                       const SymWg* __restrict__ plan, T* __restrict__ out,
                                                       CfcDev cfc, const SymChan* __restrict__ chans)
{
    extern __shared__ unsigned char s_raw[];
    T* tile = reinterpret_cast<T*>(s_raw);
    const unsigned sps = SPS > 0 ? static_cast<unsigned>(SPS) : sps_rt;
    const SymWg p = plan[blockIdx.x];
    const cf* head = nullptr;
    long long n_head = 0;
    if constexpr (CFC) {
        if (chans) { // a launch that spans channels: the workgroup's channel supplies the pointers
            const SymChan c = chans[p.chan];
            in = c.in;
            carry = c.carry;
            out = c.out;
            cfc = chan_cfc(c, cfc.ck);
            head = c.head;
            n_head = static_cast<long long>(c.n_head);
        }
    }
    const unsigned span = (p.count - 1) * sps + arm_size;
    // tile is stored phase-major: item i lives at (i % sps) * pitch + i / sps, so that for a
    // fixed tap m the 64 lanes (items sps apart) read consecutive LDS words
    const unsigned pitch = (kSymPerWg * sps + arm_size) / sps + 2;
    // the arm is the same for the whole workgroup: its taps go to LDS (broadcast reads)
    float* s_arm = reinterpret_cast<float*>(tile + pitch * sps);
    for (unsigned m = threadIdx.x; m < arm_size; m += kSymPerWg) s_arm[m] = taps[static_cast<size_t>(p.arm) * arm_size + m];
    if constexpr (CFC) {
        cfc_fill_tile<kSymPerWg>(p, span, sps, pitch, tile, in, carry, cap, cfc, head, n_head);
    } else {
        for (unsigned i = threadIdx.x; i < span; i += kSymPerWg)
            tile[(i % sps) * pitch + i / sps] = item_at(in, carry, cap, p.lo_item + i);
    }
    __syncthreads();
    if (threadIdx.x < p.count) {
        // tile index of tap m of this symbol: tid * sps + j with j = arm_size - 1 - m;
        // j % sps and j / sps are the same for every thread
        T acc = zero_item(T{});
        for (unsigned m = 0; m < arm_size; ++m) {
            const unsigned j = arm_size - 1 - m;
            acc = mac(acc, s_arm[m], tile[(j % sps) * pitch + j / sps + threadIdx.x]);
        }
        out[p.o0 + threadIdx.x] = scale_item(p.scale, acc);
    }
}

// The receiver's design (4 samples per symbol, 44-tap arms, fused CFC) as its own kernel.  Measured on the kernel above
// (343 us per 2^26 samples): no MACs -76 us, no rotation -31, no item loads -63, empty workgroups 26 -- VALU (~0.10 ms),
// the LDS pipe (~0.15 ms: 44 8-byte tile reads and 44 tap reads per symbol) and HBM (0.13 ms) add up instead of
// overlapping.  Here the LDS traffic of the MACs is 45 % of that:
//   * a lane computes TWO neighbouring symbols; they share 40 of their 44 items, and one 16-byte read delivers the two
//     tile entries (phase row, items 2l + 2k and 2l + 2k + 1) that the pair needs at one tap position: 24 ds_read_b128
//     per two symbols instead of 88 ds_read_b64
//   * the 44 taps of the workgroup's arm are uniform: scalar loads into SGPRs, no LDS copy, no tap reads
//   * 240 symbols per workgroup of 128 threads: the span is 1000 items = at most 126 checkpoint chunks, one pass of the
//     rotation phase with 98 % of the lanes busy (256 symbols: 134 chunks on 256 lanes)
// MAC order as in the reference (std::inner_product, tap index ascending), every product and sum rounded once: bit-exact.
constexpr unsigned kFastSym = 240, kFastThreads = 128, kFastArm = 44, kFastSps = 4;
constexpr unsigned kFastOrigin = 8; // tile index of the span's first item: room for a chunk that starts in front of it
constexpr unsigned kFastPitch = 256; // entries per phase row: (8 + 1000 + 7) / 4 = 254 used, 8 KiB per tile
static_assert(kFastPitch % 2 == 0 && kFastPitch * kFastSps >= kFastOrigin + (kFastSym - 1) * kFastSps + kFastArm + kRotChunk - 1 &&
                  kFastOrigin % 8 == 0,
              "16-byte reads need even rows; the margins of partly overlapping chunks need room");
// ABL (timing only, wrong results; GR4PM_SYMF_ABL): 1 = no MAC phase, 2 = no tile fill (no item loads, no rotation)
// (second launch bound: eight waves per SIMD, i.e. at most 64 VGPRs -- hipcc takes 80 when left alone, and the kernel is
// bound by the workgroups a CU holds: 16 instead of 12)
// Everything uniform stays on the scalar unit (round 5: 218.5 M -> 166.5 M vector wave-instructions per 2^28 samples, 391 ->
// 298 per wave and tile of which 176 are the multiply-adds; 558 -> 521 us alone = 5.2 TB/s of input + output):
//   * the channel's pointers come from ONE table entry through scalar loads -- the single-channel launch passes its entry
//     by value as the FIRST kernel argument and the kernel reads it where it lies in the kernarg segment, so both launch
//     forms share one code path and nothing of it stays live across the multiply-adds (the 44 taps need 44 SGPRs there;
//     with separate in / carry / out / CfcDev arguments hipcc spilled ~65 SGPRs per tile through v_writelane / v_readlane)
//   * further segments of a span are read through constant-address-space pointers (s_load; hipcc reads them with
//     flat_load when it cannot prove the tables unwritten, which made every index of the fill a 64-bit vector value)
//   * per lane only ch * 8 varies: chunk c_first + ch of the segment; checkpoint, items and tile slots are a uniform
//     base plus that, the four tile addresses of a chunk are (uniform phase row and column of item t) + 16 ch bytes
// Bit-exact as before: same packed products and sums (rot8_pk), same item-by-item path for the chunks that overlap a
// segment end, the head / input seam of a two-piece input, or a renormalisation.
typedef const unsigned long long __attribute__((address_space(4)))* symf_cu64;
typedef const unsigned __attribute__((address_space(4)))* symf_cu32;
typedef const float __attribute__((address_space(4)))* symf_cf32;
static_assert(sizeof(SymChan) == 88 && offsetof(SymChan, segs) == 32 && offsetof(SymChan, n_segs) == 56 &&
                  offsetof(SymChan, head) == 72 && offsetof(SymChan, n_head) == 80 && sizeof(RotSeg) == 48 &&
                  offsetof(RotSeg, ck0) == 20,
              "k_symbol_filter_fast reads these tables word by word");
template <int ABL, bool LOOP>
__global__ __launch_bounds__(kFastThreads, 8) void k_symbol_filter_fast(SymChan one, const SymChan* __restrict__ chans,
                                                                     const SymWg* __restrict__ plan,
                                                                     const float* __restrict__ taps,
                                                                     const cf* __restrict__ ck, unsigned cap, unsigned n_wg,
                                                                     unsigned tiles)
{
    (void)one; // read in place: the first 88 bytes of the kernarg segment
    __shared__ __attribute__((aligned(16))) cf tile[kFastSps * kFastPitch];
    // LOOP (GR4PM_SYMF_TILES > 1; the default is one tile per workgroup since round 5): `tiles` consecutive plan entries
    // per workgroup, the entry of the NEXT tile requested (one 64-byte scalar load) while this tile's results are stored.
    // Round 3 ran two tiles per workgroup with the next entry requested in FRONT of the tile's work; its 16 SGPRs do
    // not fit beside the 44 taps (80 SGPRs at eight waves per SIMD) and the loop-carried state cost ~50 v_writelane /
    // v_readlane per tile.  Measured with this kernel: 521 us with one tile, 538 with two (184.3 M instructions).
    unsigned w = blockIdx.x * tiles;
    const unsigned w_end = LOOP ? min(w + tiles, n_wg) : w + 1; // (!LOOP: one tile, tiles == 1)
    SymWg p = plan[w];
    for (; w < w_end; ++w) {
    const unsigned span = (p.count - 1) * kFastSps + kFastArm;
    cf* out;
    {
    const symf_cu64 cw = chans ? (symf_cu64)(chans + p.chan) : (symf_cu64)__builtin_amdgcn_kernarg_segment_ptr();
    out = reinterpret_cast<cf*>(cw[3]);
    if (!(ABL & 2)) {
    const cf* in = reinterpret_cast<const cf*>(cw[0]);
    const cf* head = reinterpret_cast<const cf*>(cw[9]);
    const long long n_head = static_cast<long long>(cw[10]);
    const unsigned n_segs = static_cast<unsigned>(cw[7]);
    if (p.lo_item < 0) { // the carried history is stored rotated: straight to the tile
        const cf* carry = reinterpret_cast<const cf*>(cw[1]);
        for (unsigned i = threadIdx.x; i < span && p.lo_item + i < 0; i += kFastThreads)
            tile[((i + kFastOrigin) % kFastSps) * kFastPitch + (i + kFastOrigin) / kFastSps] =
                carry[static_cast<long long>(cap) + p.lo_item + i];
    }
    const long long lo = p.lo_item < 0 ? 0 : p.lo_item;
    const long long hi = p.lo_item + span;
    for (unsigned sg = p.seg; sg < n_segs; ++sg) {
        unsigned long long g_start, g_len;
        unsigned g_ck0, c0;
        cf inc;
        if (sg == p.seg) { // the usual case, and the only segment of most spans: everything is in the plan entry
            g_start = p.seg_start, g_len = p.seg_len, g_ck0 = p.seg_ck0;
            inc = p.seg_incr, c0 = p.seg_counter0;
        } else {
            const symf_cu64 gw = (symf_cu64)(reinterpret_cast<const RotSeg*>(cw[4]) + sg);
            g_start = gw[0], g_len = gw[1];
            g_ck0 = ((symf_cu32)gw)[5];
            const symf_cf32 iw = (symf_cf32)(reinterpret_cast<const cf*>(cw[5]) + sg);
            inc = cf{ iw[0], iw[1] };
            c0 = ((symf_cu32)cw[6])[sg];
        }
        const long long g_end = static_cast<long long>(g_start + g_len);
        const long long a = max(static_cast<long long>(g_start), lo);
        const long long b = min(g_end, hi);
        if (a < b) {
            const unsigned long long c_first = static_cast<unsigned long long>(a - g_start) / kRotChunk;
            const unsigned n_chunks =
                static_cast<unsigned>(static_cast<unsigned long long>(b - 1 - g_start) / kRotChunk - c_first) + 1;
            // (uniform) chunk ch of this pass: items first + 8 ch ..., checkpoint ck_s[ch], counter counter_s + 8 ch,
            // tile slots from i0_s + 8 ch on
            const long long first = static_cast<long long>(g_start + c_first * kRotChunk);
            const cf* ck_s = ck + (g_ck0 + c_first);
            const unsigned counter_s = c0 + static_cast<unsigned>(c_first * kRotChunk);
            const unsigned i0_s = static_cast<unsigned>(first - p.lo_item + static_cast<long long>(kFastOrigin));
            // items of the segment from `first` on (whole chunks only on the straight path)
            const unsigned room = static_cast<unsigned>(min(g_end - first, static_cast<long long>(0x7fffffff)));
            // every chunk of the pass on one side of a two-piece input's seam?  (else: item by item)
            const bool in_side = first >= n_head;
            const bool one_side = in_side || first + static_cast<long long>(n_chunks * kRotChunk) <= n_head;
            typedef const float __attribute__((address_space(1))) * gflt;
            const gflt src_s = (gflt)(in_side ? in + (first - n_head) : head + first);
            for (unsigned ch = threadIdx.x; ch < n_chunks; ch += kFastThreads) {
                const unsigned t8 = ch * kRotChunk;
                cf e = ck_s[ch];
                unsigned counter = counter_s + t8;
                if (one_side && t8 + kRotChunk <= room && (counter & 511u) <= 512u - kRotChunk) {
                    const gflt src = src_s + 2 * t8;
                    cf x[kRotChunk];
#pragma unroll
                    for (unsigned t = 0; t < kRotChunk; ++t) x[t] = cf{ src[2 * t], src[2 * t + 1] };
                    rot8_pk(x, e, inc); // hpp:87
                    // items t and t + 4 share their phase row and sit side by side: four addresses, not eight
#pragma unroll
                    for (unsigned t = 0; t < 4; ++t) {
                        const unsigned it = i0_s + t; // (uniform)
                        cf* q = tile + ((it % kFastSps) * kFastPitch + it / kFastSps) + 2 * ch;
                        q[0] = x[t];
                        q[1] = x[t + 4];
                    }
                    continue;
                }
                const long long idx0 = first + t8;
#pragma unroll
                for (unsigned t = 0; t < kRotChunk; ++t) {
                    const long long idx = idx0 + t;
                    if (idx >= a && idx < b) {
                        const unsigned i = static_cast<unsigned>(idx - p.lo_item) + kFastOrigin;
                        const cf x = idx < n_head ? head[idx] : in[idx - n_head];
                        tile[(i % kFastSps) * kFastPitch + i / kFastSps] = cmul(x, e); // hpp:87
                    }
                    if (t + 1 < kRotChunk) rot_step(e, inc, counter);
                }
            }
        }
        if (g_end >= hi) break;
    }
    }
    }
    const float* __restrict__ tp = taps + static_cast<size_t>(p.arm) * kFastArm; // uniform: scalar loads
    float tap[kFastArm];
#pragma unroll
    for (unsigned m = 0; m < kFastArm; ++m) tap[m] = tp[m];
    __syncthreads();
    const unsigned l = threadIdx.x;
    if (2 * l < p.count) {
    if (ABL & 1) {
        out[p.o0 + 2 * l] = cf{ tap[0], tap[1] };
        if (2 * l + 1 < p.count) out[p.o0 + 2 * l + 1] = cf{ tap[2], tap[3] };
    } else {
    // pair(ph, k) = tile entries (ph, 2l + 2k) and (ph, 2l + 2k + 1); symbol A = 2l uses entry 2l + q at tap position
    // q = j / 4 of phase ph = j % 4 (j = 43 - m), symbol B = 2l + 1 uses entry 2l + 1 + q
    const float4* rows = reinterpret_cast<const float4*>(tile) + l + kFastOrigin / (2 * kFastSps); // two entries per float4
    constexpr unsigned kRow4 = kFastPitch / 2; // float4 per phase row
    // The MACs as packed FP32: v_pk_mul_f32 (re, im) x (tap, tap) -- the tap broadcast from one half of an SGPR pair by
    // op_sel -- and v_pk_add_f32 onto the accumulator: every product and every sum still rounded once, in the
    // reference's order, in 2 instead of 4 instructions per tap and symbol.  One statement per tap position (four
    // phases, both symbols: 16 instructions; hipcc pads register overlaps BETWEEN asm statements with s_nop).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 accA = { 0.f, 0.f }, accB = { 0.f, 0.f };
    // a3 .. a0 / b3 .. b0: the entries of phases 3 .. 0 for symbol A / B at this tap position; t01, t23: the four taps
    // (ascending index) that go with phases 3, 2, 1, 0
    auto mac4 = [&](f2 a3, f2 b3, f2 a2, f2 b2, f2 a1, f2 b1, f2 a0, f2 b0, f2 t01, f2 t23) {
        f2 pa, pb;
        asm("v_pk_mul_f32 %2, %4, %12 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %5, %12 op_sel_hi:[1,0]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "v_pk_mul_f32 %2, %6, %12 op_sel:[0,1]\n\t"
            "v_pk_mul_f32 %3, %7, %12 op_sel:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "v_pk_mul_f32 %2, %8, %13 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %9, %13 op_sel_hi:[1,0]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "v_pk_mul_f32 %2, %10, %13 op_sel:[0,1]\n\t"
            "v_pk_mul_f32 %3, %11, %13 op_sel:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3"
            : "+v"(accA), "+v"(accB), "=&v"(pa), "=&v"(pb)
            : "v"(a3), "v"(b3), "v"(a2), "v"(b2), "v"(a1), "v"(b1), "v"(a0), "v"(b0), "s"(t01), "s"(t23));
    };
    auto lo2 = [](const float4& v) { return f2{ v.x, v.y }; };
    auto hi2 = [](const float4& v) { return f2{ v.z, v.w }; };
    float4 hi[kFastSps], lo[kFastSps];
#pragma unroll
    for (unsigned ph = 0; ph < kFastSps; ++ph) hi[ph] = rows[ph * kRow4 + 5];
#pragma unroll
    for (int k = 5; k >= 0; --k) {
        if (k > 0) {
#pragma unroll
            for (unsigned ph = 0; ph < kFastSps; ++ph) lo[ph] = rows[ph * kRow4 + (k - 1)];
        }
        asm volatile("" ::: "memory"); // the reads of the next pair stay here, ahead of the MACs that hide them
        {   // q = 2k: A <- pair.lo, B <- pair.hi; taps 40 - 8k .. 43 - 8k
            const int m0 = static_cast<int>(kFastArm) - 4 - 4 * (2 * k);
            mac4(lo2(hi[3]), hi2(hi[3]), lo2(hi[2]), hi2(hi[2]), lo2(hi[1]), hi2(hi[1]), lo2(hi[0]), hi2(hi[0]),
                 f2{ tap[m0], tap[m0 + 1] }, f2{ tap[m0 + 2], tap[m0 + 3] });
        }
        if (k > 0) { // q = 2k - 1: A <- pair(k - 1).hi, B <- pair(k).lo
            const int m0 = static_cast<int>(kFastArm) - 4 - 4 * (2 * k - 1);
            mac4(hi2(lo[3]), lo2(hi[3]), hi2(lo[2]), lo2(hi[2]), hi2(lo[1]), lo2(hi[1]), hi2(lo[0]), lo2(hi[0]),
                 f2{ tap[m0], tap[m0 + 1] }, f2{ tap[m0 + 2], tap[m0 + 3] });
#pragma unroll
            for (unsigned ph = 0; ph < kFastSps; ++ph) hi[ph] = lo[ph];
        }
    }
    out[p.o0 + 2 * l] = scale_item(p.scale, cf{ accA.x, accA.y });
    if (2 * l + 1 < p.count) out[p.o0 + 2 * l + 1] = scale_item(p.scale, cf{ accB.x, accB.y });
    }
    }
    if (w + 1 < w_end) {
        p = plan[w + 1]; // (requested here, behind the multiply-adds: in front of them its 16 SGPRs do not fit beside the taps)
        __syncthreads(); // the tile is free for the next fill
    }
    }
}

// Long arms (BASELINE configs[4]: 32 arms x 1025 taps, 4 samples per symbol), plain SymbolFilter on complex items: the
// generic kernel above reads one 8-byte item and one tap from LDS per multiply-add (12 bytes per tap and symbol: 15 Gsps
// in, bound by the LDS pipe).  Here, as in k_symbol_filter_fast: a lane computes TWO neighbouring symbols, one
// ds_read_b128 delivers the two tile entries the pair needs at one tap position of one phase (4 bytes per tap and
// symbol), the arm's taps are uniform and stream through SGPRs (scalar loads, eight a time, the next eight requested
// before the current ones are used), and the multiply-adds are v_pk_mul_f32 / v_pk_add_f32 with the tap broadcast from
// an SGPR pair -- every product and every sum rounded once, tap index ascending (symbol_filter.hpp:208-214): bit-exact.
// Tap m of symbol s sits on tile item 4 s + j, j = arm_size - 1 - m: phase j % 4, entry s + j / 4 of the phase row.
// The arm starts with (arm_size - 1) % 4 + 1 "head" taps on the top tap position, then whole tap positions of four.
__global__ __launch_bounds__(kFastThreads, 8) void k_symbol_filter_long(const cf* __restrict__ in, const cf* __restrict__ carry,
                                                                      unsigned cap, const float* __restrict__ taps,
                                                                      unsigned arm_size, unsigned pitch,
                                                                      const SymWg* __restrict__ plan, cf* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_long_raw[];
    cf* tile = reinterpret_cast<cf*>(s_long_raw);
    const SymWg p = plan[blockIdx.x];
    const unsigned span = (p.count - 1) * kFastSps + arm_size;
    for (unsigned i = threadIdx.x; i < span; i += kFastThreads)
        tile[(i % kFastSps) * pitch + i / kFastSps] = item_at(in, carry, cap, p.lo_item + i);
    __syncthreads();
    const unsigned l = threadIdx.x;
    if (2 * l >= p.count) return;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float* __restrict__ tp = taps + static_cast<size_t>(p.arm) * arm_size; // uniform: scalar loads
    f2 accA = { 0.f, 0.f }, accB = { 0.f, 0.f };
    const unsigned J = arm_size - 1, q_top = J / kFastSps, r = J % kFastSps;
    // head taps m = 0 .. r: phases r .. 0 of tap position q_top
    for (unsigned m = 0; m <= r; ++m) {
        const unsigned ph = r - m;
        const cf a = tile[ph * pitch + 2 * l + q_top], b = tile[ph * pitch + 2 * l + 1 + q_top];
        const float t = tp[m];
        accA = f2{ accA.x + t * a.x, accA.y + t * a.y };
        accB = f2{ accB.x + t * b.x, accB.y + t * b.y };
    }
    // a3 .. a0 / b3 .. b0: the entries of phases 3 .. 0 for symbol A / B at one tap position; t01, t23: its four taps
    auto mac4 = [&](f2 a3, f2 b3, f2 a2, f2 b2, f2 a1, f2 b1, f2 a0, f2 b0, f2 t01, f2 t23) {
        f2 pa, pb;
        asm("v_pk_mul_f32 %2, %4, %12 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %5, %12 op_sel_hi:[1,0]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "v_pk_mul_f32 %2, %6, %12 op_sel:[0,1]\n\t"
            "v_pk_mul_f32 %3, %7, %12 op_sel:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "v_pk_mul_f32 %2, %8, %13 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %9, %13 op_sel_hi:[1,0]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "v_pk_mul_f32 %2, %10, %13 op_sel:[0,1]\n\t"
            "v_pk_mul_f32 %3, %11, %13 op_sel:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3"
            : "+v"(accA), "+v"(accB), "=&v"(pa), "=&v"(pb)
            : "v"(a3), "v"(b3), "v"(a2), "v"(b2), "v"(a1), "v"(b1), "v"(a0), "v"(b0), "s"(t01), "s"(t23));
    };
    auto lo2 = [](const float4& v) { return f2{ v.x, v.y }; };
    auto hi2 = [](const float4& v) { return f2{ v.z, v.w }; };
    // pair(ph, k) = tile entries (ph, 2l + 2k) and (ph, 2l + 2k + 1): symbol A uses entry 2l + q at tap position q,
    // symbol B entry 2l + 1 + q, so q = 2k takes (pair k .lo, pair k .hi) and q = 2k + 1 takes (pair k .hi, pair k+1 .lo)
    const float4* rows = reinterpret_cast<const float4*>(tile) + l;
    const unsigned row4 = pitch / 2; // float4 per phase row
    auto load_pair = [&](float4(&v)[kFastSps], unsigned k) {
#pragma unroll
        for (unsigned ph = 0; ph < kFastSps; ++ph) v[ph] = rows[ph * row4 + k];
    };
    auto even = [&](const float4(&P)[kFastSps], const float* t) { // q = 2k
        mac4(lo2(P[3]), hi2(P[3]), lo2(P[2]), hi2(P[2]), lo2(P[1]), hi2(P[1]), lo2(P[0]), hi2(P[0]), f2{ t[0], t[1] }, f2{ t[2], t[3] });
    };
    auto odd = [&](const float4(&P)[kFastSps], const float4(&Pn)[kFastSps], const float* t) { // q = 2k + 1, Pn = pair k + 1
        mac4(hi2(P[3]), lo2(Pn[3]), hi2(P[2]), lo2(Pn[2]), hi2(P[1]), lo2(Pn[1]), hi2(P[0]), lo2(Pn[0]), f2{ t[0], t[1] }, f2{ t[2], t[3] });
    };
    if (q_top > 0) {
        // eight taps with ONE scalar load (s_load_dwordx8; the address only needs dword alignment)
        typedef float f8 __attribute__((ext_vector_type(8), aligned(4)));
        unsigned m = r + 1;     // next tap
        unsigned Q = q_top - 1; // next tap position
        float4 P0[kFastSps], P1[kFastSps];
        unsigned k = Q / 2;
        load_pair(P0, k);
        if (Q & 1u) {
            load_pair(P1, k + 1);
            float t[4] = { tp[m], tp[m + 1], tp[m + 2], tp[m + 3] };
            odd(P0, P1, t);
            m += 4;
        }
        {
            float t[4] = { tp[m], tp[m + 1], tp[m + 2], tp[m + 3] };
            even(P0, t);
            m += 4;
        }
        // From here on two tap positions (eight taps) per pass, two passes per iteration with the pair registers P0 / P1
        // taking turns (no moves).  The pair of the NEXT pass is requested between the two halves of a pass -- its
        // registers are free once odd() has used their .lo halves -- so sixteen packed instructions cover the LDS
        // latency; the sixteen taps of the next iteration are requested before this iteration's are used.
        if (k > 0) load_pair(P1, k - 1);
        f8 ta = *reinterpret_cast<const f8*>(tp + m), tb = *reinterpret_cast<const f8*>(tp + min(m + 8, arm_size - 8));
        while (k >= 2) {
            const f8 ua = ta, ub = tb;
            m += 16;
            // (clamped: the last iterations request taps they do not use instead of branching)
            ta = *reinterpret_cast<const f8*>(tp + min(m, arm_size - 8));
            tb = *reinterpret_cast<const f8*>(tp + min(m + 8, arm_size - 8));
            {
                const float t[8] = { ua[0], ua[1], ua[2], ua[3], ua[4], ua[5], ua[6], ua[7] };
                odd(P1, P0, t); // pass a: pair k - 1, tap positions 2k - 1 and 2k - 2
                load_pair(P0, k - 2);
                asm volatile("" ::: "memory"); // the request stays here, in front of the sixteen instructions that hide it
                even(P1, t + 4);
            }
            {
                const float t[8] = { ub[0], ub[1], ub[2], ub[3], ub[4], ub[5], ub[6], ub[7] };
                odd(P0, P1, t); // pass b: pair k - 2
                if (k >= 3) load_pair(P1, k - 3);
                asm volatile("" ::: "memory");
                even(P0, t + 4);
            }
            k -= 2;
        }
        if (k == 1) {
            const float t[8] = { ta[0], ta[1], ta[2], ta[3], ta[4], ta[5], ta[6], ta[7] };
            odd(P1, P0, t);
            even(P1, t + 4);
        }
    }
    out[p.o0 + 2 * l] = scale_item(p.scale, cf{ accA.x, accA.y });
    if (2 * l + 1 < p.count) out[p.o0 + 2 * l + 1] = scale_item(p.scale, cf{ accB.x, accB.y });
}
// which (fused, item type, sps, arm size) combinations run it
// (its tile: entries per phase row = the last symbol's entry at the top tap position, + 1 for symbol B; even)
static unsigned symf_long_pitch(size_t arm_size) { return ((kFastSym + (static_cast<unsigned>(arm_size) - 1) / kFastSps + 2) + 1u) & ~1u; }
static size_t symf_long_lds(size_t arm_size) { return static_cast<size_t>(kFastSps) * symf_long_pitch(arm_size) * sizeof(cf); }
static bool symf_long(bool fused, bool cf_items, size_t sps, size_t arm_size)
{
    static const bool off = getenv("GR4PM_SYMF_GENERIC") != nullptr;
    // arms whose tile does not fit the LDS (arm sizes beyond ~20 000) stay with the generic kernel
    return !fused && cf_items && sps == kFastSps && arm_size >= 64 && symf_long_lds(arm_size) <= kFirMaxSmem && !off;
}

#ifdef GR4PM_EXPERIMENTS
// GR4PM_TIMING_SKIP=symf_fake / costas_fake: timing experiments only.  Stand-ins with the memory traffic (symbol
// filter) or the life time (Costas) of the kernel they replace and at most 32 VGPRs, no LDS: what would the chain
// gain if the real kernel fitted beside two 240-VGPR correlator waves of every SIMD?
__global__ __launch_bounds__(kFastThreads) void k_symf_fake(const cf* __restrict__ in, const SymWg* __restrict__ plan,
                                                           cf* __restrict__ out)
{
    const SymWg p = plan[blockIdx.x];
    const float4* ip = reinterpret_cast<const float4*>(in + (p.lo_item > 0 ? (p.lo_item & ~1ll) : 0));
    float4 a = ip[threadIdx.x], b = ip[threadIdx.x + 128], c = ip[threadIdx.x + 256];
    float4 d = threadIdx.x < 96 ? ip[threadIdx.x + 384] : a;
    a.x += b.x + c.x + d.x, a.y += b.y + c.y + d.y, a.z += b.z + c.z + d.z, a.w += b.w + c.w + d.w;
    if (threadIdx.x < 120) reinterpret_cast<float4*>(out + (p.o0 & ~1u))[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void k_serial_fake(unsigned ticks, float* sink)
{
    const unsigned long long t0 = wall_clock64();
    float x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 32; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    }
    if (x == 12345.0f) *sink = x;
}
#endif

// symbols per workgroup of the kernel that a (fused, sps, arm size) combination runs
static bool symf_fast(bool fused, size_t sps, size_t arm_size)
{
    static const bool off = getenv("GR4PM_SYMF_GENERIC") != nullptr; // A/B switch: the generic kernel for every design
    return fused && sps == kFastSps && arm_size == kFastArm && !off;
}
static unsigned symf_per_wg(bool fused, size_t sps, size_t arm_size, bool cf_items = true)
{
    return (symf_fast(fused, sps, arm_size) || symf_long(fused, cf_items, sps, arm_size)) ? kFastSym : kSymPerWg;
}

template <typename T, bool CFC>
static void launch_symbol_filter(hipStream_t s, unsigned n_wg, size_t smem, unsigned sps, const T* in,
                                 const T* carry, unsigned cap, const float* taps, unsigned arm_size,
                                 const SymRun* runs, unsigned n_runs, SymWg* plan, T* out, CfcDev cfc,
                                 const SymChan* chans = nullptr)
{
    const dim3 grid(n_wg), block(kSymPerWg);
    hipLaunchKernelGGL(k_symf_wg_plan, dim3((n_wg + 255) / 256), dim3(256), 0, s, runs, n_runs, n_wg, sps, arm_size,
                       cfc, plan, chans, symf_per_wg(CFC, sps, arm_size, std::is_same<T, cf>::value));
    if constexpr (!CFC && std::is_same<T, cf>::value) {
        if (symf_long(false, true, sps, arm_size)) {
            const unsigned pitch = symf_long_pitch(arm_size);
            const size_t lds = symf_long_lds(arm_size);
            // (a refusal stays in hipGetLastError(), which the caller reads behind its launches; the launch below is
            // then refused too)
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_symbol_filter_long),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            hipLaunchKernelGGL(k_symbol_filter_long, grid, dim3(kFastThreads), lds, s, in, carry, cap, taps, arm_size, pitch,
                               plan, out);
            return;
        }
    }
    if constexpr (CFC) {
        if (symf_fast(true, sps, arm_size)) {
#ifdef GR4PM_EXPERIMENTS
            if (timing_skip("symf_fake"))
                hipLaunchKernelGGL(k_symf_fake, grid, dim3(kFastThreads), 0, s, in, plan, out);
            else
#endif
            if (!timing_skip("symf")) {
                static const char* abl_e = gr4pm::experiment_env("GR4PM_SYMF_ABL", true);
                static const int abl = abl_e ? atoi(abl_e) : 0;
                static const unsigned pad = gr4pm::experiment_env_wg("GR4PM_SYMF_PAD", 0u, 0u, 64u * 1024u);
                static const unsigned tiles = gr4pm::experiment_env_wg("GR4PM_SYMF_TILES", 1u, 1u, 64u);
                const dim3 gridf((n_wg + tiles - 1) / tiles);
                SymChan one{}; // the single-channel launch's table entry (unused when `chans` is given)
                one.in = in, one.carry = carry, one.out = out;
                one.segs = cfc.segs, one.seg_incr = cfc.seg_incr, one.seg_counter0 = cfc.seg_counter0;
                one.n_segs = cfc.n_segs;
#define GR4PM_SYMF_LAUNCH(A)                                                                                         \
    do {                                                                                                             \
        if (tiles > 1)                                                                                               \
            hipLaunchKernelGGL((k_symbol_filter_fast<A, true>), gridf, dim3(kFastThreads), pad, s, one, chans, plan, \
                               taps, cfc.ck, cap, n_wg, tiles);                                                      \
        else                                                                                                         \
            hipLaunchKernelGGL((k_symbol_filter_fast<A, false>), gridf, dim3(kFastThreads), pad, s, one, chans, plan, \
                               taps, cfc.ck, cap, n_wg, tiles);                                                      \
    } while (0)
#ifdef GR4PM_EXPERIMENTS
                if (abl == 1) GR4PM_SYMF_LAUNCH(1);
                else if (abl == 2) GR4PM_SYMF_LAUNCH(2);
                else if (abl == 3) GR4PM_SYMF_LAUNCH(3);
                else
#endif
                    GR4PM_SYMF_LAUNCH(0);
                (void)abl;
#undef GR4PM_SYMF_LAUNCH
            }
            return;
        }
    }
    if (sps == 4)
        hipLaunchKernelGGL((k_symbol_filter<T, 4, CFC>), grid, block, smem, s, in, carry, cap, taps, arm_size, sps,
                           plan, out, cfc, chans);
    else if (sps == 2)
        hipLaunchKernelGGL((k_symbol_filter<T, 2, CFC>), grid, block, smem, s, in, carry, cap, taps, arm_size, sps,
                           plan, out, cfc, chans);
    else if (sps == 8)
        hipLaunchKernelGGL((k_symbol_filter<T, 8, CFC>), grid, block, smem, s, in, carry, cap, taps, arm_size, sps,
                           plan, out, cfc, chans);
    else
        hipLaunchKernelGGL((k_symbol_filter<T, 0, CFC>), grid, block, smem, s, in, carry, cap, taps, arm_size, sps,
                           plan, out, cfc, chans);
}

// PfbArbResampler (pfb_arb_resampler.hpp:134-167).  The accumulator recurrence decides which
// input and which arm every output uses; it is float/double rounding dependent, so one lane
// replays it serially and writes a plan; the two inner products per output run in parallel.
struct ArbState {
    unsigned long long last_filter;
    double phase_acc_d;
    float phase_acc_f;
    unsigned produced;
    unsigned long long consumed;
};
// The serial lane keeps to the recurrence itself and leaves a checkpoint of its state every kArbChunk outputs (round 1
// wrote three plan arrays entry by entry from that one lane: 147 ns per output, 6.8 Msamples/s); the filter kernel's
// lanes replay at most kArbChunk - 1 steps from their chunk's checkpoint -- the same operations in the same order, so
// the same (input index, arm, phase) as the serial walk -- and go on to their two inner products.
constexpr unsigned kArbChunk = 64;
template <typename TRate>
struct ArbCk {
    unsigned ii;          // inputs consumed when output k * kArbChunk is formed
    unsigned last_filter; // < filter_size there
    TRate phase_acc;
    unsigned pad[sizeof(TRate) == 8 ? 2 : 3];
};
// One pass of the reference loop between two outputs that both exist: the update of pfb_arb_resampler.hpp:161-166, then
// the input items of :135-138.  With last_filter < filter_size before, decim_rate = q0 filter_size + r0 and wrap <= 1
// the walk "while (last_filter >= filter_size) { ++ii; last_filter -= filter_size; }" takes q0 or q0 + 1 items:
// no loop, no division, 32-bit integers -- the same values as the walk.
template <typename TRate>
__device__ __forceinline__ void arb_step(unsigned& ii, unsigned& last_filter, TRate& phase_acc, unsigned filter_size,
                                         unsigned q0, unsigned r0, TRate filt_rate)
{
    phase_acc += filt_rate;
    const bool wrap = phase_acc > TRate{ 1 };
    phase_acc = wrap ? phase_acc - TRate{ 1 } : phase_acc;
    const unsigned t = last_filter + r0 + (wrap ? 1u : 0u);
    const bool c = t >= filter_size;
    ii += q0 + (c ? 1u : 0u);
    last_filter = c ? t - filter_size : t;
}
// kArbChunk steps of the phase accumulator (see k_arb_plan); eight steps per asm statement (hipcc pads register
// overlaps between statements)
__device__ __forceinline__ void arb_phase_chunk(double& acc, double rate)
{
    const double K = 0x1p1000;
    double t, m;
#pragma unroll
    for (unsigned k = 0; k < kArbChunk; k += 8)
        asm volatile("v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2\n\t"
                     "v_add_f64 %1, %0, %3\n\tv_fma_f64 %2, %1, %4, -%4 clamp\n\tv_add_f64 %0, %1, -%2"
                     : "+v"(acc), "=&v"(t), "=&v"(m)
                     : "v"(rate), "v"(K));
}
__device__ __forceinline__ void arb_phase_chunk(float& acc, float rate)
{
    const float K = 0x1p100f;
    float t, m;
#pragma unroll
    for (unsigned k = 0; k < kArbChunk; k += 8)
        asm volatile("v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2\n\t"
                     "v_add_f32 %1, %0, %3\n\tv_fma_f32 %2, %1, %4, -%4 clamp\n\tv_sub_f32 %0, %1, %2"
                     : "+v"(acc), "=&v"(t), "=&v"(m)
                     : "v"(rate), "v"(K));
}

template <typename TRate>
__global__ void k_arb_plan(ArbState* __restrict__ st, unsigned n_in, unsigned out_cap, unsigned filter_size,
                           unsigned long long decim_rate, unsigned q0, unsigned r0, TRate filt_rate,
                           ArbCk<TRate>* __restrict__ ck)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long last_filter = st->last_filter;
    TRate phase_acc = sizeof(TRate) == 8 ? static_cast<TRate>(st->phase_acc_d)
                                         : static_cast<TRate>(st->phase_acc_f);
    unsigned ii = 0, oi = 0;
    // first pass of the reference loop (n_in > 0 and out_cap > 0: checked by the caller): the carried last_filter may
    // ask for any number of items
    while (last_filter >= filter_size && ii < n_in) {
        ++ii;
        last_filter -= filter_size;
    }
    if (last_filter < filter_size) {
        unsigned lf = static_cast<unsigned>(last_filter);
        for (;;) { // state: output oi is about to be formed from (ii, lf, phase_acc)
            // whole chunks while neither the input nor the output can end inside one: a checkpoint, then kArbChunk
            // steps of straight-line code (every pass of the reference loop in between finds its loop condition true
            // and its items there); the chain of phase_acc is then all that a step costs
            while ((oi & (kArbChunk - 1)) == 0 && oi + kArbChunk < out_cap &&
                   static_cast<unsigned long long>(ii) + static_cast<unsigned long long>(kArbChunk) * (q0 + 1u) < n_in) {
                ArbCk<TRate> c{};
                c.ii = ii, c.last_filter = lf, c.phase_acc = phase_acc;
                ck[oi / kArbChunk] = c;
                // inside a chunk only phase_acc is a chain: the kArbChunk conditional subtractions of filter_size add up
                // to a division of lf + kArbChunk r0 + (number of wraps) by filter_size (every partial sum stays below
                // 2 filter_size, so the walk subtracts exactly when the running sum passes a multiple)
                // THREE dependent instructions per step, no compare, no select, no counter:
                //   t = phase_acc + filt_rate ; m = clamp(t * K - K) ; phase_acc = t - m
                // with K = 2^1000 (2^100 for float): the fused multiply-add is > 1 for every t > 1 (t - 1 >= 2^-52),
                // <= 0 for every t <= 1, so the [0, 1] clamp of the instruction's output modifier makes m exactly
                // 1.0 or 0.0 -- the reference's `if (phase_acc > 1) phase_acc -= 1` (t - 0.0 == t bit for bit).
                // The number of wraps falls out at the end: start + kArbChunk * filt_rate - end is that integer up
                // to rounding noise of 1e-5 at most.  (Round 2: add, add, compare, two selects + three instructions
                // of counting per step: 28 ns per output; now 3 per step.)
                const TRate start = phase_acc;
                arb_phase_chunk(phase_acc, filt_rate);
                const unsigned wraps = static_cast<unsigned>(
                    __double2ll_rn(static_cast<double>(start) + static_cast<double>(kArbChunk) * static_cast<double>(filt_rate) -
                                   static_cast<double>(phase_acc)));
                const unsigned long long sum = static_cast<unsigned long long>(lf) + static_cast<unsigned long long>(kArbChunk) * r0 + wraps;
                const unsigned long long sub = sum / filter_size;
                lf = static_cast<unsigned>(sum - sub * filter_size);
                ii += kArbChunk * q0 + static_cast<unsigned>(sub);
                oi += kArbChunk;
            }
            if ((oi & (kArbChunk - 1)) == 0) {
                ArbCk<TRate> c{};
                c.ii = ii, c.last_filter = lf, c.phase_acc = phase_acc;
                ck[oi / kArbChunk] = c;
            }
            ++oi;
            if (ii < n_in && oi < out_cap) { // the reference's loop condition for the next pass
                const unsigned need = q0 + ((lf + r0 + 1u >= filter_size) ? 1u : 0u); // at most this many items
                if (ii + need <= n_in) {
                    arb_step(ii, lf, phase_acc, filter_size, q0, r0, filt_rate);
                    continue;
                }
            }
            // last pass: the update, then -- if the loop goes on at all -- the walk over what is left of the input
            phase_acc += filt_rate;
            last_filter = static_cast<unsigned long long>(lf) + decim_rate;
            if (phase_acc > TRate{ 1 }) {
                phase_acc -= TRate{ 1 };
                ++last_filter;
            }
            if (!(ii < n_in && oi < out_cap)) break;
            while (last_filter >= filter_size && ii < n_in) {
                ++ii;
                last_filter -= filter_size;
            }
            if (last_filter >= filter_size) break;
            lf = static_cast<unsigned>(last_filter); // the items sufficed after all (need was the upper bound)
        }
    }
    st->last_filter = last_filter;
    st->phase_acc_d = static_cast<double>(phase_acc);
    st->phase_acc_f = static_cast<float>(phase_acc);
    st->produced = oi;
    st->consumed = ii;
}
template <typename TRate>
__global__ void k_arb_filter(const cf* __restrict__ in, const cf* __restrict__ carry, unsigned cap,
                             const float* __restrict__ taps, const float* __restrict__ diff_taps,
                             unsigned arm_size, const ArbState* __restrict__ st, const ArbCk<TRate>* __restrict__ ck,
                             unsigned filter_size, unsigned q0, unsigned r0, TRate filt_rate, cf* __restrict__ out)
{
    const unsigned n_out = st->produced;
    for (unsigned o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x) {
        const ArbCk<TRate> c = ck[o / kArbChunk];
        unsigned ii = c.ii, last_filter = c.last_filter;
        TRate phase_acc = c.phase_acc;
        for (unsigned r = o % kArbChunk; r > 0; --r) arb_step(ii, last_filter, phase_acc, filter_size, q0, r0, filt_rate);
        const long long idx = static_cast<long long>(ii) - 1;
        const float* arm = taps + static_cast<size_t>(last_filter) * arm_size;
        const float* darm = diff_taps + static_cast<size_t>(last_filter) * arm_size;
        cf filt = { 0.f, 0.f }, diff = { 0.f, 0.f };
        for (unsigned m = 0; m < arm_size; ++m) filt = mac(filt, arm[m], item_at(in, carry, cap, idx - m));
        for (unsigned m = 0; m < arm_size; ++m) diff = mac(diff, darm[m], item_at(in, carry, cap, idx - m));
        out[o] = cadd(filt, fmulc(static_cast<float>(phase_acc), diff)); // :153-160
    }
}
// history after the call: last cap items of (carry ++ in[0..consumed))
__global__ void k_arb_update_hist(const cf* __restrict__ in, const cf* __restrict__ carry,
                                  cf* __restrict__ carry_next, unsigned cap,
                                  const ArbState* __restrict__ st)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    carry_next[i] = item_at(in, carry, cap, static_cast<long long>(st->consumed) - cap + i);
}

template <typename T>
gr4pm_status upload_vec(DevBuf<T>& buf, const std::vector<T>& v, hipStream_t s)
{
    if (buf.n < v.size()) GR4PM_TRY(buf.alloc(std::max<size_t>(v.size() * 2, 64)));
    return buf.upload_staged(v.data(), v.size(), s);
}

unsigned grid_for(size_t n, unsigned block, unsigned cap = 65535u * 16u)
{
    const size_t g = (n + block - 1) / block;
    return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(g, cap)));
}

} // namespace
} // namespace gr4pm

using namespace gr4pm;

// ------------------------------------------------------------------------ Rotator / CFC
struct gr4pm_rotator {
    int mode;
    float phase_incr;
    size_t delay, n_channels;
    hipStream_t stream;
    // [kStates][n_channels], st_cur selects the row a call reads; it writes the next one (round 6: a ring instead of two
    // halves -- the chain kernels of several plans are in flight at once, see PlanSync)
    static constexpr int kStates = GR4PM_CFC_PLANS + 2;
    DevBuf<RotState> state;
    int st_cur = 0;
    // Round 6: the chains of CONSECUTIVE ring plans run side by side.  A plan's segments that start at a set_freq event
    // (or are fixed points) depend on nothing before them: their kernel goes to one of kAux streams of the handle's own.
    // The segments that continue the carried phasor -- at most one per channel -- need the state the plan before wrote:
    // their kernel goes to `dep`, one stream for all plans, behind the other kernel of the plan before.  The consumers
    // (the fused symbol filter) wait for the plan's two events on THEIR stream; the stream the plan was made on carries the
    // uploads only, so the pipeline stage that makes the plans no longer waits for a chain.  With one packet per 2^20
    // samples a chain is 2^20 dependent steps (16.7 ms) per 2^28-sample batch: one behind the other they were the
    // receiver's period (15 Gsps); side by side they are its latency.  GR4PM_ROT_SERIAL=1: one kernel on the handle's stream.
    // Three kernels a plan, each on a stream of its own (kAux of each kind, taken in turn): `writer` = the channels' LAST
    // segments where they start at an event (they write the carried state: the plan behind waits for this kernel alone, not
    // for the other 2^20-step chains of the plan), `indep` = the other event-started segments, `dep` = the continuations.
    static constexpr int kAux = 4;
    hipStream_t aux[3 * kAux] = {};
    bool async_ready = false;
    struct PlanSync {
        hipEvent_t up = nullptr, indep = nullptr, writer = nullptr, dep = nullptr;
        bool async = false;
        bool dep_writes_state = false; // a continuation is its channel's last segment (no event in the call)
    } sync[GR4PM_CFC_PLANS];
    int last_async_plan = -1; // the plan whose kernels wrote the state row st_cur (or -1: written on `stream`)
    // When: a ring plan whose longest chain is at least kAsyncMinItems long (2 ms of dependent steps; packets back to back
    // make chains of 26 000 items and gain nothing: 56.6 -> 55 Gsps with three kernels and their events per plan), in a
    // process whose HIP runtime has at least eight hardware queues (GPU_MAX_HW_QUEUES; its default is four, and a chain
    // kernel that shares a queue with another stream's work holds that work back for as long as it lives: with four
    // queues the side-by-side form is SLOWER than one kernel, 14.8 against 18.3 Gsps at one packet per 2^20 samples,
    // with sixteen it is 41.8).  Read at creation: GR4PM_ROT_SERIAL=1 never, GR4PM_ROT_ASYNC=1 always (tests).
    static constexpr size_t kAsyncMinItems = size_t{ 1 } << 17;
    int async_policy = 0; // 0: by chain length and queue count, 1: always, -1: never
    unsigned test_delay_us = 0; // GR4PM_TEST_ROT_DELAY_US: every chain kernel of an asynchronous plan starts that much later
    // the plan of a call (segment table, phasor checkpoints, increments, counters): a ring of
    // GR4PM_CFC_PLANS sets, so that the next calls can be planned while the consumers of earlier
    // plans are still running (gr4pm_cfc_symbol_filter_plan / _run; buffers are allocated on first use)
    struct Plan {
        DevBuf<RotSeg> segs;
        DevBuf<cf> ck, seg_incr;
        DevBuf<unsigned> seg_counter0, order;
        DevBuf<unsigned> const_list; // the mode-2 segments of THIS plan (its own staging buffer: plans are issued ahead)
        unsigned n_segs = 0;
        size_t n_in = 0;
        std::vector<unsigned> seg_first; // [n_channels + 1]: the segments of channel c are [seg_first[c], seg_first[c + 1])
    } plans[GR4PM_CFC_PLANS];
    int plan_cur = 0;
    bool ring_sized = false;
    // gr4pm_cfc_symbol_filter_run_channels: channel table + runs of all channels (one upload), workgroup plan
    DevBuf<unsigned long long> mc_tab;
    DevBuf<SymWg> mc_wg;
    std::vector<unsigned long long> mc_host;
    std::vector<float> next_freq;     // per channel, coarse_frequency_correction.hpp:44
    std::vector<long> next_freq_delay; // :45
    // Per channel: the carried phasor is a fixed point of the recurrence -- exp = (1, -+0) with incr = (1, -+0): every
    // product e * incr gives e again and the renormalisation divides by hypot(1, 0) = 1.  That is the state of a
    // CoarseFrequencyCorrection from start() to its first syncword_freq tag (set_freq(0) on the first item,
    // coarse_frequency_correction.hpp:44-45,84-86), after every tag whose frequency is exactly 0, and of a Rotator with
    // phase_incr 0: the stream the reference publishes its receiver benchmark on (zeros: no tag, ever) and every
    // stream until its first detection.  Such a segment needs no serial chain: its checkpoints are the constant
    // (k_rot_const_fill, parallel), the consumers multiply by it as before -- the same bits (x * (1, -0) is not a
    // copy: it turns -0 into +0 in places, as the reference's multiplication does).
    std::vector<uint8_t> fixed;
    std::vector<cf> fixed_exp, fixed_incr;
};

// the handle's own streams idle (the chains of ring plans run there)
static void rotator_sync_own_streams(gr4pm_rotator* h)
{
    for (hipStream_t a : h->aux)
        if (a) (void)hipStreamSynchronize(a);
}

static gr4pm_status rotator_reset_impl(gr4pm_rotator* h)
{
    rotator_sync_own_streams(h);
    h->last_async_plan = -1;
    for (auto& y : h->sync) y.async = false;
    std::vector<RotState> st(h->n_channels);
    for (auto& s : st) {
        s.exp = { 1.0f, 0.0f };
        s.counter = 0;
        s.pad = 0;
        if (h->mode == 0) // rotator.hpp:44-48 settingsChanged + :50-54 start
            s.incr = { std::cos(h->phase_incr), std::sin(h->phase_incr) };
        else
            s.incr = { 1.0f, 0.0f };
    }
    GR4PM_HIP_TRY(hipMemcpyAsync(h->state.p, st.data(), st.size() * sizeof(RotState), hipMemcpyHostToDevice,
                                 h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    h->st_cur = 0;
    h->next_freq.assign(h->n_channels, 0.0f);
    h->next_freq_delay.assign(h->n_channels, 0); // :45 -> set_freq(0) on the first item
    // (a Rotator whose increment is (1, +-0) never leaves exp = (1, +0); a CFC starts with set_freq(0) on item 0)
    h->fixed.assign(h->n_channels, h->mode == 0 && st[0].incr.x == 1.0f && st[0].incr.y == 0.0f ? 1 : 0);
    h->fixed_exp.assign(h->n_channels, st[0].exp);
    h->fixed_incr.assign(h->n_channels, st[0].incr);
    return GR4PM_OK;
}

extern "C" {

gr4pm_status gr4pm_rotator_create(const gr4pm_rotator_params* p, gr4pm_rotator** out)
try {
    if (!p || !out || p->n_channels == 0 || (p->mode != 0 && p->mode != 1)) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_rotator;
    if (!h) return GR4PM_ERR_NOMEM;
    h->mode = p->mode;
    h->phase_incr = p->phase_incr;
    h->delay = p->delay;
    h->n_channels = p->n_channels;
    h->stream = static_cast<hipStream_t>(p->stream);
    {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int hw_queues = q ? atoi(q) : 4;
        h->async_policy = getenv("GR4PM_ROT_SERIAL") ? -1 : getenv("GR4PM_ROT_ASYNC") ? 1 : hw_queues >= 8 ? 0 : -1;
        if (const char* d = getenv("GR4PM_TEST_ROT_DELAY_US")) h->test_delay_us = static_cast<unsigned>(std::max(0, atoi(d)));
    }
    gr4pm_status s = h->state.alloc(static_cast<size_t>(gr4pm_rotator::kStates) * h->n_channels);
    if (s == GR4PM_OK) s = rotator_reset_impl(h);
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_rotator_destroy(gr4pm_rotator* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    rotator_sync_own_streams(h);
    for (auto& y : h->sync) {
        if (y.up) (void)hipEventDestroy(y.up);
        if (y.indep) (void)hipEventDestroy(y.indep);
        if (y.writer) (void)hipEventDestroy(y.writer);
        if (y.dep) (void)hipEventDestroy(y.dep);
    }
    for (hipStream_t a : h->aux)
        if (a) (void)hipStreamDestroy(a);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_rotator_reset(gr4pm_rotator* h)
try {
    return h ? rotator_reset_impl(h) : GR4PM_ERR_INVALID;
}
GR4PM_ABI_CATCH

} // extern "C"

// host replay of the tag-driven control flow + the serial phasor checkpoints; leaves the segment
// table, checkpoints, increments and counters of this call on the device (h->plans[h->plan_cur])
// ring: the plan goes to the next set of the ring (callers that keep several plans alive:
// gr4pm_cfc_symbol_filter_plan*); otherwise the current set is reused.  When the ring is used for
// the first time every set gets the capacity of the first plan, so that no later call of a
// steady stream has to allocate.
static gr4pm_status rotator_plan_impl(gr4pm_rotator* h, size_t n, const gr4pm_tag* tags, const uint32_t* tag_channel,
                                      size_t n_tags, std::vector<RotSeg>& segs, bool ring);
// The host half of the carried state (pending frequency, fixed-point flags) is advanced while the segments are formed,
// before the allocations, uploads and launches that can still fail: a call that fails leaves it as it found it, in step
// with the device's RotState (st_cur flips only on success).
static gr4pm_status rotator_plan(gr4pm_rotator* h, size_t n, const gr4pm_tag* tags, const uint32_t* tag_channel,
                                 size_t n_tags, std::vector<RotSeg>& segs, bool ring = false)
{
    static thread_local std::vector<float> nf;
    static thread_local std::vector<long> nd;
    static thread_local std::vector<uint8_t> fx;
    static thread_local std::vector<cf> fe, fi;
    nf = h->next_freq, nd = h->next_freq_delay, fx = h->fixed, fe = h->fixed_exp, fi = h->fixed_incr;
    const int plan_was = h->plan_cur;
    gr4pm_status st;
    try {
        st = rotator_plan_impl(h, n, tags, tag_channel, n_tags, segs, ring);
    } catch (...) {
        h->next_freq.swap(nf), h->next_freq_delay.swap(nd), h->fixed.swap(fx), h->fixed_exp.swap(fe), h->fixed_incr.swap(fi);
        h->plan_cur = plan_was;
        throw;
    }
    if (st != GR4PM_OK) {
        h->next_freq.swap(nf), h->next_freq_delay.swap(nd), h->fixed.swap(fx), h->fixed_exp.swap(fe), h->fixed_incr.swap(fi);
        h->plan_cur = plan_was;
    }
    return st;
}
static gr4pm_status rotator_plan_impl(gr4pm_rotator* h, size_t n, const gr4pm_tag* tags, const uint32_t* tag_channel,
                                      size_t n_tags, std::vector<RotSeg>& segs, bool ring)
{
    unsigned ck = 0;
    size_t n_const = 0;
    for (size_t c = 0; c < h->n_channels; ++c) {
        // set_freq events (item, freq) of this channel: coarse_frequency_correction.hpp:76-96
        struct Ev {
            size_t at;
            float freq;
        };
        std::vector<Ev> evs;
        bool pending = false;
        size_t pending_at = 0;
        float pending_freq = 0.0f;
        if (h->mode == 1) {
            if (h->next_freq_delay[c] >= 0) {
                pending = true;
                pending_at = static_cast<size_t>(h->next_freq_delay[c]);
                pending_freq = h->next_freq[c];
            }
            for (size_t t = 0; t < n_tags; ++t) {
                const size_t tc = tag_channel ? tag_channel[t] : 0;
                if (tc != c || !(tags[t].flags & GR4PM_TAG_SYNCWORD) || tags[t].index >= n) continue;
                const size_t i = static_cast<size_t>(tags[t].index);
                // a countdown that has not reached zero when the next tag arrives is
                // overwritten (:79-80 runs before the item loop of that chunk)
                if (pending && pending_at < i) evs.push_back({ pending_at, pending_freq });
                pending = true;
                pending_at = i + h->delay;
                pending_freq = static_cast<float>(tags[t].freq); // :79 cast to float
            }
            if (pending && pending_at < n) {
                evs.push_back({ pending_at, pending_freq });
                pending = false;
            }
            h->next_freq[c] = pending_freq;
            h->next_freq_delay[c] = pending ? static_cast<long>(pending_at - n) : -1;
        }
        // pieces of this channel: [0, first event) continues the carried phasor; every event
        // starts a piece with a fresh phasor (set_freq resets _exp and _counter, :55-58)
        size_t pos = 0;
        static const bool no_fixed = getenv("GR4PM_ROT_NO_FIXED_POINT") != nullptr; // A/B: every segment as a chain
        auto push = [&](size_t start, size_t end, int mode, float freq) {
            if (end <= start) return;
            RotSeg g{};
            g.start = start;
            g.len = end - start;
            g.channel = static_cast<unsigned>(c);
            ck = (ck + 1u) & ~1u; // an even slot: 16-byte checkpoint stores without a test (k_rot_checkpoints_fresh)
            g.ck0 = ck;
            g.mode = mode;
            g.last = 0;
            if (mode == 1) { // set_freq(), :50-59 (float cos/sin of the host libm)
                const float d = static_cast<float>(h->delay);
                g.exp0 = { std::cos(freq * d), -std::sin(freq * d) };
                g.incr = { std::cos(freq), -std::sin(freq) };
                // exp0 = (1, -+0), incr = (1, -+0) (freq = +-0): e * incr == e for ever
                h->fixed[c] = g.exp0.x == 1.0f && g.exp0.y == 0.0f && g.incr.x == 1.0f && g.incr.y == 0.0f;
                h->fixed_exp[c] = g.exp0;
                h->fixed_incr[c] = g.incr;
            }
            if (h->fixed[c] && !no_fixed) { // (mode 0: the carried phasor is the fixed point)
                g.mode = 2;
                g.exp0 = h->fixed_exp[c];
                g.incr = h->fixed_incr[c];
                n_const += 1;
            }
            ck += static_cast<unsigned>((g.len + kRotChunk - 1) / kRotChunk);
            segs.push_back(g);
        };
        for (size_t k = 0; k < evs.size(); ++k) {
            if (evs[k].at > pos) push(pos, evs[k].at, 0, 0.0f); // only possible for k == 0
            const size_t end = k + 1 < evs.size() ? evs[k + 1].at : n;
            push(evs[k].at, end, 1, evs[k].freq);
            pos = end;
        }
        if (pos < n) push(pos, n, 0, 0.0f);
        segs.back().last = 1; // the channel's final piece writes the carried state
    }
    hipStream_t s = h->stream;
    const unsigned n_segs = static_cast<unsigned>(segs.size());
    unsigned ck_total = 0;
    for (const auto& g : segs) ck_total = std::max<unsigned>(ck_total, g.ck0 + static_cast<unsigned>((g.len + kRotChunk - 1) / kRotChunk));
    if (ring) {
        h->plan_cur = (h->plan_cur + 1) % GR4PM_CFC_PLANS;
        if (!h->ring_sized) {
            h->ring_sized = true;
            for (auto& q : h->plans) {
                if (q.ck.n < ck) GR4PM_TRY(q.ck.alloc(static_cast<size_t>(ck) * 2));
                if (q.seg_incr.n < n_segs) {
                    GR4PM_TRY(q.seg_incr.alloc(n_segs * 2));
                    GR4PM_TRY(q.seg_counter0.alloc(n_segs * 2));
                }
                if (q.segs.n < n_segs) GR4PM_TRY(q.segs.alloc(n_segs * 2));
                GR4PM_TRY(q.segs.reserve_stage(n_segs));
                if (q.order.n < n_segs) GR4PM_TRY(q.order.alloc(n_segs * 2));
                GR4PM_TRY(q.order.reserve_stage(n_segs));
                if (q.const_list.n < n_segs) GR4PM_TRY(q.const_list.alloc(n_segs * 2));
                GR4PM_TRY(q.const_list.reserve_stage(n_segs));
            }
        }
    }
    auto& pl = h->plans[h->plan_cur];
    pl.n_segs = n_segs;
    pl.n_in = n;
    pl.seg_first.assign(h->n_channels + 1, n_segs); // segments were generated channel by channel
    for (unsigned i = n_segs; i-- > 0;) pl.seg_first[segs[i].channel] = i;
    for (size_t c = h->n_channels; c-- > 0;) pl.seg_first[c] = std::min(pl.seg_first[c], pl.seg_first[c + 1]);
    GR4PM_TRY(upload_vec(pl.segs, segs, s));
    if (pl.ck.n < ck) GR4PM_TRY(pl.ck.alloc(static_cast<size_t>(ck) * 2));
    if (pl.seg_incr.n < n_segs) {
        GR4PM_TRY(pl.seg_incr.alloc(n_segs * 2));
        GR4PM_TRY(pl.seg_counter0.alloc(n_segs * 2));
    }
    // order[]: first the segments that depend on nothing before this call (a set_freq event starts them, or they are fixed
    // points), then the ones that continue the carried phasor (mode 0: at most one per channel); each part by descending
    // length, so that the long ones (a stream with missed detections) share waves
    unsigned n_indep = 0, n_writer = 0;
    bool dep_writes_state = false;
    {
        // ONE sort over 64-bit keys (part | longest first | position): this runs in the pipeline stage that makes the plans,
        // 10 000 segments a batch -- two stable sorts with indirect comparisons were half a millisecond of that stage
        static thread_local std::vector<unsigned> order;
        static thread_local std::vector<unsigned long long> keys;
        static const bool no_sort = gr4pm::experiment_env("GR4PM_ROT_NO_SORT", false) != nullptr;
        auto part = [&](unsigned a) { return segs[a].mode == 0 ? 2u : segs[a].last ? 1u : 0u; }; // indep | writer | dep
        keys.resize(n_segs);
        for (unsigned i = 0; i < n_segs; ++i) {
            const unsigned pt = part(i);
            n_indep += pt == 0;
            n_writer += pt == 1;
            dep_writes_state |= pt == 2 && segs[i].last;
            // (a segment is shorter than 2^36 items -- 2^33 checkpoint slots are 32-bit --, a call has fewer than 2^26 segments)
            const unsigned long long by_len = no_sort ? 0ull : (~segs[i].len & ((1ull << 36) - 1));
            keys[i] = (static_cast<unsigned long long>(pt) << 62) | (by_len << 26) | i;
        }
        std::sort(keys.begin(), keys.end());
        order.resize(n_segs);
        for (unsigned i = 0; i < n_segs; ++i) order[i] = static_cast<unsigned>(keys[i] & ((1u << 26) - 1));
        GR4PM_TRY(upload_vec(pl.order, order, s));
    }
    if (n_const) {
        static thread_local std::vector<unsigned> list;
        list.clear();
        for (unsigned i = 0; i < n_segs; ++i)
            if (segs[i].mode == 2) list.push_back(i);
        GR4PM_TRY(upload_vec(pl.const_list, list, s));
    }
    unsigned long long longest_const = 0;
    for (unsigned i = 0; i < n_segs; ++i)
        if (segs[i].mode == 2) longest_const = std::max(longest_const, segs[i].len);
    const RotState* st_in = h->state.p + static_cast<size_t>(h->st_cur) * h->n_channels;
    const int st_next = (h->st_cur + 1) % gr4pm_rotator::kStates;
    RotState* st_out = h->state.p + static_cast<size_t>(st_next) * h->n_channels;
    static const unsigned wg = gr4pm::experiment_env_wg("GR4PM_ROT_WG", 64u, 1u, 64u); // __launch_bounds__(64)
    // the entries of order[] in front of n_indep + n_writer start at an event (or are fixed points): k_rot_checkpoints_fresh;
    // GR4PM_ROT_GENERIC=1: the one kernel of rounds 1 - 5 for everything (A/B)
    static const bool generic_only = getenv("GR4PM_ROT_GENERIC") != nullptr;
    auto launch_chains = [&](hipStream_t on, unsigned first, unsigned count) {
        if (!count || timing_skip("rot")) return; // GR4PM_TIMING_SKIP: what a kernel costs the pipeline (results are garbage)
        const unsigned n_fresh = generic_only ? 0u : n_indep + n_writer;
        const unsigned fresh = first < n_fresh ? std::min(count, n_fresh - first) : 0u;
        if (fresh && count > fresh) {
            const unsigned fresh_blocks = grid_for(fresh, wg);
            hipLaunchKernelGGL(k_rot_checkpoints_both, dim3(fresh_blocks + grid_for(count - fresh, wg)), dim3(wg), 0, on,
                               pl.segs.p, fresh, fresh_blocks, count - fresh, st_in, st_out, pl.ck.p, pl.seg_incr.p,
                               pl.seg_counter0.p, pl.order.p + first);
        } else if (fresh) {
            hipLaunchKernelGGL(k_rot_checkpoints_fresh, dim3(grid_for(fresh, wg)), dim3(wg), 0, on, pl.segs.p, fresh, st_out,
                               pl.ck.p, pl.seg_incr.p, pl.seg_counter0.p, pl.order.p + first);
        } else {
            hipLaunchKernelGGL(k_rot_checkpoints, dim3(grid_for(count, wg)), dim3(wg), 0, on, pl.segs.p, count, st_in, st_out,
                               pl.ck.p, pl.seg_incr.p, pl.seg_counter0.p, pl.order.p + first);
        }
    };
    auto launch_const_fill = [&](hipStream_t on) {
        if (!n_const) return;
        const unsigned gx = static_cast<unsigned>(std::min<unsigned long long>((longest_const / kRotChunk + 255) / 256 + 1, 2048));
        for (size_t first = 0; first < n_const; first += 65535) {
            const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, n_const - first));
            hipLaunchKernelGGL(k_rot_const_fill, dim3(gx, rows), dim3(256), 0, on, pl.segs.p, pl.const_list.p + first, pl.ck.p);
        }
    };
    // (tests: GR4PM_TEST_ROT_DELAY_US holds every chain kernel of such a plan back by that long, so that a consumer that
    // does not wait for the plan's events reads checkpoints that are not there yet)
    const unsigned test_delay_us = h->test_delay_us;
    unsigned long long longest_chain = 0;
    for (unsigned i = 0; i < n_segs; ++i)
        if (segs[i].mode != 2) longest_chain = std::max(longest_chain, segs[i].len);
    const bool side_by_side = ring && (h->async_policy > 0 || (h->async_policy == 0 && longest_chain >= gr4pm_rotator::kAsyncMinItems));
    auto& sy = h->sync[h->plan_cur];
    if (side_by_side) {
        if (!h->async_ready) { // the handle's own streams (at the priority of the one it was given) and the plans' events
            int prio = 0;
            GR4PM_HIP_TRY(hipStreamGetPriority(s, &prio));
            for (auto& a : h->aux) GR4PM_HIP_TRY(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prio));
            for (auto& y : h->sync) {
                GR4PM_HIP_TRY(hipEventCreateWithFlags(&y.up, hipEventDisableTiming));
                GR4PM_HIP_TRY(hipEventCreateWithFlags(&y.indep, hipEventDisableTiming));
                GR4PM_HIP_TRY(hipEventCreateWithFlags(&y.writer, hipEventDisableTiming));
                GR4PM_HIP_TRY(hipEventCreateWithFlags(&y.dep, hipEventDisableTiming));
            }
            h->async_ready = true;
        }
        constexpr int K = gr4pm_rotator::kAux;
        const int turn = h->plan_cur % K;
        hipStream_t s_indep = h->aux[turn], s_writer = h->aux[K + turn], s_dep = h->aux[2 * K + turn];
        GR4PM_HIP_TRY(hipEventRecord(sy.up, s)); // tables of this plan on the device, and everything `s` carried before
        GR4PM_HIP_TRY(hipStreamWaitEvent(s_indep, sy.up, 0));
        if (test_delay_us) hipLaunchKernelGGL(k_test_delay, dim3(1), dim3(64), 0, s_indep, test_delay_us);
        launch_chains(s_indep, 0, n_indep);
        launch_const_fill(s_indep);
        GR4PM_HIP_TRY(hipEventRecord(sy.indep, s_indep));
        GR4PM_HIP_TRY(hipStreamWaitEvent(s_writer, sy.up, 0));
        if (test_delay_us) hipLaunchKernelGGL(k_test_delay, dim3(1), dim3(64), 0, s_writer, test_delay_us / 3);
        launch_chains(s_writer, n_indep, n_writer);
        GR4PM_HIP_TRY(hipEventRecord(sy.writer, s_writer));
        // the continuations read the carried phasor: behind the kernels of the plan before that wrote it
        GR4PM_HIP_TRY(hipStreamWaitEvent(s_dep, sy.up, 0));
        if (h->last_async_plan >= 0) {
            const auto& before = h->sync[h->last_async_plan];
            GR4PM_HIP_TRY(hipStreamWaitEvent(s_dep, before.writer, 0));
            if (before.dep_writes_state) GR4PM_HIP_TRY(hipStreamWaitEvent(s_dep, before.dep, 0));
        }
        if (test_delay_us) hipLaunchKernelGGL(k_test_delay, dim3(1), dim3(64), 0, s_dep, test_delay_us / 2);
        launch_chains(s_dep, n_indep + n_writer, n_segs - n_indep - n_writer);
        GR4PM_HIP_TRY(hipEventRecord(sy.dep, s_dep));
        sy.async = true;
        sy.dep_writes_state = dep_writes_state;
        h->last_async_plan = h->plan_cur;
    } else {
        if (h->last_async_plan >= 0) { // (a plain call behind ring plans: their kernels wrote the state this one reads)
            GR4PM_HIP_TRY(hipStreamWaitEvent(s, h->sync[h->last_async_plan].writer, 0));
            GR4PM_HIP_TRY(hipStreamWaitEvent(s, h->sync[h->last_async_plan].dep, 0));
            h->last_async_plan = -1;
        }
        launch_chains(s, 0, n_segs);
        launch_const_fill(s);
        sy.async = false;
    }
    GR4PM_HIP_TRY(hipGetLastError());
    h->st_cur = st_next;
    return GR4PM_OK;
}

extern "C" {

gr4pm_status gr4pm_rotator_process(gr4pm_rotator* h, const gr4pm_c64* in, size_t stride, size_t n,
                                   gr4pm_c64* out, const gr4pm_tag* tags, const uint32_t* tag_channel,
                                   size_t n_tags)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (n == 0) return GR4PM_OK; // an empty chunk is legal (and may come with null pointers)
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<RotSeg> segs;
    GR4PM_TRY(rotator_plan(h, n, tags, tag_channel, n_tags, segs));
    hipStream_t s = h->stream;
    const unsigned n_segs = static_cast<unsigned>(segs.size());
    {
        size_t longest = 0;
        for (const auto& g : segs) longest = std::max<size_t>(longest, g.len);
        const unsigned gx = static_cast<unsigned>(std::min<size_t>((longest + 255) / 256, 4096));
        // grid.y = segment (at most 65535 per launch)
        for (unsigned s0 = 0; s0 < n_segs; s0 += 65535u) {
            const unsigned ns = std::min(65535u, n_segs - s0);
            const auto& pl = h->plans[h->plan_cur];
            hipLaunchKernelGGL(k_rot_apply, dim3(gx, ns), dim3(256), 0, s, pl.segs.p + s0, ns, pl.ck.p,
                               pl.seg_incr.p + s0, pl.seg_counter0.p + s0, reinterpret_cast<const cf*>(in),
                               reinterpret_cast<cf*>(out), stride);
        }
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipStreamSynchronize(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"

// ------------------------------------------------------------------------ CostasLoop
struct gr4pm_costas_loop {
    double loop_bandwidth;
    int constellation;
    float k1, k2;
    size_t n_channels;
    hipStream_t stream;
    struct Memo {
        bool valid = false;
        double bw = 0.0;
        int constellation = 0;
        float k1 = 0.0f, k2 = 0.0f;
    } memo[4];
    unsigned memo_next = 0;
    DevBuf<CostasState> state; // [2][n_channels], st_cur selects the current half
    int st_cur = 0;
    int small_footprint = 0; // 0: k_costas<C, 8> (112 VGPRs, fastest alone), 1: k_costas<C, 2> (62), 2: k_costas_cap<C, 2> (32)
    DevBuf<CostasSeg> segs;
    DevBuf<CostasChain> chains;
    DevBuf<CostasPiece> pieces;
};

static void costas_coeffs(gr4pm_costas_loop* h)
{
    // tag-driven settings alternate between a handful of (bandwidth, constellation) pairs, three
    // times per packet: remember the last few results instead of redoing the cube roots
    for (const auto& m : h->memo)
        if (m.valid && m.bw == h->loop_bandwidth && m.constellation == h->constellation) {
            h->k1 = m.k1;
            h->k2 = m.k2;
            return;
        }
    // settingsChanged(), costas_loop.hpp:62-87
    double gain = 1.0;
    if (h->constellation == 2) gain = 1.41421356237309504880;
    const double bw = h->loop_bandwidth, bw2 = bw * bw, bw3 = bw2 * bw, bw4 = bw2 * bw2;
    const double s = std::cbrt(36.0 * bw2 +
                               std::sqrt(3.0) * std::sqrt(432.0 * bw4 + 848.0 * bw3 + 624.0 * bw2 +
                                                          204.0 * bw + 25.0) +
                               36.0 * bw + 9.0);
    const double z = -(-12.0 * bw - 6.0) / (3.0 * std::cbrt(6.0) * (2.0 * bw + 1.0) * s) +
                     (std::cbrt(2.0) * s) / (std::cbrt(9.0) * (2.0 * bw + 1.0)) - 1.0;
    h->k1 = static_cast<float>((1.0 - z * z) / gain);
    h->k2 = static_cast<float>(((1.0 - z) * (1.0 - z)) / gain);
    auto& slot = h->memo[h->memo_next++ % 4];
    slot = { true, h->loop_bandwidth, h->constellation, h->k1, h->k2 };
}

extern "C" {

gr4pm_status gr4pm_costas_loop_create(const gr4pm_costas_loop_params* p, gr4pm_costas_loop** out)
try {
    if (!p || !out || p->n_channels == 0 || p->constellation < 0 || p->constellation > 2)
        return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_costas_loop;
    if (!h) return GR4PM_ERR_NOMEM;
    h->loop_bandwidth = p->loop_bandwidth;
    h->constellation = p->constellation;
    h->n_channels = p->n_channels;
    h->stream = static_cast<hipStream_t>(p->stream);
    costas_coeffs(h);
    gr4pm_status s = h->state.alloc(static_cast<size_t>(gr4pm_rotator::kStates) * h->n_channels);
    if (s == GR4PM_OK) s = h->state.zero(h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    // GR4PM_COSTAS_FORM = 0 .. 2: the kernel form every CostasLoop starts with (tests and A/B; same results)
    static const char* form = gr4pm::experiment_env("GR4PM_COSTAS_FORM", false);
    if (form) h->small_footprint = std::min(2, std::max(0, atoi(form)));
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_costas_loop_destroy(gr4pm_costas_loop* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_costas_loop_reset(gr4pm_costas_loop* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    GR4PM_TRY(h->state.zero(h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    h->st_cur = 0;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_costas_loop_coeffs(const gr4pm_costas_loop* h, float* k1, float* k2)
try {
    *k1 = h->k1;
    *k2 = h->k2;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_costas_loop_set(gr4pm_costas_loop* h, double loop_bandwidth, int constellation)
try {
    if (!h || constellation < 0 || constellation > 2) return GR4PM_ERR_INVALID;
    h->loop_bandwidth = loop_bandwidth;
    h->constellation = constellation;
    costas_coeffs(h);
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

// n_of(c): items of channel c in this call (channels with 0 items keep their state)
extern "C++" {
template <typename NOf>
static gr4pm_status costas_process_impl(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t stride, NOf n_of,
                                        gr4pm_c64* out, const gr4pm_tag* tags, const uint32_t* tag_channel,
                                        size_t n_tags)
{
    std::vector<CostasSeg> segs;
    for (size_t c = 0; c < h->n_channels; ++c) {
        const size_t n = n_of(c);
        if (n == 0) { // a piece of length 0 that only hands the carried state on to the other slot
            CostasSeg g{};
            g.channel = static_cast<unsigned>(c);
            g.last = 1;
            segs.push_back(g);
            continue;
        }
        size_t pos = 0;
        int mode = 0;
        float phase0 = 0.0f;
        auto push = [&](size_t end) {
            if (end <= pos) return;
            CostasSeg g{};
            g.start = pos;
            g.len = static_cast<unsigned>(end - pos);
            g.channel = static_cast<unsigned>(c);
            g.mode = mode;
            g.phase0 = phase0;
            g.last = 0;
            segs.push_back(g);
            pos = end;
        };
        for (size_t t = 0; t < n_tags; ++t) {
            const size_t tc = tag_channel ? tag_channel[t] : 0;
            if (tc != c || !(tags[t].flags & GR4PM_TAG_SYNCWORD) || tags[t].index >= n) continue;
            const size_t i = static_cast<size_t>(tags[t].index);
            push(i);
            if (i == pos) { // set_phase at the head of the chunk, costas_loop.hpp:101-106
                mode = 1;
                phase0 = tags[t].phase;
            }
        }
        push(n);
        if (!segs.empty() && segs.back().channel == c) segs.back().last = 1;
    }
    hipStream_t s = h->stream;
    // A wave lives as long as its longest lane.  Segments are independent of one another (carried state travels through
    // the ping-pong state array, not through their order), so the longest ones are put together: a stream with missed
    // detections (segments that run through several packets: 64 channels of configs[2] hold ~80 of five packets'
    // length among 9700) then keeps two waves alive for the long tail instead of eighty.
    static const bool costas_no_sort = gr4pm::experiment_env("GR4PM_COSTAS_NO_SORT", false) != nullptr;
    if (!costas_no_sort)
        std::stable_sort(segs.begin(), segs.end(), [](const CostasSeg& a, const CostasSeg& b) { return a.len > b.len; });
    GR4PM_TRY(upload_vec(h->segs, segs, s));
    if (timing_skip("seg_stats")) { // GR4PM_TIMING_SKIP=seg_stats: what the serial kernel is given
        size_t longest = 0, total = 0;
        for (const auto& g : segs) longest = std::max<size_t>(longest, g.len), total += g.len;
        fprintf(stderr, "[gr4pm costas] %zu segments, %zu items, longest %zu\n", segs.size(), total, longest);
    }
    static const unsigned wg = gr4pm::experiment_env_wg("GR4PM_COSTAS_WG", 64u, 1u, 1024u);
    const dim3 grid(grid_for(segs.size(), wg)), block(wg);
    const unsigned n_segs = static_cast<unsigned>(segs.size());
    const CostasState* st_in = h->state.p + h->st_cur * h->n_channels;
    CostasState* st_out = h->state.p + (h->st_cur ^ 1) * h->n_channels;
    h->st_cur ^= 1;
    auto launch = [&](auto kernel) {
#ifdef GR4PM_EXPERIMENTS
        if (timing_skip("costas_fake")) { // GR4PM_FAKE=workgroups,ticks(10 ns),bytes of LDS
            unsigned wgs = grid.x, ticks = 83000u, lds = 0u;
            static const char* fake = gr4pm::experiment_env("GR4PM_FAKE", true);
            if (fake) sscanf(fake, "%u,%u,%u", &wgs, &ticks, &lds);
            wgs = std::max(wgs, 1u);
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_serial_fake),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            hipLaunchKernelGGL(k_serial_fake, dim3(wgs), block, lds, s, ticks, reinterpret_cast<float*>(st_out));
            return;
        }
#endif
        if (timing_skip("costas")) return;
        hipLaunchKernelGGL(kernel, grid, block, 0, s, h->segs.p, n_segs, st_in, st_out, h->k1, h->k2,
                           reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out), stride);
    };
    // Form 2 pays when the call is long enough for a correlator launch of the same size to keep the chip busy beside it:
    // a PLL wave lives for one packet's chain however small the call is (0.75 ms in the 112-VGPR form, 1.77 ms in the
    // 32-VGPR form), and with batches of 2^26 samples and less that life, not the correlator, is what the receiver waits
    // for (64 channels x 2^20 samples per batch: 40.8 against 25.5 Gsps sustained).  Below 2^25 symbols: the fast form.
    size_t call_symbols = 0;
    for (const auto& g : segs) call_symbols += g.len;
    static const char* cap_min = gr4pm::experiment_env("GR4PM_COSTAS_CAP_MIN_LOG2", false);
    const size_t cap_from = size_t{ 1 } << (cap_min ? std::min(40, std::max(0, atoi(cap_min))) : 25);
    if (h->small_footprint >= 2 && call_symbols >= cap_from) {
        if (h->constellation == 0) launch(k_costas_cap<0, 2>);
        else if (h->constellation == 1) launch(k_costas_cap<1, 2>);
        else launch(k_costas_cap<2, 2>);
    } else if (h->small_footprint == 1) {
        if (h->constellation == 0) launch(k_costas<0, 2>);
        else if (h->constellation == 1) launch(k_costas<1, 2>);
        else launch(k_costas<2, 2>);
    } else {
        if (h->constellation == 0) launch(k_costas<0, 8>);
        else if (h->constellation == 1) launch(k_costas<1, 8>);
        else launch(k_costas<2, 8>);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}
} // extern "C++"

gr4pm_status gr4pm_costas_loop_set_small_footprint(gr4pm_costas_loop* h, int on)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->small_footprint = on < 0 ? 0 : (on > 2 ? 2 : on);
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_costas_loop_process(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t stride, size_t n,
                                       gr4pm_c64* out, const gr4pm_tag* tags, const uint32_t* tag_channel,
                                       size_t n_tags)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (n == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    return costas_process_impl(h, in, stride, [n](size_t) { return n; }, out, tags, tag_channel, n_tags);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_costas_loop_process_ragged(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t stride,
                                              const size_t* n_per_channel, gr4pm_c64* out, const gr4pm_tag* tags,
                                              const uint32_t* tag_channel, size_t n_tags)
try {
    if (!h || !n_per_channel) return GR4PM_ERR_INVALID;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    return costas_process_impl(h, in, stride, [n_per_channel](size_t c) { return n_per_channel[c]; }, out, tags,
                               tag_channel, n_tags);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_costas_loop_process_packets(gr4pm_costas_loop* h, const gr4pm_c64* in, size_t n,
                                               gr4pm_c64* out, const gr4pm_packet_tag* tags, size_t n_tags)
try {
    return gr4pm::costas_loop_process_packets_from(h, in, nullptr, 0, n, out, tags, n_tags);
}
GR4PM_ABI_CATCH

} // extern "C"

// (library-internal: csrc/packet_receiver.hip) gr4pm_costas_loop_process_packets with the gather of the block in front
// folded in (round 6): the loop's input stream is not in memory as such -- item i of it is `in[spans[k].src + (i -
// spans[k].dst)]` for the span that holds i (PayloadMetadataInsert's span table, ascending, covering [0, n)).  Saves that
// block's gather: a read and a write of the whole symbol stream.  spans == nullptr: the stream is `in` itself.
gr4pm_status gr4pm::costas_loop_process_packets_from(gr4pm_costas_loop* h, const gr4pm_c64* in, const hostlogic::CopySpan* spans,
                                                     size_t n_spans, size_t n, gr4pm_c64* out, const gr4pm_packet_tag* tags,
                                                     size_t n_tags)
{
    if (!h) return GR4PM_ERR_INVALID;
    if (h->n_channels != 1) {
        set_error("process_packets needs a single-channel CostasLoop");
        return GR4PM_ERR_INVALID;
    }
    if (n == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<CostasChain> chains;
    std::vector<CostasPiece> pieces;
    CostasChain cur{};
    cur.piece0 = 0;
    cur.mode = 0;
    size_t pos = 0;
    size_t span_at = 0; // cursor into spans (pieces are closed in ascending order)
    auto close_piece = [&](size_t end) {
        while (pos < end) {
            size_t stop = end;
            long long in_off = 0;
            if (spans) {
                while (span_at < n_spans && spans[span_at].dst + spans[span_at].len <= pos) ++span_at;
                if (span_at >= n_spans || spans[span_at].dst > pos) { // (a hole in the table: the caller's error)
                    pos = end;
                    span_at = n_spans + 1;
                    return;
                }
                stop = std::min<size_t>(end, spans[span_at].dst + spans[span_at].len);
                in_off = static_cast<long long>(spans[span_at].src) - static_cast<long long>(spans[span_at].dst);
            }
            while (pos < stop) { // (len is 32 bits wide)
                const size_t m = std::min<size_t>(stop - pos, 1u << 30);
                CostasPiece pc{};
                pc.start = pos;
                pc.in_off = in_off;
                pc.len = static_cast<unsigned>(m);
                pc.constellation = h->constellation;
                pc.k1 = h->k1;
                pc.k2 = h->k2;
                pieces.push_back(pc);
                pos += m;
            }
        }
    };
    auto close_chain = [&]() {
        cur.n_pieces = static_cast<unsigned>(pieces.size()) - cur.piece0;
        if (cur.n_pieces) chains.push_back(cur);
        cur = CostasChain{};
        cur.piece0 = static_cast<unsigned>(pieces.size());
    };
    for (size_t t = 0; t < n_tags; ++t) {
        if (tags[t].index >= n) break;
        close_piece(static_cast<size_t>(tags[t].index));
        // keys naming settings are applied before the chunk, then settingsChanged(), :52-88
        bool changed = false;
        if (tags[t].constellation >= 0) {
            if (tags[t].constellation > 2) {
                set_error("constellation %d", tags[t].constellation);
                return GR4PM_ERR_INVALID;
            }
            h->constellation = tags[t].constellation;
            changed = true;
        }
        if (tags[t].loop_bandwidth >= 0.0) {
            h->loop_bandwidth = tags[t].loop_bandwidth;
            changed = true;
        }
        if (changed) costas_coeffs(h);
        if (tags[t].kind == GR4PM_PKT_SYNCWORD && (tags[t].syncword.flags & GR4PM_TAG_SYNCWORD)) { // :101-106
            close_chain();
            cur.mode = 1;
            cur.phase0 = tags[t].syncword.phase;
        }
    }
    close_piece(n);
    close_chain();
    if (span_at > n_spans) {
        set_error("process_packets: the span table does not cover the stream");
        return GR4PM_ERR_INVALID;
    }
    if (chains.empty()) return GR4PM_OK;
    chains.back().last = 1;
    hipStream_t s = h->stream;
    GR4PM_TRY(upload_vec(h->chains, chains, s));
    GR4PM_TRY(upload_vec(h->pieces, pieces, s));
    if (!timing_skip("costas_chains")) { // (EXPERIMENTS builds: GR4PM_TIMING_SKIP=costas_chains, wrong results)
        // the kernel form as in gr4pm_costas_loop_process: 32 VGPRs beside a correlator launch where the call is long enough
        static const char* cap_min = gr4pm::experiment_env("GR4PM_COSTAS_CAP_MIN_LOG2", false);
        const size_t cap_from = size_t{ 1 } << (cap_min ? std::min(40, std::max(0, atoi(cap_min))) : 25);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid_for(chains.size(), 64)), dim3(64), 0, s, h->chains.p,
                               static_cast<unsigned>(chains.size()), h->pieces.p, h->state.p + h->st_cur,
                               h->state.p + (h->st_cur ^ 1), reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out));
        };
        if (h->small_footprint >= 2 && n >= cap_from) launch(k_costas_chains_cap);
        else if (h->small_footprint == 1) launch(k_costas_chains<2>);
        else launch(k_costas_chains<8>);
    }
    h->st_cur ^= 1;
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}

// ------------------------------------------------------------------------ SyncwordWipeoff
struct gr4pm_syncword_wipeoff {
    std::vector<float> syncword;
    hipStream_t stream;
    DevBuf<float> d_syncword;
    DevBuf<WipeSpan> spans;
    bool in_syncword = false; // syncword_wipeoff.hpp:27-28
    size_t position = 0;
};

extern "C" {

gr4pm_status gr4pm_syncword_wipeoff_create(const gr4pm_syncword_wipeoff_params* p,
                                           gr4pm_syncword_wipeoff** out)
try {
    if (!p || !out || !p->syncword || p->n_syncword == 0) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_syncword_wipeoff;
    if (!h) return GR4PM_ERR_NOMEM;
    h->syncword.assign(p->syncword, p->syncword + p->n_syncword);
    h->stream = static_cast<hipStream_t>(p->stream);
    gr4pm_status s = h->d_syncword.alloc(p->n_syncword);
    if (s == GR4PM_OK) s = h->d_syncword.upload(h->syncword.data(), h->syncword.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_syncword_wipeoff_destroy(gr4pm_syncword_wipeoff* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_syncword_wipeoff_reset(gr4pm_syncword_wipeoff* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->in_syncword = false;
    h->position = 0;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"

// replay of syncword_wipeoff.hpp:53-75 over the tag list of one call: the spans of the syncword inside it
// (base: offset of the channel's item 0 in the buffer the kernel indexes)
static void wipe_replay(gr4pm_syncword_wipeoff* h, size_t n, const gr4pm_tag* tags, size_t n_tags, size_t base,
                        std::vector<WipeSpan>& spans)
{
    size_t pos = 0, t = 0;
    const size_t L = h->syncword.size();
    while (pos < n) {
        while (t < n_tags && tags[t].index < pos) ++t;
        const bool has_tag = t < n_tags && tags[t].index == pos && (tags[t].flags & GR4PM_TAG_SYNCWORD);
        if (!h->in_syncword && has_tag) {
            h->in_syncword = true;
            h->position = 0;
        }
        size_t end = n;
        for (size_t u = t; u < n_tags; ++u)
            if (tags[u].index > pos) {
                end = std::min<size_t>(end, tags[u].index);
                break;
            }
        if (h->in_syncword) {
            const size_t m = std::min(end - pos, L - h->position);
            spans.push_back({ base + pos, static_cast<unsigned>(h->position), static_cast<unsigned>(m) });
            h->position += m;
            if (h->position == L) h->in_syncword = false;
        }
        pos = end;
        if (t < n_tags && tags[t].index < pos) ++t;
    }
}

extern "C" {

gr4pm_status gr4pm_sincosf(const float* x, size_t n, float* sin_out, float* cos_out)
try {
    if (!x || !sin_out || !cos_out) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    if (n == 0) return GR4PM_OK;
    hipLaunchKernelGGL(k_sincosf, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, x, n, sin_out,
                       cos_out);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipStreamSynchronize(nullptr));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_costas_phase_wrap(const float* x, size_t n, float* out)
try {
    if (!x || !out) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    if (n == 0) return GR4PM_OK;
    hipLaunchKernelGGL(k_costas_wrap, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, x, n, out);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipStreamSynchronize(nullptr));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_wipeoff_process(gr4pm_syncword_wipeoff* h, const gr4pm_c64* in, size_t n,
                                            gr4pm_c64* out, const gr4pm_tag* tags, size_t n_tags)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (n == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<WipeSpan> spans;
    wipe_replay(h, n, tags, n_tags, 0, spans);
    hipStream_t s = h->stream;
    if (in != out) // in place: only the syncword spans are touched
        hipLaunchKernelGGL(k_copy<cf>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s,
                           reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out), n);
    if (!spans.empty()) {
        GR4PM_TRY(upload_vec(h->spans, spans, s));
        hipLaunchKernelGGL(k_wipe, dim3(static_cast<unsigned>(spans.size())), dim3(64), 0, s, h->spans.p,
                           h->d_syncword.p, reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out));
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_wipeoff_process_channels(gr4pm_syncword_wipeoff* const* h, size_t n_channels,
                                                     gr4pm_c64* buf, size_t stride, const size_t* n,
                                                     const gr4pm_tag* const* tags, const size_t* n_tags)
try {
    if (!h || n_channels == 0 || !n || !tags || !n_tags) return GR4PM_ERR_INVALID;
    if (!buf) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<WipeSpan> spans;
    for (size_t c = 0; c < n_channels; ++c) {
        if (!h[c] || h[c]->syncword != h[0]->syncword || n[c] > stride) {
            set_error("a launch that spans channels needs wipe-off blocks of one syncword and n <= stride");
            return GR4PM_ERR_INVALID;
        }
        wipe_replay(h[c], n[c], tags[c], n_tags[c], c * stride, spans);
    }
    hipStream_t s = h[0]->stream;
    if (!spans.empty()) {
        GR4PM_TRY(upload_vec(h[0]->spans, spans, s));
        hipLaunchKernelGGL(k_wipe, dim3(static_cast<unsigned>(spans.size())), dim3(64), 0, s, h[0]->spans.p,
                           h[0]->d_syncword.p, reinterpret_cast<const cf*>(buf), reinterpret_cast<cf*>(buf));
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"

// ------------------------------------------------------------------ SyncwordDetectionFilter
// the state machine itself: hostlogic/sdf_gate.hpp (no HIP; also built with sanitizers by tests/hostlogic/)
struct gr4pm_syncword_detection_filter : gr4pm::hostlogic::SdfState {
    hipStream_t stream = nullptr;
};

extern "C" {

gr4pm_status gr4pm_syncword_detection_filter_create(const gr4pm_syncword_detection_filter_params* p,
                                                    gr4pm_syncword_detection_filter** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_syncword_detection_filter;
    if (!h) return GR4PM_ERR_NOMEM;
    h->sps = p->samples_per_symbol;
    h->syncword_size = p->syncword_size;
    h->header_size = p->header_size;
    h->stream = static_cast<hipStream_t>(p->stream);
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_syncword_detection_filter_destroy(gr4pm_syncword_detection_filter* h)
try {
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_syncword_detection_filter_reset(gr4pm_syncword_detection_filter* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->in_packet = false; // start(), :52
    h->gate_in_packet = false;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_detection_filter_process(gr4pm_syncword_detection_filter* h, const gr4pm_c64* in,
                                                     size_t n_in, gr4pm_c64* out, size_t out_cap,
                                                     int head_tag_flags, const gr4pm_header_msg* headers,
                                                     size_t n_headers, size_t n_ignored, size_t* consumed_,
                                                     size_t* headers_consumed, size_t* ignored_consumed,
                                                     int* tag_out_flags)
try {
    if (!h || !consumed_ || !headers_consumed || !ignored_consumed || !tag_out_flags) return GR4PM_ERR_INVALID;
    gr4pm::hostlogic::CopySpan runs[2];
    int n_runs = 0;
    GR4PM_TRY(gr4pm::hostlogic::sdf_process_plan(*h, n_in, out_cap, head_tag_flags, headers, n_headers, n_ignored,
                                                 consumed_, headers_consumed, ignored_consumed, tag_out_flags, runs,
                                                 &n_runs));
    for (int r = 0; r < n_runs; ++r)
        GR4PM_HIP_TRY(hipMemcpyAsync(out + runs[r].dst, in + runs[r].src, runs[r].len * sizeof(gr4pm_c64),
                                     hipMemcpyDeviceToDevice, h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"

extern "C" gr4pm_status gr4pm_syncword_detection_filter_gate(gr4pm_syncword_detection_filter* h,
                                                             const uint64_t* tag_index, size_t n_tags,
                                                             const gr4pm_header_msg* headers, size_t n_headers,
                                                             int headers_per_tag, uint8_t* accepted,
                                                             size_t* headers_used)
try {
    if (!h || !accepted || !headers_used) return GR4PM_ERR_INVALID;
    return gr4pm::hostlogic::sdf_gate(*h, tag_index, n_tags, headers, n_headers, headers_per_tag, accepted, headers_used);
}
GR4PM_ABI_CATCH

extern "C" gr4pm_status gr4pm_syncword_detection_filter_gate_resolve(gr4pm_syncword_detection_filter* h,
                                                                     const gr4pm_header_msg* msg)
try {
    if (!h || !msg) return GR4PM_ERR_INVALID;
    return gr4pm::hostlogic::sdf_gate_resolve(*h, *msg);
}
GR4PM_ABI_CATCH

// ------------------------------------------------------------------ InterpolatingFirFilter
struct gr4pm_interp_fir {
    size_t L, n_taps;
    int item_kind;
    unsigned cap, arm_stride;
    hipStream_t stream;
    DevBuf<float> taps;
    DevBuf<unsigned> arm_len;
    DevBuf<char> carry[2];
    int cur = 0;
};

template <typename T>
static gr4pm_status interp_fir_run(gr4pm_interp_fir* h, const void* in, size_t n_in, void* out)
{
    hipStream_t s = h->stream;
    const size_t n_out = n_in * h->L;
    const T* carry = reinterpret_cast<const T*>(h->carry[h->cur].p);
    T* carry_next = reinterpret_cast<T*>(h->carry[h->cur ^ 1].p);
    (void)n_out;
    const size_t smem = interp_fir_smem(h->L, h->arm_stride, sizeof(T));
    if (smem > kFirMaxSmem) {
        set_error("InterpolatingFirFilter: %zu taps x %zu arms need %zu bytes of LDS per workgroup (limit %zu)",
                  static_cast<size_t>(h->arm_stride), static_cast<size_t>(h->L), smem, kFirMaxSmem);
        return GR4PM_ERR_INVALID;
    }
    if (smem > 48 * 1024) // beyond the default dynamic-LDS window
        GR4PM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_interp_fir<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
    hipLaunchKernelGGL(k_interp_fir<T>, dim3(grid_for(n_in, kFirItems, 65536)), dim3(kFirItems), smem, s,
                       static_cast<const T*>(in), carry, h->cap, h->taps.p, h->arm_len.p, h->arm_stride,
                       static_cast<unsigned>(h->L), n_in, static_cast<T*>(out));
    hipLaunchKernelGGL(k_update_hist<T>, dim3((h->cap + 63) / 64), dim3(64), 0, s, static_cast<const T*>(in),
                       carry, carry_next, h->cap, n_in);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipStreamSynchronize(s));
    h->cur ^= 1;
    return GR4PM_OK;
}

extern "C" {

gr4pm_status gr4pm_interp_fir_create(const gr4pm_interp_fir_params* p, gr4pm_interp_fir** out)
try {
    if (!p || !out || !p->taps) return GR4PM_ERR_INVALID;
    *out = nullptr;
    if (p->interpolation == 0) { // interpolating_fir_filter.hpp:45-47
        set_error("interpolation cannot be zero");
        return GR4PM_ERR_INVALID;
    }
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_interp_fir;
    if (!h) return GR4PM_ERR_NOMEM;
    h->L = p->interpolation;
    h->n_taps = p->n_taps;
    h->item_kind = p->item_kind;
    h->stream = static_cast<hipStream_t>(p->stream);
    const size_t arm_max = (p->n_taps + h->L - 1) / h->L;
    h->arm_stride = static_cast<unsigned>(std::max<size_t>(arm_max, 1));
    h->cap = static_cast<unsigned>(bit_ceil_sz(std::max<size_t>(arm_max, 1))); // :63-64
    std::vector<float> taps(h->L * h->arm_stride, 0.0f);
    std::vector<unsigned> arm_len(h->L, 0);
    for (size_t j = 0; j < h->L; ++j)
        for (size_t k = j; k < p->n_taps; k += h->L) taps[j * h->arm_stride + arm_len[j]++] = p->taps[k]; // :54-60
    const size_t isz = p->item_kind == 0 ? sizeof(cf) : sizeof(float);
    gr4pm_status s = h->taps.alloc(taps.size());
    if (s == GR4PM_OK) s = h->arm_len.alloc(arm_len.size());
    for (int i = 0; i < 2 && s == GR4PM_OK; ++i) {
        s = h->carry[i].alloc(h->cap * isz);
        if (s == GR4PM_OK) s = h->carry[i].zero(h->stream);
    }
    if (s == GR4PM_OK) s = h->taps.upload(taps.data(), taps.size(), h->stream);
    if (s == GR4PM_OK) s = h->arm_len.upload(arm_len.data(), arm_len.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_interp_fir_destroy(gr4pm_interp_fir* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_interp_fir_reset(gr4pm_interp_fir* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    for (int i = 0; i < 2; ++i) GR4PM_TRY(h->carry[i].zero(h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
gr4pm_status gr4pm_interp_fir_process(gr4pm_interp_fir* h, const void* in, size_t n_in, void* out)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (n_in == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    return h->item_kind == 0 ? interp_fir_run<cf>(h, in, n_in, out) : interp_fir_run<float>(h, in, n_in, out);
}
GR4PM_ABI_CATCH

} // extern "C"

// ------------------------------------------------------------------ SymbolFilter
struct gr4pm_symbol_filter : gr4pm::hostlogic::SymfHostState { // the tag-driven state: hostlogic/symbol_filter_replay.hpp
    size_t arm_size;
    int item_kind;
    unsigned cap;
    hipStream_t stream;
    DevBuf<float> taps;
    DevBuf<char> carry[2];
    DevBuf<SymRun> runs;
    DevBuf<SymWg> wg_plan;
    int cur = 0;
};

extern "C" {

gr4pm_status gr4pm_symbol_filter_create(const gr4pm_symbol_filter_params* p, gr4pm_symbol_filter** out)
try {
    if (!p || !out || !p->taps) return GR4PM_ERR_INVALID;
    *out = nullptr;
    if (p->samples_per_symbol == 0) { // symbol_filter.hpp:67-69
        set_error("samples_per_symbol cannot be zero");
        return GR4PM_ERR_INVALID;
    }
    if (p->num_arms == 0) { // :71-73
        set_error("num_arms cannot be zero");
        return GR4PM_ERR_INVALID;
    }
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_symbol_filter;
    if (!h) return GR4PM_ERR_NOMEM;
    h->sps = p->samples_per_symbol;
    h->num_arms = p->num_arms;
    h->delay = p->delay;
    h->item_kind = p->item_kind;
    h->stream = static_cast<hipStream_t>(p->stream);
    // polyphase split, :84-90; inner products run over arm 0's length for every arm is NOT
    // what the reference does: each arm has its own length (taps[k], k = j, j+arms, ...)
    h->arm_size = (p->n_taps + p->num_arms - 1) / p->num_arms;
    std::vector<float> taps(h->num_arms * h->arm_size, 0.0f); // shorter arms zero padded
    for (size_t j = 0; j < h->num_arms; ++j) {
        size_t m = 0;
        for (size_t k = j; k < p->n_taps; k += h->num_arms) taps[j * h->arm_size + m++] = p->taps[k];
    }
    const size_t arm0 = (p->n_taps + p->num_arms - 1) / p->num_arms; // _taps[0].size(), :93
    h->cap = static_cast<unsigned>(bit_ceil_sz(std::max<size_t>(arm0, 1)));
    h->reset_clock_phase = (h->sps - (h->delay % h->sps)) % h->sps; // :106-107
    const size_t isz = p->item_kind == 0 ? sizeof(cf) : sizeof(float);
    gr4pm_status s = h->taps.alloc(taps.size());
    for (int i = 0; i < 2 && s == GR4PM_OK; ++i) {
        s = h->carry[i].alloc(h->cap * isz);
        if (s == GR4PM_OK) s = h->carry[i].zero(h->stream);
    }
    if (s == GR4PM_OK) s = h->taps.upload(taps.data(), taps.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_symbol_filter_destroy(gr4pm_symbol_filter* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_symbol_filter_reset(gr4pm_symbol_filter* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->clock_phase = 0; // start(), :110
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"

using hostlogic::SymReplay; // the host replay of symbol_filter.hpp:130-238: hostlogic/symbol_filter_replay.hpp
using hostlogic::symf_replay;

// fuse == nullptr: plain SymbolFilter.  Otherwise the input is rotated by the CFC plan on the fly.
static gr4pm_status symbol_filter_impl(gr4pm_symbol_filter* h, const void* in, size_t n_in, void* out,
                                       size_t out_cap, const gr4pm_tag* tags_in, size_t n_tags_in,
                                       gr4pm_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                                       size_t* consumed_, size_t* produced_, const CfcDev* fuse)
{
    if (!h || !consumed_ || !produced_) return GR4PM_ERR_INVALID;
    *consumed_ = *produced_ = 0;
    if (n_tags_out) *n_tags_out = 0;
    if (n_in == 0) return GR4PM_OK; // nothing consumed, nothing produced, queued tags keep waiting
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    const size_t sps = h->sps;
    SymReplay rp;
    symf_replay(*h, n_in, out_cap, tags_in, n_tags_in, tags_out, tags_cap, rp);
    std::vector<SymRun>& runs = rp.runs;
    const size_t pos = rp.pos, produced = rp.produced, n_pub = rp.n_pub;
    const bool tag_overflow = rp.tag_overflow;
    hipStream_t s = h->stream;
    if (!runs.empty()) {
        unsigned n_wg = 0;
        const unsigned per_wg = symf_per_wg(fuse != nullptr, sps, h->arm_size, h->item_kind == 0);
        for (auto& r : runs) { // workgroups never straddle runs
            r.wg0 = n_wg;
            r.chan = 0;
            n_wg += (r.count + per_wg - 1u) / per_wg;
        }
        GR4PM_TRY(upload_vec(h->runs, runs, s));
        const size_t pitch = (kSymPerWg * sps + h->arm_size) / sps + 2;
        if (h->wg_plan.n < n_wg) GR4PM_TRY(h->wg_plan.alloc(static_cast<size_t>(n_wg) * 2));
        const size_t arm_bytes = ((h->arm_size + 1) & ~size_t{ 1 }) * sizeof(float);
        const unsigned n_runs = static_cast<unsigned>(runs.size());
        if (fuse)
            launch_symbol_filter<cf, true>(
                s, n_wg, pitch * sps * sizeof(cf) + arm_bytes,
                static_cast<unsigned>(sps), static_cast<const cf*>(in),
                reinterpret_cast<const cf*>(h->carry[h->cur].p), h->cap, h->taps.p,
                static_cast<unsigned>(h->arm_size), h->runs.p, n_runs, h->wg_plan.p, static_cast<cf*>(out), *fuse);
        else if (h->item_kind == 0)
            launch_symbol_filter<cf, false>(s, n_wg, pitch * sps * sizeof(cf) + arm_bytes, static_cast<unsigned>(sps),
                                            static_cast<const cf*>(in),
                                            reinterpret_cast<const cf*>(h->carry[h->cur].p), h->cap, h->taps.p,
                                            static_cast<unsigned>(h->arm_size), h->runs.p, n_runs, h->wg_plan.p,
                                            static_cast<cf*>(out), CfcDev{});
        else
            launch_symbol_filter<float, false>(s, n_wg, pitch * sps * sizeof(float) + arm_bytes,
                                               static_cast<unsigned>(sps), static_cast<const float*>(in),
                                               reinterpret_cast<const float*>(h->carry[h->cur].p), h->cap, h->taps.p,
                                               static_cast<unsigned>(h->arm_size), h->runs.p, n_runs, h->wg_plan.p,
                                               static_cast<float*>(out), CfcDev{});
    }
    if (pos > 0) {
        if (fuse)
            hipLaunchKernelGGL(k_update_hist_cfc, dim3((h->cap + 63) / 64), dim3(64), 0, s,
                               static_cast<const cf*>(in), reinterpret_cast<const cf*>(h->carry[h->cur].p),
                               reinterpret_cast<cf*>(h->carry[h->cur ^ 1].p), h->cap, pos, *fuse);
        else if (h->item_kind == 0)
            hipLaunchKernelGGL(k_update_hist<cf>, dim3((h->cap + 63) / 64), dim3(64), 0, s,
                               static_cast<const cf*>(in), reinterpret_cast<const cf*>(h->carry[h->cur].p),
                               reinterpret_cast<cf*>(h->carry[h->cur ^ 1].p), h->cap, pos);
        else
            hipLaunchKernelGGL(k_update_hist<float>, dim3((h->cap + 63) / 64), dim3(64), 0, s,
                               static_cast<const float*>(in),
                               reinterpret_cast<const float*>(h->carry[h->cur].p),
                               reinterpret_cast<float*>(h->carry[h->cur ^ 1].p), h->cap, pos);
        h->cur ^= 1;
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    *consumed_ = pos;
    *produced_ = produced;
    if (n_tags_out) *n_tags_out = n_pub;
    if (tag_overflow) {
        set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}

extern "C" {

gr4pm_status gr4pm_symbol_filter_process(gr4pm_symbol_filter* h, const void* in, size_t n_in, void* out,
                                         size_t out_cap, const gr4pm_tag* tags_in, size_t n_tags_in,
                                         gr4pm_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                                         size_t* consumed, size_t* produced)
try {
    return symbol_filter_impl(h, in, n_in, out, out_cap, tags_in, n_tags_in, tags_out, tags_cap, n_tags_out,
                              consumed, produced, nullptr);
}
GR4PM_ABI_CATCH

// the chains of a ring plan run on the rotator's own streams (gr4pm_rotator::PlanSync): what reads its checkpoints waits
// for them on its own stream, not on the host
static gr4pm_status cfc_wait_plan(gr4pm_rotator* cfc, int plan, hipStream_t consumer)
{
    const auto& y = cfc->sync[plan];
    if (!y.async) return GR4PM_OK;
    GR4PM_HIP_TRY(hipStreamWaitEvent(consumer, y.indep, 0));
    GR4PM_HIP_TRY(hipStreamWaitEvent(consumer, y.writer, 0));
    GR4PM_HIP_TRY(hipStreamWaitEvent(consumer, y.dep, 0));
    return GR4PM_OK;
}

static gr4pm_status cfc_plan_impl(gr4pm_rotator* cfc, size_t n_in, const gr4pm_tag* tags_in,
                                  const uint32_t* tag_channel, size_t n_tags_in, int* plan, bool ring)
{
    if (!cfc || !plan) return GR4PM_ERR_INVALID;
    *plan = -1;
    if (cfc->mode != 1) {
        set_error("fused call needs a CoarseFrequencyCorrection handle");
        return GR4PM_ERR_INVALID;
    }
    if (!tag_channel && cfc->n_channels != 1 && n_tags_in) {
        set_error("tag_channel is required with more than one channel");
        return GR4PM_ERR_INVALID;
    }
    if (n_in == 0) return GR4PM_OK;
    std::vector<RotSeg> segs;
    GR4PM_TRY(rotator_plan(cfc, n_in, tags_in, tag_channel, n_tags_in, segs, ring)); // checkpoints on the CFC's stream
    *plan = cfc->plan_cur;
    GR4PM_HIP_TRY(final_sync(cfc->stream));
    return GR4PM_OK;
}

gr4pm_status gr4pm_cfc_symbol_filter_plan_channels(gr4pm_rotator* cfc, size_t n_in, const gr4pm_tag* tags_in,
                                                   const uint32_t* tag_channel, size_t n_tags_in, int* plan)
try {
    return cfc_plan_impl(cfc, n_in, tags_in, tag_channel, n_tags_in, plan, true);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_cfc_symbol_filter_plan(gr4pm_rotator* cfc, size_t n_in, const gr4pm_tag* tags_in,
                                          size_t n_tags_in, int* plan)
try {
    if (cfc && cfc->n_channels != 1) {
        set_error("fused call needs a single-channel CoarseFrequencyCorrection (or ..._plan_channels)");
        return GR4PM_ERR_INVALID;
    }
    return gr4pm_cfc_symbol_filter_plan_channels(cfc, n_in, tags_in, nullptr, n_tags_in, plan);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_cfc_symbol_filter_run(gr4pm_rotator* cfc, int plan, gr4pm_symbol_filter* sf,
                                         const gr4pm_c64* in, size_t n_in, gr4pm_c64* out, size_t out_cap,
                                         const gr4pm_tag* tags_in, size_t n_tags_in, gr4pm_tag* tags_out,
                                         size_t tags_cap, size_t* n_tags_out, size_t* consumed, size_t* produced)
try {
    return gr4pm_cfc_symbol_filter_run_channel(cfc, plan, 0, sf, in, n_in, out, out_cap, tags_in, n_tags_in, tags_out,
                                               tags_cap, n_tags_out, consumed, produced);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_cfc_symbol_filter_run_channel(gr4pm_rotator* cfc, int plan, size_t channel, gr4pm_symbol_filter* sf,
                                                 const gr4pm_c64* in, size_t n_in, gr4pm_c64* out, size_t out_cap,
                                                 const gr4pm_tag* tags_in, size_t n_tags_in, gr4pm_tag* tags_out,
                                                 size_t tags_cap, size_t* n_tags_out, size_t* consumed,
                                                 size_t* produced)
try {
    if (!cfc || !sf || !consumed || !produced) return GR4PM_ERR_INVALID;
    *consumed = *produced = 0;
    if (n_tags_out) *n_tags_out = 0;
    if (sf->item_kind != 0) {
        set_error("fused call needs a complex SymbolFilter");
        return GR4PM_ERR_INVALID;
    }
    if (n_in == 0) return GR4PM_OK;
    if (plan < 0 || plan >= GR4PM_CFC_PLANS || cfc->plans[plan].n_in != n_in) {
        set_error("no rotation plan for this call");
        return GR4PM_ERR_INVALID;
    }
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    // the rotation plan covers all n_in items, so the filter must be able to consume them all
    if (out_cap < n_in / sf->sps + n_tags_in + 2) {
        set_error("out_cap too small for a fused call");
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    const auto& pl = cfc->plans[plan];
    if (channel >= cfc->n_channels || pl.seg_first.size() != cfc->n_channels + 1) {
        set_error("channel %zu outside the rotation plan", channel);
        return GR4PM_ERR_INVALID;
    }
    GR4PM_TRY(cfc_wait_plan(cfc, plan, sf->stream));
    // the channel's own segments (they tile its [0, n_in)); checkpoint slots are plan-wide
    const unsigned first = pl.seg_first[channel];
    CfcDev f;
    f.segs = pl.segs.p + first;
    f.ck = pl.ck.p;
    f.seg_incr = pl.seg_incr.p + first;
    f.seg_counter0 = pl.seg_counter0.p + first;
    f.n_segs = pl.seg_first[channel + 1] - first;
    const gr4pm_status st = symbol_filter_impl(sf, in, n_in, out, out_cap, tags_in, n_tags_in, tags_out, tags_cap,
                                               n_tags_out, consumed, produced, &f);
    if (st == GR4PM_OK && *consumed != n_in) {
        set_error("fused call consumed %zu of %zu items", *consumed, n_in);
        return GR4PM_ERR_INVALID;
    }
    return st;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_cfc_symbol_filter_run_channels(gr4pm_rotator* cfc, int plan, gr4pm_symbol_filter* const* sf,
                                                  size_t n_channels, const gr4pm_c64* in, size_t in_stride, size_t n_in,
                                                  gr4pm_c64* out, size_t out_stride, const gr4pm_tag* const* tags_in,
                                                  const size_t* n_tags_in, gr4pm_tag* const* tags_out, size_t tags_cap,
                                                  size_t* n_tags_out, size_t* produced, const gr4pm_c64* head,
                                                  size_t head_stride, size_t n_head)
try {
    if (!cfc || !sf || !n_tags_in || !tags_in || !tags_out || !n_tags_out || !produced || n_channels == 0)
        return GR4PM_ERR_INVALID;
    for (size_t c = 0; c < n_channels; ++c) n_tags_out[c] = produced[c] = 0;
    if (n_in == 0) return GR4PM_OK;
    if (plan < 0 || plan >= GR4PM_CFC_PLANS || cfc->plans[plan].n_in != n_in) {
        set_error("no rotation plan for this call");
        return GR4PM_ERR_INVALID;
    }
    const auto& pl = cfc->plans[plan];
    if (n_channels != cfc->n_channels || pl.seg_first.size() != cfc->n_channels + 1) {
        set_error("the rotation plan has %zu channels, the call %zu", cfc->n_channels, n_channels);
        return GR4PM_ERR_INVALID;
    }
    if (!in || !out || in_stride + n_head < n_in || (n_head && (!head || head_stride < n_head || n_head > n_in))) {
        set_error("null sample pointer, in_stride + n_head < n_in or a bad head");
        return GR4PM_ERR_INVALID;
    }
    gr4pm_symbol_filter* h0 = sf[0];
    size_t max_tags = 0;
    for (size_t c = 0; c < n_channels; ++c) {
        const gr4pm_symbol_filter* h = sf[c];
        if (!h || h->item_kind != 0 || h->sps != h0->sps || h->arm_size != h0->arm_size || h->cap != h0->cap ||
            h->num_arms != h0->num_arms) {
            set_error("a launch that spans channels needs complex SymbolFilters of one design");
            return GR4PM_ERR_INVALID;
        }
        max_tags = std::max(max_tags, n_tags_in[c]);
    }
    // the rotation plan covers all n_in items, so every filter must be able to consume them all
    if (out_stride < n_in / h0->sps + max_tags + 2) {
        set_error("out_stride too small for a fused call");
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    // host replay per channel; the runs of all channels in one table
    static_assert(sizeof(SymChan) % 8 == 0 && sizeof(SymRun) % 8 == 0, "table layout");
    std::vector<SymChan> chans(n_channels);
    std::vector<SymRun> runs;
    unsigned n_wg = 0;
    const unsigned per_wg = symf_per_wg(true, h0->sps, h0->arm_size);
    bool overflow = false;
    for (size_t c = 0; c < n_channels; ++c) {
        gr4pm_symbol_filter* h = sf[c];
        SymReplay rp;
        symf_replay(*h, n_in, out_stride, tags_in[c], n_tags_in[c], tags_out[c], tags_cap, rp);
        if (rp.pos != n_in) {
            set_error("fused call consumed %zu of %zu items (channel %zu)", rp.pos, n_in, c);
            return GR4PM_ERR_INVALID;
        }
        overflow |= rp.tag_overflow;
        for (auto& r : rp.runs) { // workgroups never straddle runs
            r.wg0 = n_wg;
            r.chan = static_cast<unsigned>(c);
            n_wg += (r.count + per_wg - 1u) / per_wg;
            runs.push_back(r);
        }
        const unsigned first = pl.seg_first[c];
        SymChan& d = chans[c];
        d.in = reinterpret_cast<const cf*>(in) + c * in_stride;
        d.carry = reinterpret_cast<const cf*>(h->carry[h->cur].p);
        d.carry_next = reinterpret_cast<cf*>(h->carry[h->cur ^ 1].p);
        d.out = reinterpret_cast<cf*>(out) + c * out_stride;
        d.segs = pl.segs.p + first;
        d.seg_incr = pl.seg_incr.p + first;
        d.seg_counter0 = pl.seg_counter0.p + first;
        d.n_segs = pl.seg_first[c + 1] - first;
        d.pad = 0;
        d.n = rp.pos;
        d.head = n_head ? reinterpret_cast<const cf*>(head) + c * head_stride : nullptr;
        d.n_head = n_head;
        h->cur ^= 1;
        produced[c] = rp.produced;
        n_tags_out[c] = rp.n_pub;
    }
    hipStream_t s = h0->stream;
    GR4PM_TRY(cfc_wait_plan(cfc, plan, s));
    const size_t chan_words = n_channels * sizeof(SymChan) / 8, run_words = runs.size() * sizeof(SymRun) / 8;
    cfc->mc_host.resize(chan_words + run_words);
    std::memcpy(cfc->mc_host.data(), chans.data(), chan_words * 8);
    if (run_words) std::memcpy(cfc->mc_host.data() + chan_words, runs.data(), run_words * 8);
    GR4PM_TRY(upload_vec(cfc->mc_tab, cfc->mc_host, s));
    const SymChan* d_chans = reinterpret_cast<const SymChan*>(cfc->mc_tab.p);
    const SymRun* d_runs = reinterpret_cast<const SymRun*>(cfc->mc_tab.p + chan_words);
    CfcDev f{};
    f.ck = pl.ck.p;
    f.n_segs = 1; // per channel from the table
    if (n_wg) {
        if (cfc->mc_wg.n < n_wg) GR4PM_TRY(cfc->mc_wg.alloc(static_cast<size_t>(n_wg) * 2));
        const size_t sps = h0->sps;
        const size_t pitch = (kSymPerWg * sps + h0->arm_size) / sps + 2;
        const size_t arm_bytes = ((h0->arm_size + 1) & ~size_t{ 1 }) * sizeof(float);
        launch_symbol_filter<cf, true>(
            s, n_wg, pitch * sps * sizeof(cf) + arm_bytes,
            static_cast<unsigned>(sps), static_cast<const cf*>(nullptr), static_cast<const cf*>(nullptr), h0->cap,
            h0->taps.p, static_cast<unsigned>(h0->arm_size), d_runs, static_cast<unsigned>(runs.size()), cfc->mc_wg.p,
            static_cast<cf*>(nullptr), f, d_chans);
    }
    hipLaunchKernelGGL(k_update_hist_cfc_channels, dim3((h0->cap + 63) / 64, static_cast<unsigned>(n_channels)),
                       dim3(64), 0, s, d_chans, pl.ck.p, h0->cap);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    if (overflow) {
        set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_cfc_symbol_filter_process(gr4pm_rotator* cfc, gr4pm_symbol_filter* sf, const gr4pm_c64* in,
                                             size_t n_in, gr4pm_c64* out, size_t out_cap,
                                             const gr4pm_tag* tags_in, size_t n_tags_in, gr4pm_tag* tags_out,
                                             size_t tags_cap, size_t* n_tags_out, size_t* consumed,
                                             size_t* produced)
try {
    if (!cfc || !sf || !consumed || !produced) return GR4PM_ERR_INVALID;
    *consumed = *produced = 0;
    if (n_tags_out) *n_tags_out = 0;
    if (cfc->stream != sf->stream) {
        set_error("fused call needs the CoarseFrequencyCorrection and the SymbolFilter on one stream");
        return GR4PM_ERR_INVALID;
    }
    if (n_in == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    if (out_cap < n_in / sf->sps + n_tags_in + 2) { // checked before the plan consumes the tags
        set_error("out_cap too small for a fused call");
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    if (cfc->n_channels != 1) { // validated before any state is touched
        set_error("fused call needs a single-channel CoarseFrequencyCorrection");
        return GR4PM_ERR_INVALID;
    }
    int plan = -1;
    gr4pm_status st;
    {
        // same stream: the filter queues up behind the checkpoints
        // (one stream, nothing in between: the current plan set is reused, no ring)
        DeferredSyncScope defer;
        st = cfc_plan_impl(cfc, n_in, tags_in, nullptr, n_tags_in, &plan, false);
    }
    if (st != GR4PM_OK) return st;
    return gr4pm_cfc_symbol_filter_run(cfc, plan, sf, in, n_in, out, out_cap, tags_in, n_tags_in, tags_out,
                                       tags_cap, n_tags_out, consumed, produced);
}
GR4PM_ABI_CATCH

} // extern "C"

// ------------------------------------------------------------------ PfbArbResampler
struct gr4pm_pfb_arb_resampler {
    size_t filter_size, arm_size, n_taps;
    int rate_is_double;
    unsigned long long decim_rate;
    double filt_rate_d;
    float filt_rate_f;
    unsigned cap, plan_cap = 0;
    hipStream_t stream;
    DevBuf<float> taps, diff_taps;
    DevBuf<cf> carry[2];
    DevBuf<ArbState> st;
    DevBuf<char> plan_ck; // ArbCk<TRate> per kArbChunk outputs
    PinnedBuf<ArbState> st_host;
    int cur = 0;
};

static gr4pm_status arb_reset_impl(gr4pm_pfb_arb_resampler* h)
{
    ArbState st{};
    st.last_filter = (h->n_taps / 2) % h->filter_size; // pfb_arb_resampler.hpp:119
    st.phase_acc_d = 0.0;                              // :118
    st.phase_acc_f = 0.0f;
    *h->st_host.p = st;
    GR4PM_HIP_TRY(hipMemcpyAsync(h->st.p, h->st_host.p, sizeof(ArbState), hipMemcpyHostToDevice, h->stream));
    for (int i = 0; i < 2; ++i) GR4PM_TRY(h->carry[i].zero(h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    return GR4PM_OK;
}

extern "C" {

gr4pm_status gr4pm_pfb_arb_resampler_create(const gr4pm_pfb_arb_resampler_params* p,
                                            gr4pm_pfb_arb_resampler** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    if (p->filter_size == 0) { // :70-72
        set_error("filter_size cannot be 0");
        return GR4PM_ERR_INVALID;
    }
    if (!p->taps || p->n_taps < 2) {
        set_error("taps required (the default prototype is supplied by the host wrapper)");
        return GR4PM_ERR_INVALID;
    }
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_pfb_arb_resampler;
    if (!h) return GR4PM_ERR_NOMEM;
    h->filter_size = p->filter_size;
    h->n_taps = p->n_taps;
    h->rate_is_double = p->rate_is_double;
    h->stream = static_cast<hipStream_t>(p->stream);
    h->arm_size = (p->n_taps + p->filter_size - 1) / p->filter_size; // :74
    std::vector<float> taps(h->filter_size * h->arm_size, 0.0f), diff(h->filter_size * h->arm_size, 0.0f);
    for (size_t j = 0; j < h->filter_size; ++j) { // :77-102
        size_t m = 0;
        for (size_t k = j; k < p->n_taps; k += h->filter_size) taps[j * h->arm_size + m++] = p->taps[k];
        m = 0;
        for (size_t k = j; k < p->n_taps - 1; k += h->filter_size)
            diff[j * h->arm_size + m++] = p->taps[k + 1] - p->taps[k];
    }
    h->cap = static_cast<unsigned>(bit_ceil_sz(h->arm_size)); // :105
    if (h->rate_is_double) { // :115-117
        const double fr = static_cast<double>(h->filter_size) / p->rate;
        h->decim_rate = static_cast<unsigned long long>(std::floor(fr));
        h->filt_rate_d = fr - static_cast<double>(h->decim_rate);
        h->filt_rate_f = 0.0f;
    } else {
        const float fr = static_cast<float>(h->filter_size) / static_cast<float>(p->rate);
        h->decim_rate = static_cast<unsigned long long>(std::floor(fr));
        h->filt_rate_f = fr - static_cast<float>(h->decim_rate);
        h->filt_rate_d = 0.0;
    }
    gr4pm_status s = h->taps.alloc(taps.size());
    if (s == GR4PM_OK) s = h->diff_taps.alloc(diff.size());
    if (s == GR4PM_OK) s = h->st.alloc(1);
    if (s == GR4PM_OK) s = h->st_host.alloc(1);
    for (int i = 0; i < 2 && s == GR4PM_OK; ++i) s = h->carry[i].alloc(h->cap);
    if (s == GR4PM_OK) s = h->taps.upload(taps.data(), taps.size(), h->stream);
    if (s == GR4PM_OK) s = h->diff_taps.upload(diff.data(), diff.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s == GR4PM_OK) s = arb_reset_impl(h);
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_pfb_arb_resampler_destroy(gr4pm_pfb_arb_resampler* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_pfb_arb_resampler_reset(gr4pm_pfb_arb_resampler* h)
try {
    return h ? arb_reset_impl(h) : GR4PM_ERR_INVALID;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_pfb_arb_resampler_process(gr4pm_pfb_arb_resampler* h, const gr4pm_c64* in, size_t n_in,
                                             gr4pm_c64* out, size_t out_cap, size_t* consumed, size_t* produced)
try {
    if (!h || !consumed || !produced) return GR4PM_ERR_INVALID;
    *consumed = *produced = 0;
    if (n_in == 0 || out_cap == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    if (out_cap > 0xffffffffull || n_in > 0x7fffffffull) return GR4PM_ERR_INVALID;
    hipStream_t s = h->stream;
    const size_t n_ck = out_cap / kArbChunk + 2;
    if (h->plan_cap < n_ck) {
        GR4PM_TRY(h->plan_ck.alloc(n_ck * 24)); // ArbCk<float> / ArbCk<double>: 24 bytes each
        h->plan_cap = static_cast<unsigned>(n_ck);
    }
    static_assert(sizeof(ArbCk<float>) == 24 && sizeof(ArbCk<double>) == 24, "checkpoint layout");
    const cf* carry = h->carry[h->cur].p;
    const unsigned grid = grid_for(out_cap, 256, 16384);
    const unsigned fs = static_cast<unsigned>(h->filter_size);
    if (h->decim_rate / fs >= (1ull << 30)) {
        // the plan kernels walk the input with 32-bit item counts (q0 items per output, q0 + 1 after a wrap): a rate this
        // small would wrap them where the reference's 64-bit walk (pfb_arb_resampler.hpp:135-138) does not
        set_error("PfbArbResampler: rate too small for the device path (decim_rate / filter_size = %llu >= 2^30)",
                  static_cast<unsigned long long>(h->decim_rate / fs));
        return GR4PM_ERR_INVALID;
    }
    const unsigned q0 = static_cast<unsigned>(h->decim_rate / fs), r0 = static_cast<unsigned>(h->decim_rate % fs);
    if (h->rate_is_double) {
        auto* ck = reinterpret_cast<ArbCk<double>*>(h->plan_ck.p);
        hipLaunchKernelGGL(k_arb_plan<double>, dim3(1), dim3(64), 0, s, h->st.p, static_cast<unsigned>(n_in),
                           static_cast<unsigned>(out_cap), fs, h->decim_rate, q0, r0, h->filt_rate_d, ck);
        hipLaunchKernelGGL(k_arb_filter<double>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const cf*>(in), carry,
                           h->cap, h->taps.p, h->diff_taps.p, static_cast<unsigned>(h->arm_size), h->st.p, ck, fs, q0, r0,
                           h->filt_rate_d, reinterpret_cast<cf*>(out));
    } else {
        auto* ck = reinterpret_cast<ArbCk<float>*>(h->plan_ck.p);
        hipLaunchKernelGGL(k_arb_plan<float>, dim3(1), dim3(64), 0, s, h->st.p, static_cast<unsigned>(n_in),
                           static_cast<unsigned>(out_cap), fs, h->decim_rate, q0, r0, h->filt_rate_f, ck);
        hipLaunchKernelGGL(k_arb_filter<float>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const cf*>(in), carry,
                           h->cap, h->taps.p, h->diff_taps.p, static_cast<unsigned>(h->arm_size), h->st.p, ck, fs, q0, r0,
                           h->filt_rate_f, reinterpret_cast<cf*>(out));
    }
    hipLaunchKernelGGL(k_arb_update_hist, dim3((h->cap + 63) / 64), dim3(64), 0, s,
                       reinterpret_cast<const cf*>(in), carry, h->carry[h->cur ^ 1].p, h->cap, h->st.p);
    GR4PM_HIP_TRY(hipMemcpyAsync(h->st_host.p, h->st.p, sizeof(ArbState), hipMemcpyDeviceToHost, s));
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipStreamSynchronize(s));
    h->cur ^= 1;
    *consumed = h->st_host.p->consumed;
    *produced = h->st_host.p->produced;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"

// =====================================================================================
// Symbol-rate control blocks behind SyncwordWipeoff (include/gr4pm_hip.h).
// The per-item work of PayloadMetadataInsert and SyncwordRemove is a gather of item spans;
// which spans is decided by a host replay of the blocks' state machines over the tags.
// =====================================================================================
namespace gr4pm {
namespace {

using hostlogic::CopySpan; // hostlogic/base.hpp
// grid (x, n_spans): the blocks of a row walk their span with coalesced 8-byte accesses
__global__ __launch_bounds__(256) void k_gather_spans(const CopySpan* __restrict__ spans, const cf* __restrict__ in,
                                                      cf* __restrict__ out)
{
    const CopySpan sp = spans[blockIdx.y];
    const cf* src = in + sp.src;
    cf* dst = out + sp.dst;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < sp.len;
         i += static_cast<unsigned long long>(gridDim.x) * blockDim.x)
        dst[i] = src[i];
}
gr4pm_status launch_gather(hipStream_t s, DevBuf<CopySpan>& buf, const std::vector<CopySpan>& spans, const cf* in,
                           cf* out)
{
    if (spans.empty()) return GR4PM_OK;
    GR4PM_TRY(upload_vec(buf, spans, s));
    unsigned long long longest = 0;
    for (const auto& sp : spans) longest = std::max(longest, sp.len);
    const unsigned gx = static_cast<unsigned>(std::min<unsigned long long>((longest + 2047) / 2048, 1024));
    for (size_t first = 0; first < spans.size(); first += 65535) { // gridDim.y limit
        const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, spans.size() - first));
        hipLaunchKernelGGL(k_gather_spans, dim3(std::max(gx, 1u), rows), dim3(256), 0, s, buf.p + first, in, out);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    return GR4PM_OK;
}

// LLR mapping of one run of symbols with one constellation: BPSK scale * re, QPSK
// (scale * re, scale * im) = a scaled copy of the interleaved floats
struct LlrRun {
    unsigned long long in0, out0, n_out;
    int qpsk;
    int pad;
};
__global__ __launch_bounds__(256) void k_llr(const LlrRun* __restrict__ runs, float scale,
                                             const float* __restrict__ in, float* __restrict__ out)
{
    const LlrRun r = runs[blockIdx.y];
    const float* src = in + 2 * r.in0;
    float* dst = out + r.out0;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < r.n_out;
         i += static_cast<unsigned long long>(gridDim.x) * blockDim.x)
        dst[i] = scale * src[r.qpsk ? i : 2 * i]; // constellation_llr_decoder.hpp:106-116
}

} // namespace
} // namespace gr4pm

struct gr4pm_payload_metadata_insert : gr4pm::hostlogic::PmiState {
    hipStream_t stream = nullptr;
    DevBuf<gr4pm::hostlogic::CopySpan> spans;
};
struct gr4pm_syncword_remove : gr4pm::hostlogic::SrState {
    hipStream_t stream = nullptr;
    DevBuf<gr4pm::hostlogic::CopySpan> spans;
};
struct gr4pm_constellation_llr_decoder {
    float noise_sigma, scale;
    int constellation;
    hipStream_t stream;
    DevBuf<LlrRun> runs;
};

// the host half of ConstellationLLRDecoder::processBulk over one call (constellation_llr_decoder.hpp:84-130): runs of
// symbols with one constellation, the tags re-indexed to LLR positions (:93-99); the block's constellation follows the tags
static gr4pm_status llr_runs(gr4pm_constellation_llr_decoder* h, size_t n, size_t out_cap, const gr4pm_packet_tag* tags_in,
                             size_t n_tags_in, gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* produced,
                             std::vector<gr4pm::LlrRun>& runs)
{
    using gr4pm::LlrRun;
    size_t pos = 0, opos = 0, n_pub = 0;
    bool tag_overflow = false;
    auto close_run = [&](size_t end) {
        if (end <= pos) return;
        LlrRun r{};
        r.in0 = pos;
        r.out0 = opos;
        r.qpsk = h->constellation == 2;
        r.n_out = (end - pos) * (r.qpsk ? 2 : 1);
        runs.push_back(r);
        opos += r.n_out;
        pos = end;
    };
    for (size_t t = 0; t < n_tags_in; ++t) {
        if (tags_in[t].index >= n) break;
        close_run(static_cast<size_t>(tags_in[t].index));
        if (tags_in[t].constellation >= 0) {
            if (tags_in[t].constellation != 1 && tags_in[t].constellation != 2) {
                gr4pm::set_error("constellation %d not supported", tags_in[t].constellation);
                return GR4PM_ERR_INVALID;
            }
            h->constellation = tags_in[t].constellation;
        }
        if (tags_out && n_pub < tags_cap) { // :93-99
            tags_out[n_pub] = tags_in[t];
            tags_out[n_pub].index = opos;
        } else {
            tag_overflow = true;
        }
        ++n_pub;
    }
    close_run(n);
    if (opos > out_cap) {
        gr4pm::set_error("out_cap %zu < %zu LLRs", out_cap, opos);
        return GR4PM_INSUFFICIENT_OUTPUT_ITEMS;
    }
    *produced = opos;
    if (n_tags_out) *n_tags_out = n_pub;
    if (tag_overflow) {
        gr4pm::set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}
// (library-internal) PayloadMetadataInsert::processBulk's host half: the state machine over the tags (hostlogic/packet_control.hpp)
gr4pm_status gr4pm::payload_metadata_insert_plan(gr4pm_payload_metadata_insert* h, size_t n_in, size_t out_cap,
                                                 const gr4pm_tag* tags_in, size_t n_tags_in, const gr4pm_header_msg* headers,
                                                 size_t n_headers, int headers_per_tag, gr4pm_packet_tag* tags_out, size_t tags_cap,
                                                 size_t* n_tags_out, size_t* consumed, size_t* produced, size_t* headers_used,
                                                 size_t* ignored_syncwords, std::vector<hostlogic::CopySpan>& spans)
{
    if (!h || !n_tags_out || !consumed || !produced || !headers_used || !ignored_syncwords) return GR4PM_ERR_INVALID;
    *n_tags_out = *consumed = *produced = *headers_used = *ignored_syncwords = 0;
    spans.clear();
    if (headers_per_tag && n_headers != n_tags_in) {
        set_error("headers_per_tag needs one message per tag (%zu != %zu)", n_headers, n_tags_in);
        return GR4PM_ERR_INVALID;
    }
    if (n_in == 0) return GR4PM_OK;
    hostlogic::PmiReplay rp;
    GR4PM_TRY(hostlogic::pmi_replay(*h, n_in, out_cap, tags_in, n_tags_in, headers, n_headers, headers_per_tag, tags_out,
                                    tags_cap, rp));
    spans.swap(rp.spans);
    *n_tags_out = rp.n_pub;
    *consumed = rp.consumed;
    *produced = rp.produced;
    *headers_used = rp.headers_used;
    *ignored_syncwords = rp.ignored;
    if (rp.tag_overflow) {
        set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}
// (library-internal: csrc/packet_receiver.hip, the packets_only receiver) the host halves alone: state, tags, spans
gr4pm_status gr4pm::syncword_remove_plan(gr4pm_syncword_remove* h, size_t n, const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                                         gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* produced,
                                         std::vector<hostlogic::CopySpan>& spans)
{
    if (!h || !produced) return GR4PM_ERR_INVALID;
    hostlogic::SrReplay rp; // the state machine: hostlogic/packet_control.hpp
    hostlogic::sr_replay(*h, n, tags_in, n_tags_in, tags_out, tags_cap, rp);
    spans.swap(rp.spans);
    *produced = rp.produced;
    if (n_tags_out) *n_tags_out = rp.n_pub;
    if (rp.tag_overflow) {
        set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}
gr4pm_status gr4pm::llr_decoder_plan(gr4pm_constellation_llr_decoder* h, size_t n, const gr4pm_packet_tag* tags_in,
                                     size_t n_tags_in, gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                                     size_t* produced, bool* all_qpsk, float* scale)
{
    if (!h || !produced || !all_qpsk || !scale) return GR4PM_ERR_INVALID;
    std::vector<LlrRun> runs;
    *produced = 0;
    if (n_tags_out) *n_tags_out = 0;
    const gr4pm_status st = llr_runs(h, n, static_cast<size_t>(-1), tags_in, n_tags_in, tags_out, tags_cap, n_tags_out, produced, runs);
    *all_qpsk = true;
    for (const auto& r : runs) *all_qpsk = *all_qpsk && r.qpsk;
    *scale = h->scale;
    return st;
}

extern "C" {

gr4pm_status gr4pm_payload_metadata_insert_create(const gr4pm_payload_metadata_insert_params* p,
                                                  gr4pm_payload_metadata_insert** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_payload_metadata_insert;
    if (!h) return GR4PM_ERR_NOMEM;
    h->syncword_size = p->syncword_size;
    h->header_size = p->header_size;
    h->syncword_bw = p->syncword_costas_loop_bandwidth;
    h->header_bw = p->header_costas_loop_bandwidth;
    h->payload_bw = p->payload_costas_loop_bandwidth;
    h->stream = static_cast<hipStream_t>(p->stream);
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_payload_metadata_insert_destroy(gr4pm_payload_metadata_insert* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_payload_metadata_insert_reset(gr4pm_payload_metadata_insert* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->in_packet = false; // start(), :71-75
    h->position = 0;
    h->has_held = false;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_payload_metadata_insert_process(
    gr4pm_payload_metadata_insert* h, const gr4pm_c64* in, size_t n_in, gr4pm_c64* out, size_t out_cap,
    const gr4pm_tag* tags_in, size_t n_tags_in, const gr4pm_header_msg* headers, size_t n_headers,
    int headers_per_tag, gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out, size_t* consumed,
    size_t* produced, size_t* headers_used, size_t* ignored_syncwords)
try {
    if (!h || !n_tags_out || !consumed || !produced || !headers_used || !ignored_syncwords) return GR4PM_ERR_INVALID;
    *n_tags_out = *consumed = *produced = *headers_used = *ignored_syncwords = 0;
    if (n_in && (!in || !out)) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<CopySpan> spans;
    const gr4pm_status st = gr4pm::payload_metadata_insert_plan(h, n_in, out_cap, tags_in, n_tags_in, headers, n_headers,
                                                                headers_per_tag, tags_out, tags_cap, n_tags_out, consumed,
                                                                produced, headers_used, ignored_syncwords, spans);
    if (st != GR4PM_OK && st != GR4PM_ERR_OVERFLOW) return st;
    GR4PM_TRY(launch_gather(h->stream, h->spans, spans, reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out)));
    GR4PM_HIP_TRY(final_sync(h->stream));
    return st;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_payload_metadata_insert_resolve(gr4pm_payload_metadata_insert* h, const gr4pm_header_msg* msg)
try {
    if (!h || !msg) return GR4PM_ERR_INVALID;
    if (h->in_packet && !h->has_held) {
        h->held = *msg;
        h->has_held = true;
    }
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_syncword_remove_create(const gr4pm_syncword_remove_params* p, gr4pm_syncword_remove** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_syncword_remove;
    if (!h) return GR4PM_ERR_NOMEM;
    h->syncword_size = p->syncword_size;
    h->stream = static_cast<hipStream_t>(p->stream);
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_syncword_remove_destroy(gr4pm_syncword_remove* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_syncword_remove_reset(gr4pm_syncword_remove* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->in_syncword = false;
    h->position = 0;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
gr4pm_status gr4pm_syncword_remove_process(gr4pm_syncword_remove* h, const gr4pm_c64* in, size_t n, gr4pm_c64* out,
                                           const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                                           gr4pm_packet_tag* tags_out, size_t tags_cap, size_t* n_tags_out,
                                           size_t* produced)
try {
    if (!h || !produced) return GR4PM_ERR_INVALID;
    *produced = 0;
    if (n_tags_out) *n_tags_out = 0;
    if (n == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<CopySpan> spans;
    const gr4pm_status st = gr4pm::syncword_remove_plan(h, n, tags_in, n_tags_in, tags_out, tags_cap, n_tags_out, produced, spans);
    if (st != GR4PM_OK && st != GR4PM_ERR_OVERFLOW) return st;
    GR4PM_TRY(launch_gather(h->stream, h->spans, spans, reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out)));
    GR4PM_HIP_TRY(final_sync(h->stream));
    return st;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_constellation_llr_decoder_create(const gr4pm_constellation_llr_decoder_params* p,
                                                    gr4pm_constellation_llr_decoder** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    if (p->constellation != 1 && p->constellation != 2) { // :72-74
        set_error("constellation %d not supported", p->constellation);
        return GR4PM_ERR_INVALID;
    }
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_constellation_llr_decoder;
    if (!h) return GR4PM_ERR_NOMEM;
    h->noise_sigma = p->noise_sigma;
    h->scale = 2.0f / (p->noise_sigma * p->noise_sigma); // :77
    h->constellation = p->constellation;
    h->stream = static_cast<hipStream_t>(p->stream);
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_constellation_llr_decoder_destroy(gr4pm_constellation_llr_decoder* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_constellation_llr_decoder_process(gr4pm_constellation_llr_decoder* h, const gr4pm_c64* in,
                                                     size_t n, float* out, size_t out_cap,
                                                     const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                                                     gr4pm_packet_tag* tags_out, size_t tags_cap,
                                                     size_t* n_tags_out, size_t* produced)
try {
    if (!h || !produced) return GR4PM_ERR_INVALID;
    *produced = 0;
    if (n_tags_out) *n_tags_out = 0;
    if (n == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<LlrRun> runs;
    const gr4pm_status st = llr_runs(h, n, out_cap, tags_in, n_tags_in, tags_out, tags_cap, n_tags_out, produced, runs);
    if (st != GR4PM_OK && st != GR4PM_ERR_OVERFLOW) return st;
    GR4PM_TRY(upload_vec(h->runs, runs, h->stream));
    unsigned long long longest = 0;
    for (const auto& r : runs) longest = std::max(longest, r.n_out);
    const unsigned gx = static_cast<unsigned>(std::min<unsigned long long>((longest + 2047) / 2048, 1024));
    for (size_t first = 0; first < runs.size(); first += 65535) {
        const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, runs.size() - first));
        hipLaunchKernelGGL(k_llr, dim3(std::max(gx, 1u), rows), dim3(256), 0, h->stream, h->runs.p + first, h->scale,
                           reinterpret_cast<const float*>(in), out);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(h->stream));
    return st;
}
GR4PM_ABI_CATCH

} // extern "C"
