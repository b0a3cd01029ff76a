// header_blocks.hip -- the header decode loop of the receiver (packet_receiver.hpp:131-139,
// SURVEY.md 8(f) rank 2) behind the C ABI of include/gr4pm_hip.h:
//   AdditiveScrambler<float|uint8_t>  additive_scrambler.hpp:58-100
//   HeaderPayloadSplit<float>         header_payload_split.hpp:38-135
//   HeaderFecDecoder                  header_fec_decoder.hpp:290-347 (+ the LDPC decoder the
//                                     reference takes from its Rust dependency ldpc-toolbox)
//   HeaderParser                      header_parser.hpp:46-95 (host)
// Compiled with -ffp-contract=off: the LDPC arithmetic is bit-identical to oracle/.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "hostlogic/packet_control.hpp"
#include "hostlogic/tail_plan.hpp"

namespace gr4pm {
namespace {

// ------------------------------------------------------------------ AdditiveScrambler
// The LFSR output depends only on the number of items since the last reset, so the sequence
// is tabulated once: `prefix` items, then a cycle of `period` items.
using ScrRun = hostlogic::ScrambleRun; // items [start, start+len) use sequence index phase, phase+1, ... (hostlogic/tail_plan.hpp)
template <typename T>
__global__ __launch_bounds__(256) void k_scramble(const ScrRun* __restrict__ runs, const uint8_t* __restrict__ seq,
                                                  unsigned long long prefix, unsigned long long period,
                                                  const T* __restrict__ in, T* __restrict__ out)
{
    const ScrRun r = runs[blockIdx.y];
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < r.len;
         i += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
        unsigned long long q = r.phase + i;
        if (q >= prefix + period) q = prefix + (q - prefix) % period;
        const uint8_t bit = seq[q];
        const T a = in[r.start + i];
        if constexpr (sizeof(T) == 1) out[r.start + i] = a ^ bit;  // additive_scrambler.hpp:90
        else out[r.start + i] = bit ? -a : a;                       // :92-93
    }
}

// ------------------------------------------------------------------ span gather (float / complex<float> items)
using FSpan = hostlogic::CopySpan; // hostlogic/base.hpp
template <typename T>
__global__ __launch_bounds__(256) void k_gather_items(const FSpan* __restrict__ spans, const T* __restrict__ in,
                                                      T* __restrict__ out)
{
    const FSpan sp = spans[blockIdx.y];
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < sp.len;
         i += static_cast<unsigned long long>(gridDim.x) * blockDim.x)
        out[sp.dst + i] = in[sp.src + i];
}
template <typename T>
gr4pm_status gather_items(hipStream_t s, DevBuf<FSpan>& buf, const std::vector<FSpan>& spans, const T* in, T* out)
{
    if (spans.empty()) return GR4PM_OK;
    if (buf.n < spans.size()) GR4PM_TRY(buf.alloc(spans.size() * 2));
    GR4PM_TRY(buf.upload_staged(spans.data(), spans.size(), s));
    unsigned long long longest = 0;
    for (const auto& sp : spans) longest = std::max(longest, sp.len);
    const unsigned gx = static_cast<unsigned>(std::max<unsigned long long>(
        1, std::min<unsigned long long>((longest + 2047) / 2048, 1024)));
    for (size_t first = 0; first < spans.size(); first += 65535) {
        const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, spans.size() - first));
        hipLaunchKernelGGL(k_gather_items<T>, dim3(gx, rows), dim3(256), 0, s, buf.p + first, in, out);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    return GR4PM_OK;
}

// ------------------------------------------------------------------ header FEC decoder
// One wavefront per codeword.  Posteriors P[n] and check-to-variable messages R[m][kMaxDeg]
// live in LDS.  Horizontal-layered schedule.  The layers are built greedily: scan the checks
// not yet placed in index order and take every one that shares no variable with the checks
// already taken in this pass (at most 64 edges); a pass = one layer = one step of the kernel,
// one lane per (check, edge).  Checks of a layer touch disjoint variables, so doing them at
// once is exactly the serial layered decoder that visits the checks layer by layer (the order
// oracle/ uses); for the header code that is 8 steps per iteration instead of 96.
constexpr int kMaxDeg = 8;
constexpr int kMaxN = 256, kMaxM = 192, kMaxSteps = 24;
// one schedule entry = everything a lane needs for its (check, edge): the check's variables
// (bytes 0-4 ... kMaxDeg-1 would not fit: degrees above 5 use bytes 0-4 + the table), its degree,
// the lane's edge and the check index, so that a step starts with ONE 8-byte LDS read
constexpr int kPackDeg = 5;
struct LdpcDev {
    const uint8_t* row_var; // [m][kMaxDeg], 0xFF = unused
    const uint8_t* row_deg; // [m]
    const unsigned long long* sched; // [n_steps][64]: see pack_entry(); all ones = idle lane
    const float* corr;      // [64]: ln(1 + e^-x), x = i / 8
    unsigned n, m, n_steps;
};
// Q8 (gr4pm_header_fec_decoder_params::arithmetic == 1): the same schedule and check-node rule with 8-bit messages, as the
// name of the reference's decoder says it runs ("HLAminstari8", header_fec_decoder.hpp:276: horizontal-layered, A-Min*,
// i8): every LLR is a whole number of eighths, the channel LLRs and every message saturate at +-127 (= +-15.875), the
// correction table holds rounded eighths, posteriors keep 16 bits (a posterior cut to the message width with every
// update loses 3 dB at the receiver's LLR scale, 2 / 0.7^2: measured).  Values are integers carried in floats (exact).
// The crate's own rounding and saturation points are not visible here: this form measures what 8-bit messages change
// (tests: frame-error rates of the two forms side by side), it is not a restatement of the crate.
template <bool Q8>
__device__ __forceinline__ float ldpc_corr(const float* corr, float x) // x >= 0
{
    if (Q8) return x >= 64.0f ? 0.0f : corr[static_cast<int>(x)];
    return x >= 8.0f ? 0.0f : corr[static_cast<int>(x * 8.0f)];
}
template <bool Q8>
__device__ __forceinline__ float ldpc_boxplus(const float* corr, float a, float b) // magnitudes
{
    const float mn = a < b ? a : b;
    const float r = mn + ldpc_corr<Q8>(corr, a + b) - ldpc_corr<Q8>(corr, a < b ? b - a : a - b);
    return r > 0.0f ? r : 0.0f;
}
__device__ __forceinline__ float ldpc_sat(float x, float limit) { return fminf(fmaxf(x, -limit), limit); }
template <bool Q8>
__global__ __launch_bounds__(64) void k_header_fec(const float* __restrict__ llrs, unsigned n_codewords, LdpcDev d,
                                                   unsigned max_iterations, unsigned n_llrs_per_codeword,
                                                   uint8_t* __restrict__ headers, uint8_t* __restrict__ invalid)
{
    __shared__ float P[kMaxN];
    __shared__ __attribute__((aligned(16))) float R[kMaxM * kMaxDeg];
    __shared__ float corr[64];
    // the code's tables, once per wavefront, so that no step waits for global memory
    __shared__ __attribute__((aligned(8))) uint8_t s_var[kMaxM * kMaxDeg];
    __shared__ uint8_t s_deg[kMaxM];
    __shared__ unsigned long long s_sched[kMaxSteps * 64];
    const unsigned cw = blockIdx.x;
    if (cw >= n_codewords) return;
    const int lane = threadIdx.x;
    const float* x = llrs + static_cast<size_t>(cw) * n_llrs_per_codeword;
    corr[lane] = d.corr[lane];
    for (unsigned e = lane; e < d.m * kMaxDeg; e += 64) s_var[e] = d.row_var[e];
    for (unsigned e = lane; e < d.m; e += 64) s_deg[e] = d.row_deg[e];
    for (unsigned e = lane; e < d.n_steps * 64; e += 64) s_sched[e] = d.sched[e];
    // header_fec_decoder.hpp:308-312: accumulate the two copies of the repetition code
    for (unsigned v = lane; v < d.n; v += 64) {
        const float acc = x[v] + x[d.n + v];
        P[v] = Q8 ? ldpc_sat(rintf(8.0f * acc), 127.0f) : acc;
    }
    for (unsigned e = lane; e < d.m * kMaxDeg; e += 64) R[e] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    bool found = false;
    for (unsigned it = 0; it <= max_iterations; ++it) {
        // is the hard decision a codeword?
        bool bad = false;
        for (unsigned c = lane; c < d.m; c += 64) {
            unsigned parity = 0;
            const unsigned dc = s_deg[c];
#pragma unroll
            for (unsigned j = 0; j < kMaxDeg; ++j)
                if (j < dc) parity ^= P[s_var[c * kMaxDeg + j]] < 0.0f ? 1u : 0u;
            bad |= parity != 0;
        }
        if (!__any(bad)) {
            found = true;
            break;
        }
        if (it == max_iterations) break;
        for (unsigned s = 0; s < d.n_steps; ++s) {
            const unsigned long long entry = s_sched[s * 64 + lane];
            float q_own = 0.0f, r_own = 0.0f;
            unsigned v_own = 0, slot = 0;
            const bool active = entry != ~0ull;
            if (active) {
                const unsigned c = static_cast<unsigned>(entry >> 48) & 0xFFu;
                const unsigned dc = static_cast<unsigned>(entry >> 40) & 0xFu;
                const unsigned k = static_cast<unsigned>(entry >> 44) & 0xFu;
                // everything below is indexed with compile-time constants (registers, no scratch)
                float Q[kMaxDeg];
                unsigned neg = 0;
                const float4 r0 = *reinterpret_cast<const float4*>(R + c * kMaxDeg);
                const float4 r1 = *reinterpret_cast<const float4*>(R + c * kMaxDeg + 4);
                const float rr[kMaxDeg] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w };
#pragma unroll
                for (unsigned j = 0; j < kMaxDeg; ++j) {
                    Q[j] = 0.0f;
                    if (j < dc) {
                        const unsigned v = j < kPackDeg ? static_cast<unsigned>(entry >> (8 * j)) & 0xFFu
                                                        : s_var[c * kMaxDeg + j];
                        Q[j] = P[v] - rr[j];
                        if (Q8) {
                            if (j == k) q_own = Q[j]; // the posterior keeps its 16 bits ...
                            Q[j] = ldpc_sat(Q[j], 127.0f); // ... the check node sees 8-bit messages
                        }
                        if (Q[j] < 0.0f) neg ^= 1u;
                    }
                }
                unsigned imin = 0;
                float minabs = fabsf(Q[0]);
#pragma unroll
                for (unsigned j = 1; j < kMaxDeg; ++j)
                    if (j < dc && fabsf(Q[j]) < minabs) { // strict <: the first minimum, like the oracle
                        minabs = fabsf(Q[j]);
                        imin = j;
                    }
                float others = -1.0f;
#pragma unroll
                for (unsigned j = 0; j < kMaxDeg; ++j) {
                    if (j < dc && j != imin) {
                        const float a = fabsf(Q[j]);
                        others = others < 0.0f ? a : ldpc_boxplus<Q8>(corr, others, a);
                    }
                }
                if (others < 0.0f) others = 0.0f;
                const float all = ldpc_boxplus<Q8>(corr, others, minabs);
                float qk = 0.0f;
#pragma unroll
                for (unsigned j = 0; j < kMaxDeg; ++j)
                    if (j == k) qk = Q[j];
                const float mag = k == imin ? others : all;
                const unsigned sgn = neg ^ (qk < 0.0f ? 1u : 0u);
                r_own = sgn ? -mag : mag;
                if (!Q8) q_own = qk;
                v_own = k < kPackDeg ? static_cast<unsigned>(entry >> (8 * k)) & 0xFFu : s_var[c * kMaxDeg + k];
                slot = c * kMaxDeg + k;
            }
            // all reads of this step are issued above, all writes below: LDS operations of one
            // wavefront execute in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (active) {
                R[slot] = r_own;
                P[v_own] = Q8 ? ldpc_sat(q_own + r_own, 32767.0f) : q_own + r_own;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    // the first k = n - m bits are the header (systematic), packed MSB first (:329-335)
    const unsigned k_bits = d.n - d.m;
    const unsigned long long mask = __ballot(static_cast<unsigned>(lane) < k_bits && P[lane] < 0.0f);
    if (lane < static_cast<int>(k_bits / 8)) {
        const unsigned b = static_cast<unsigned>((mask >> (8 * lane)) & 0xFFull);
        headers[static_cast<size_t>(cw) * (k_bits / 8) + lane] = static_cast<uint8_t>(__brev(b) >> 24);
    }
    if (lane == 0) invalid[cw] = found ? 0 : 1;
}

// ------------------------------------------------------------------ payload tail
__global__ __launch_bounds__(256) void k_slice(const float* __restrict__ in, size_t n, uint8_t* __restrict__ out,
                                               int invert)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        out[i] = invert ? in[i] < 0.0f : in[i] > 0.0f; // binary_slicer.hpp:28-33
}
__global__ __launch_bounds__(256) void k_pack(const uint8_t* __restrict__ in, size_t n_out, uint8_t* __restrict__ out,
                                              unsigned per, unsigned bits, int msb_first)
{
    const unsigned mask = (1u << bits) - 1u;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_out;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        unsigned join = 0, shift = 0;
        for (unsigned k = 0; k < per; ++k) { // pack_bits.hpp: MSB: first input in the top bits
            const unsigned chunk = in[i * per + k] & mask;
            if (msb_first) join = (join << bits) | chunk;
            else {
                join |= chunk << shift;
                shift += bits;
            }
        }
        out[i] = static_cast<uint8_t>(join);
    }
}
// BinarySlicer<true> + PackBits<MSB>(8 x 1 bit): one thread per output byte, two 16-byte loads
__global__ __launch_bounds__(256) void k_slice_pack(const float* __restrict__ in, size_t n_out,
                                                    uint8_t* __restrict__ out)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_out;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        unsigned b = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) b = (b << 1) | (in[i * 8 + k] < 0.0f ? 1u : 0u);
        out[i] = static_cast<uint8_t>(b);
    }
}
// the same over a stream that lies in two pieces (the native receiver's payload tail: the LLRs of the packet that the
// batch before left unfinished, then this batch's): item j comes from a[j] for j < na, from b[j - na] behind that
__global__ __launch_bounds__(256) void k_slice_pack_two(const float* __restrict__ a, size_t na, const float* __restrict__ b,
                                                        size_t n_out, uint8_t* __restrict__ out)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_out;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t j = i * 8 + k;
            v = (v << 1) | ((j < na ? a[j] : b[j - na]) < 0.0f ? 1u : 0u);
        }
        out[i] = static_cast<uint8_t>(v);
    }
}
// ------------------------------------------------------------------ the tail behind the Costas loop in one pass (round 6)
// SyncwordRemove -> ConstellationLLRDecoder (QPSK) -> AdditiveScrambler -> HeaderPayloadSplit -> { header LLRs |
// BinarySlicer<true> -> PackBits<MSB> } over the table hostlogic/tail_plan.hpp composes from the blocks' own state
// machines: every symbol of the Costas loop's output is read ONCE (8 bytes); a header symbol leaves as two descrambled
// LLR floats -- scale * re, scale * im (constellation_llr_decoder.hpp:106-116), negated where the LFSR bit is 1
// (additive_scrambler.hpp:92-93): the unfused chain's bits --, a payload symbol as two bits of a packed byte
// (binary_slicer.hpp:28-33 with invert: llr < 0; pack_bits.hpp MSB first).  grid (x, span).
using hostlogic::TailSpan;
__global__ __launch_bounds__(256) void k_tail_fused(const TailSpan* __restrict__ spans, const float* __restrict__ sym,
                                                    float scale, const uint8_t* __restrict__ seq,
                                                    unsigned long long prefix, unsigned long long period,
                                                    float* __restrict__ header_llr, uint8_t* packed)
{
    const TailSpan sp = spans[blockIdx.y];
    const unsigned step = gridDim.x * blockDim.x;
    if (sp.kind == 0) { // header: one lane per symbol
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < sp.n_sym; i += step) {
            const float2 x = *reinterpret_cast<const float2*>(sym + 2 * (sp.src + i));
            float2 o;
            hostlogic::tail_llr_pair(x.x, x.y, scale, seq, sp.phase + 2ull * i, prefix, period, o.x, o.y);
            *reinterpret_cast<float2*>(header_llr + sp.dst + 2ull * i) = o; // (dst is even: 8-byte aligned)
        }
        return;
    }
    // payload: one lane per output byte = up to four symbols (hostlogic/tail_plan.hpp: tail_payload_byte)
    const unsigned long long byte0 = sp.dst >> 3;
    const unsigned n_bytes = static_cast<unsigned>(((sp.dst + 2ull * sp.n_sym + 7) >> 3) - byte0);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += step) {
        const unsigned long long B = byte0 + i;
        unsigned v, mask;
        hostlogic::tail_payload_byte(sp, B, sym, scale, seq, prefix, period, v, mask);
        packed[B] = mask == 0xFFu ? static_cast<uint8_t>(v) : static_cast<uint8_t>((packed[B] & ~mask) | v);
    }
}

// CrcCheck: one lane per packet (table-driven, byte by byte, crc.hpp:130-147); the table sits in
// LDS.  ok[i] = the CRC at the end of the packet matches.
struct CrcPacket {
    unsigned long long offset, len;
};
__global__ __launch_bounds__(64) void k_crc_check(const uint8_t* __restrict__ in, const CrcPacket* __restrict__ pk,
                                                  unsigned n_packets, const unsigned long long* __restrict__ table,
                                                  unsigned num_bits, unsigned long long mask,
                                                  unsigned long long initial_value, unsigned long long final_xor,
                                                  int input_reflected, int result_reflected, int swap_endianness,
                                                  unsigned long long skip, uint8_t* __restrict__ ok)
{
    __shared__ unsigned long long t[256];
    for (int i = threadIdx.x; i < 256; i += 64) t[i] = table[i];
    __syncthreads();
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_packets) return;
    const CrcPacket q = pk[p];
    const unsigned crc_bytes = num_bits / 8;
    if (q.len <= crc_bytes) { // too short, crc_check.hpp:152-161
        ok[p] = 0;
        return;
    }
    const unsigned long long payload = q.len - crc_bytes;
    const uint8_t* d = in + q.offset;
    unsigned long long rem = initial_value;
    const unsigned long long first = skip < payload ? skip : payload;
    if (input_reflected) {
        for (unsigned long long k = first; k < payload; ++k) rem = t[(rem ^ d[k]) & 0xff] ^ (rem >> 8);
    } else {
        for (unsigned long long k = first; k < payload; ++k)
            rem = (t[((rem >> (num_bits - 8)) ^ d[k]) & 0xff] ^ (rem << 8)) & mask;
    }
    if (input_reflected != result_reflected) { // reflect(), crc.hpp:44-53
        unsigned long long w = rem, r = w & 1;
        for (unsigned i = 1; i < num_bits; ++i) {
            w >>= 1;
            r = (r << 1) | (w & 1);
        }
        rem = r;
    }
    rem ^= final_xor;
    unsigned long long in_packet = 0; // :167-178
    if (swap_endianness) {
        for (unsigned long long i = q.len; i-- > payload;) in_packet = (in_packet << 8) | d[i];
    } else {
        for (unsigned long long i = payload; i < q.len; ++i) in_packet = (in_packet << 8) | d[i];
    }
    ok[p] = in_packet == rem ? 1 : 0;
}
// Round 6: the same check EIGHT bytes a step (slicing by eight).  The byte-wise loop above is one dependent LDS look-up
// per byte -- 1 500 round trips a packet, ~ 1 ms for a batch of 1500-byte packets whatever their number.  The CRC is
// linear over GF(2): the register after eight bytes is the XOR of eight look-ups that do not depend on each other,
// slice[k][b] = the register after byte b and k zero bytes from a zero register, the old register folded into the first
// bytes.  Registers of at most 32 bits (the tables are 8 KiB of LDS: the workgroup still starts beside a correlator
// workgroup), reflected input of any such width or plain input of exactly 32 bits; everything else keeps the byte-wise
// kernel.  The head up to the first 8-byte boundary and the tail go byte by byte through slice[0] = the reference's table.
template <bool REFLECTED>
__global__ __launch_bounds__(64) void k_crc_check_sliced(const uint8_t* __restrict__ in, const CrcPacket* __restrict__ pk,
                                                         unsigned n_packets, const uint32_t* __restrict__ slice,
                                                         unsigned num_bits, unsigned long long mask,
                                                         unsigned long long initial_value, unsigned long long final_xor,
                                                         int result_reflected, int swap_endianness,
                                                         unsigned long long skip, uint8_t* __restrict__ ok)
{
    __shared__ uint32_t T[8][256];
    for (int i = threadIdx.x; i < 8 * 256; i += 64) (&T[0][0])[i] = slice[i];
    __syncthreads();
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_packets) return;
    const CrcPacket q = pk[p];
    const unsigned crc_bytes = num_bits / 8;
    if (q.len <= crc_bytes) { // too short, crc_check.hpp:152-161
        ok[p] = 0;
        return;
    }
    const unsigned long long payload = q.len - crc_bytes;
    const uint8_t* d = in + q.offset;
    uint32_t rem = static_cast<uint32_t>(initial_value);
    unsigned long long k = skip < payload ? skip : payload;
    auto byte_step = [&](uint8_t b) {
        if (REFLECTED) rem = T[0][(rem ^ b) & 0xff] ^ (rem >> 8);
        else rem = T[0][((rem >> 24) ^ b) & 0xff] ^ (rem << 8);
    };
    auto word_step = [&](unsigned long long w) {
        uint32_t lo = static_cast<uint32_t>(w), hi = static_cast<uint32_t>(w >> 32);
        if (REFLECTED) {
            lo ^= rem;
            rem = T[7][lo & 0xff] ^ T[6][(lo >> 8) & 0xff] ^ T[5][(lo >> 16) & 0xff] ^ T[4][lo >> 24] ^ T[3][hi & 0xff] ^
                  T[2][(hi >> 8) & 0xff] ^ T[1][(hi >> 16) & 0xff] ^ T[0][hi >> 24];
        } else { // the first byte in memory is the top byte of the register's word
            lo = __builtin_bswap32(lo) ^ rem;
            hi = __builtin_bswap32(hi);
            rem = T[7][lo >> 24] ^ T[6][(lo >> 16) & 0xff] ^ T[5][(lo >> 8) & 0xff] ^ T[4][lo & 0xff] ^ T[3][hi >> 24] ^
                  T[2][(hi >> 16) & 0xff] ^ T[1][(hi >> 8) & 0xff] ^ T[0][hi & 0xff];
        }
    };
    while (k < payload && (reinterpret_cast<uintptr_t>(d + k) & 7u)) byte_step(d[k++]);
    for (; k + 16 <= payload; k += 16) { // (both loads in flight before the first look-up)
        const unsigned long long w0 = *reinterpret_cast<const unsigned long long*>(d + k);
        const unsigned long long w1 = *reinterpret_cast<const unsigned long long*>(d + k + 8);
        word_step(w0);
        word_step(w1);
    }
    if (k + 8 <= payload) {
        word_step(*reinterpret_cast<const unsigned long long*>(d + k));
        k += 8;
    }
    while (k < payload) byte_step(d[k++]);
    unsigned long long r64 = rem & mask;
    if ((REFLECTED ? 1 : 0) != result_reflected) { // reflect(), crc.hpp:44-53
        unsigned long long w = r64, r = w & 1;
        for (unsigned i = 1; i < num_bits; ++i) {
            w >>= 1;
            r = (r << 1) | (w & 1);
        }
        r64 = r;
    }
    r64 ^= final_xor;
    unsigned long long in_packet = 0; // :167-178
    if (swap_endianness) {
        for (unsigned long long i = q.len; i-- > payload;) in_packet = (in_packet << 8) | d[i];
    } else {
        for (unsigned long long i = payload; i < q.len; ++i) in_packet = (in_packet << 8) | d[i];
    }
    ok[p] = in_packet == r64 ? 1 : 0;
}
// ------------------------------------------------------------------ burst generator pieces
__device__ __forceinline__ float2 operator*(float2 a, float t) { return make_float2(a.x * t, a.y * t); }
template <typename T>
__global__ __launch_bounds__(256) void k_mapper(const uint8_t* __restrict__ in, size_t n, const T* __restrict__ map,
                                                unsigned mask, T* __restrict__ out)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        out[i] = map[in[i] & mask]; // mapper.hpp:47-50
}
struct ShapePacket {
    unsigned long long offset, len;
};
// grid (x, n_packets): the edges of one packet (burst_shaper.hpp:98-124); the body was copied before
template <typename T>
__global__ __launch_bounds__(64) void k_burst_edges(const T* __restrict__ in, T* __restrict__ out,
                                                    const ShapePacket* __restrict__ pk,
                                                    const float* __restrict__ leading, unsigned n_lead,
                                                    const float* __restrict__ trailing, unsigned n_trail)
{
    const ShapePacket p = pk[blockIdx.x];
    const unsigned long long lead = p.len < n_lead ? p.len : n_lead; // :98-105
    for (unsigned long long j = threadIdx.x; j < lead; j += blockDim.x) out[p.offset + j] = in[p.offset + j] * leading[j];
    // what is left after the leading edge (and an unshaped middle) takes the END of the trailing shape
    const unsigned long long rest = p.len - lead;
    const unsigned long long tr = rest < n_trail ? rest : n_trail; // :115-124
    for (unsigned long long j = threadIdx.x; j < tr; j += blockDim.x) {
        const unsigned long long i = p.len - tr + j;
        out[p.offset + i] = in[p.offset + i] * trailing[n_trail - tr + j];
    }
}

struct BSpan {
    unsigned long long src, dst, len;
};
// CrcCheck, "passing packets leave back to back" (crc_check.hpp:180-202) without a trip to the host: one workgroup turns
// the verdicts into the copy spans of k_gather_u8 -- an exclusive sum of the passing packets' output lengths (a failing
// packet gets a span of length 0).  1024 lanes walk the packets 1024 at a time with a running carry.
__global__ __launch_bounds__(1024) void k_crc_spans(const CrcPacket* __restrict__ pk, const uint8_t* __restrict__ ok,
                                                    unsigned n_packets, unsigned drop, BSpan* __restrict__ spans)
{
    __shared__ unsigned long long part[16];
    __shared__ unsigned long long carry_s;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (unsigned base = 0; base < n_packets; base += 1024) {
        const unsigned i = base + threadIdx.x;
        unsigned long long n = 0;
        CrcPacket q{ 0, 0 };
        if (i < n_packets) {
            q = pk[i];
            n = ok[i] ? q.len - drop : 0;
        }
        // inclusive sum inside the wave, then over the sixteen waves
        unsigned long long v = n;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long u = __shfl_up(v, d);
            if (static_cast<int>(lane) >= d) v += u;
        }
        if (lane == 63) part[wave] = v;
        __syncthreads();
        unsigned long long before = carry_s;
        for (unsigned w = 0; w < wave; ++w) before += part[w];
        if (i < n_packets) spans[i] = BSpan{ q.offset, before + v - n, n };
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + v;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_gather_u8(const BSpan* __restrict__ spans, const uint8_t* __restrict__ in,
                                                   uint8_t* __restrict__ out)
{
    const BSpan sp = spans[blockIdx.y];
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < sp.len;
         i += static_cast<unsigned long long>(gridDim.x) * blockDim.x)
        out[sp.dst + i] = in[sp.src + i];
}

} // namespace
} // namespace gr4pm

using namespace gr4pm;

struct gr4pm_crc_check {
    gr4pm_crc_check_params p;
    unsigned long long table[256];
    unsigned long long mask;
    DevBuf<unsigned long long> d_table;
    DevBuf<uint32_t> d_slice; // slicing-by-eight tables (k_crc_check_sliced); empty: the byte-wise kernel
    bool sliced = false;
    DevBuf<CrcPacket> pk;
    DevBuf<uint8_t> ok;
    PinnedBuf<uint8_t> ok_host;
    DevBuf<BSpan> spans;
    unsigned long long reflect(unsigned long long word) const
    {
        unsigned long long ret = word & 1;
        for (unsigned i = 1; i < p.num_bits; ++i) {
            word >>= 1;
            ret = (ret << 1) | (word & 1);
        }
        return ret;
    }
};

struct gr4pm_additive_scrambler : gr4pm::hostlogic::ScrState { // (count, position, shape of the tabulated LFSR output)
    uint64_t mask, seed, length;
    int item_kind;
    hipStream_t stream;
    DevBuf<uint8_t> seq;
    DevBuf<ScrRun> runs;
};
struct gr4pm_header_payload_split : gr4pm::hostlogic::HpsState {
    hipStream_t stream = nullptr;
    DevBuf<gr4pm::hostlogic::CopySpan> hspans, pspans;
};
struct gr4pm_header_fec_decoder {
    unsigned n = 0, m = 0, n_steps = 0, max_iterations = 25;
    int arithmetic = 0; // 0: float32 messages, 1: 8-bit messages (k_header_fec<true>)
    hipStream_t stream;
    DevBuf<uint8_t> row_var, row_deg;
    DevBuf<unsigned long long> sched;
    DevBuf<float> corr;
    DevBuf<uint8_t> d_out;      // [n][hb] header bytes ++ [n] verdicts
    PinnedBuf<uint8_t> h_out;
};

extern "C" {

gr4pm_status gr4pm_additive_scrambler_create(const gr4pm_additive_scrambler_params* p,
                                             gr4pm_additive_scrambler** out)
try {
    if (!p || !out || (p->item_kind != 1 && p->item_kind != 2) || p->length > 63) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_additive_scrambler;
    if (!h) return GR4PM_ERR_NOMEM;
    h->mask = p->mask;
    h->seed = p->seed;
    h->length = p->length;
    h->count = p->count;
    h->item_kind = p->item_kind;
    h->stream = static_cast<hipStream_t>(p->stream);
    // run the LFSR (additive_scrambler.hpp:84-87) from the seed until a register value repeats
    // (or `count` items, after which it is reset anyway)
    std::vector<uint8_t> seq;
    const uint64_t cap = p->count ? p->count : (1ull << 24);
    uint64_t reg = p->seed;
    // Brent's cycle detection keeps this O(prefix + period) without a table of states
    uint64_t power = 1, lam = 1, tortoise = reg;
    auto step = [&](uint64_t r) {
        const uint64_t shift_in = static_cast<uint64_t>(__builtin_parityll(r & p->mask));
        return (shift_in << p->length) | (r >> 1);
    };
    uint64_t hare = step(reg);
    uint64_t walked = 1;
    while (tortoise != hare && walked < cap) {
        if (power == lam) {
            tortoise = hare;
            power *= 2;
            lam = 0;
        }
        hare = step(hare);
        ++lam;
        ++walked;
    }
    uint64_t mu = 0;
    if (tortoise == hare) {
        uint64_t a = reg, b = reg;
        for (uint64_t i = 0; i < lam; ++i) b = step(b);
        while (a != b) {
            a = step(a);
            b = step(b);
            ++mu;
        }
    } else { // no repeat within the cap: tabulate the cap, no cycle needed below it
        mu = cap;
        lam = 1;
        h->cyclic = false;
    }
    const uint64_t total = std::min<uint64_t>(mu + lam, cap + 1);
    seq.resize(total);
    for (uint64_t i = 0; i < total; ++i) {
        seq[i] = static_cast<uint8_t>(reg & 1); // :84
        reg = step(reg);
    }
    h->prefix = std::min<uint64_t>(mu, total - 1);
    h->period = total - h->prefix;
    h->table_len = total;
    gr4pm_status s = h->seq.alloc(seq.size());
    if (s == GR4PM_OK) s = h->seq.upload(seq.data(), seq.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_additive_scrambler_destroy(gr4pm_additive_scrambler* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_additive_scrambler_reset(gr4pm_additive_scrambler* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->position = 0;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
gr4pm_status gr4pm_additive_scrambler_process(gr4pm_additive_scrambler* h, const void* in, size_t n, void* out,
                                              const uint64_t* reset_index, size_t n_resets)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (n == 0) return GR4PM_OK;
    if (!in || !out) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    std::vector<ScrRun> runs;
    GR4PM_TRY(hostlogic::scramble_runs(*h, n, reset_index, n_resets, runs));
    if (h->runs.n < runs.size()) GR4PM_TRY(h->runs.alloc(runs.size() * 2));
    GR4PM_TRY(h->runs.upload_staged(runs.data(), runs.size(), h->stream));
    unsigned long long longest = 0;
    for (const auto& r : runs) longest = std::max(longest, r.len);
    const unsigned gx = static_cast<unsigned>(std::max<unsigned long long>(
        1, std::min<unsigned long long>((longest + 2047) / 2048, 2048)));
    for (size_t first = 0; first < runs.size(); first += 65535) {
        const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, runs.size() - first));
        if (h->item_kind == 1)
            hipLaunchKernelGGL(k_scramble<float>, dim3(gx, rows), dim3(256), 0, h->stream, h->runs.p + first, h->seq.p,
                               h->prefix, h->period, static_cast<const float*>(in), static_cast<float*>(out));
        else
            hipLaunchKernelGGL(k_scramble<uint8_t>, dim3(gx, rows), dim3(256), 0, h->stream, h->runs.p + first,
                               h->seq.p, h->prefix, h->period, static_cast<const uint8_t*>(in),
                               static_cast<uint8_t*>(out));
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(h->stream));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_header_payload_split_create(const gr4pm_header_payload_split_params* p,
                                               gr4pm_header_payload_split** out)
try {
    if (!p || !out || p->header_size == 0) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_header_payload_split;
    if (!h) return GR4PM_ERR_NOMEM;
    h->header_size = p->header_size;
    h->stream = static_cast<hipStream_t>(p->stream);
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_header_payload_split_destroy(gr4pm_header_payload_split* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_header_payload_split_reset(gr4pm_header_payload_split* h)
try {
    if (!h) return GR4PM_ERR_INVALID;
    h->in_payload = false; // start(), :41-45
    h->position = 0;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
} // extern "C"
// HeaderPayloadSplit<T>::processBulk for T = float (the header loop, packet_receiver.hpp:136-137) and T = complex<float>
// (the symbol tap of zmq_output, :159-162): one state machine over the tags, the items gathered span by span
template <typename T>
static gr4pm_status header_payload_split_impl(gr4pm_header_payload_split* h, const T* in, size_t n, T* header,
                                              size_t* n_header, T* payload, size_t* n_payload,
                                              const gr4pm_packet_tag* tags_in, size_t n_tags_in,
                                              gr4pm_packet_tag* header_tags, size_t* n_header_tags,
                                              gr4pm_packet_tag* payload_tags, size_t* n_payload_tags, size_t tags_cap)
{
    if (!h || !n_header || !n_payload) return GR4PM_ERR_INVALID;
    *n_header = *n_payload = 0;
    if (n_header_tags) *n_header_tags = 0;
    if (n_payload_tags) *n_payload_tags = 0;
    if (n == 0) return GR4PM_OK;
    if (!in || !header || !payload) {
        set_error("null sample pointer");
        return GR4PM_ERR_INVALID;
    }
    gr4pm::hostlogic::HpsReplay rp; // the state machine: hostlogic/packet_control.hpp
    GR4PM_TRY(gr4pm::hostlogic::hps_replay(*h, n, tags_in, n_tags_in, header_tags, payload_tags, tags_cap, rp));
    const std::vector<FSpan>& hs = rp.header_spans;
    const std::vector<FSpan>& ps = rp.payload_spans;
    const size_t hp = rp.n_header, pp = rp.n_payload, nht = rp.n_header_tags, npt = rp.n_payload_tags;
    const bool overflow = rp.tag_overflow;
    GR4PM_TRY(gather_items<T>(h->stream, h->hspans, hs, in, header));
    GR4PM_TRY(gather_items<T>(h->stream, h->pspans, ps, in, payload));
    GR4PM_HIP_TRY(final_sync(h->stream));
    *n_header = hp;
    *n_payload = pp;
    if (n_header_tags) *n_header_tags = nht;
    if (n_payload_tags) *n_payload_tags = npt;
    if (overflow) {
        set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}
extern "C" {
gr4pm_status gr4pm_header_payload_split_process(gr4pm_header_payload_split* h, const float* in, size_t n,
                                                float* header, size_t* n_header, float* payload,
                                                size_t* n_payload, const gr4pm_packet_tag* tags_in,
                                                size_t n_tags_in, gr4pm_packet_tag* header_tags,
                                                size_t* n_header_tags, gr4pm_packet_tag* payload_tags,
                                                size_t* n_payload_tags, size_t tags_cap)
try {
    return header_payload_split_impl<float>(h, in, n, header, n_header, payload, n_payload, tags_in, n_tags_in,
                                            header_tags, n_header_tags, payload_tags, n_payload_tags, tags_cap);
}
GR4PM_ABI_CATCH
gr4pm_status gr4pm_header_payload_split_process_c64(gr4pm_header_payload_split* h, const gr4pm_c64* in, size_t n,
                                                    gr4pm_c64* header, size_t* n_header, gr4pm_c64* payload,
                                                    size_t* n_payload, const gr4pm_packet_tag* tags_in,
                                                    size_t n_tags_in, gr4pm_packet_tag* header_tags,
                                                    size_t* n_header_tags, gr4pm_packet_tag* payload_tags,
                                                    size_t* n_payload_tags, size_t tags_cap)
try {
    return header_payload_split_impl<float2>(h, reinterpret_cast<const float2*>(in), n, reinterpret_cast<float2*>(header),
                                             n_header, reinterpret_cast<float2*>(payload), n_payload, tags_in, n_tags_in,
                                             header_tags, n_header_tags, payload_tags, n_payload_tags, tags_cap);
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_header_fec_decoder_create(const gr4pm_header_fec_decoder_params* p, gr4pm_header_fec_decoder** out)
try {
    if (!p || !out || !p->alist) return GR4PM_ERR_INVALID;
    *out = nullptr;
    GR4PM_TRY(require_device());
    // alist: n m / max column weight, max row weight / column weights / row weights /
    // column lists / row lists (1-based, optionally zero-padded to the maximum weight)
    std::vector<long> v;
    for (const char* c = p->alist; *c;) {
        if (*c >= '0' && *c <= '9') {
            char* e;
            v.push_back(std::strtol(c, &e, 10));
            c = e;
        } else {
            ++c;
        }
    }
    if (v.size() < 4 || v[0] <= 0 || v[1] <= 0 || v[0] > kMaxN || v[1] > kMaxM || v[3] > kMaxDeg || v[1] >= v[0] ||
        (v[0] - v[1]) % 8 != 0 || v[0] - v[1] > 64) {
        set_error("alist: unsupported code dimensions");
        return GR4PM_ERR_INVALID;
    }
    const unsigned n = static_cast<unsigned>(v[0]), m = static_cast<unsigned>(v[1]);
    const unsigned max_col = static_cast<unsigned>(v[2]), max_row = static_cast<unsigned>(v[3]);
    if (v.size() < 4 + static_cast<size_t>(n) + m) {
        set_error("alist: truncated");
        return GR4PM_ERR_INVALID;
    }
    size_t i = 4;
    std::vector<unsigned> colw(n), roww(m);
    size_t total_col = 0, total_row = 0;
    for (auto& w : colw) total_col += (w = static_cast<unsigned>(v[i++]));
    for (auto& w : roww) total_row += (w = static_cast<unsigned>(v[i++]));
    const size_t remaining = v.size() - i;
    const bool padded = remaining == static_cast<size_t>(n) * max_col + static_cast<size_t>(m) * max_row;
    if (!padded && remaining != total_col + total_row) {
        set_error("alist: %zu list entries, expected %zu", remaining, total_col + total_row);
        return GR4PM_ERR_INVALID;
    }
    i += padded ? static_cast<size_t>(n) * max_col : total_col;
    std::vector<uint8_t> row_var(static_cast<size_t>(m) * kMaxDeg, 0xFF), row_deg(m, 0);
    for (unsigned c = 0; c < m; ++c) {
        const unsigned cnt = padded ? max_row : roww[c];
        for (unsigned e = 0; e < cnt; ++e) {
            const long x = v[i++];
            if (x <= 0) continue;
            if (x > static_cast<long>(n) || row_deg[c] >= kMaxDeg) {
                set_error("alist: bad row entry");
                return GR4PM_ERR_INVALID;
            }
            row_var[c * kMaxDeg + row_deg[c]++] = static_cast<uint8_t>(x - 1);
        }
    }
    // layers (see the kernel's comment)
    std::vector<unsigned long long> sched;
    std::vector<bool> placed(m, false);
    unsigned n_placed = 0;
    while (n_placed < m) {
        std::vector<unsigned long long> step(64, ~0ull);
        std::vector<bool> used(n, false);
        unsigned lanes = 0;
        for (unsigned c = 0; c < m; ++c) {
            if (placed[c]) continue;
            bool clash = lanes + row_deg[c] > 64;
            for (unsigned e = 0; e < row_deg[c] && !clash; ++e) clash = used[row_var[c * kMaxDeg + e]];
            if (clash) continue;
            for (unsigned e = 0; e < row_deg[c]; ++e) {
                used[row_var[c * kMaxDeg + e]] = true;
                unsigned long long pk = 0;
                for (unsigned j = 0; j < static_cast<unsigned>(kPackDeg); ++j)
                    pk |= static_cast<unsigned long long>(row_var[c * kMaxDeg + j]) << (8 * j);
                pk |= static_cast<unsigned long long>(row_deg[c]) << 40;
                pk |= static_cast<unsigned long long>(e) << 44;
                pk |= static_cast<unsigned long long>(c) << 48;
                step[lanes++] = pk; // bits 56-63 stay 0: never equal to the idle pattern
            }
            placed[c] = true;
            ++n_placed;
        }
        if (lanes == 0) { // a check wider than a wavefront cannot be scheduled
            set_error("alist: row weight above 64");
            return GR4PM_ERR_INVALID;
        }
        sched.insert(sched.end(), step.begin(), step.end());
    }
    if (sched.size() > static_cast<size_t>(kMaxSteps) * 64) {
        set_error("alist: %zu schedule steps do not fit the kernel's table", sched.size() / 64);
        return GR4PM_ERR_INVALID;
    }
    if (p->arithmetic != 0 && p->arithmetic != 1) {
        set_error("HeaderFecDecoder: arithmetic %d (0: float32 messages, 1: 8-bit messages)", p->arithmetic);
        return GR4PM_ERR_INVALID;
    }
    std::vector<float> corr(64);
    for (int k = 0; k < 64; ++k) {
        corr[k] = static_cast<float>(std::log1p(std::exp(-k / 8.0)));
        if (p->arithmetic == 1) corr[k] = static_cast<float>(std::nearbyint(8.0 * std::log1p(std::exp(-k / 8.0))));
    }
    auto* h = new (std::nothrow) gr4pm_header_fec_decoder;
    if (!h) return GR4PM_ERR_NOMEM;
    h->arithmetic = p->arithmetic;
    h->n = n;
    h->m = m;
    h->n_steps = static_cast<unsigned>(sched.size() / 64);
    h->max_iterations = p->max_iterations;
    h->stream = static_cast<hipStream_t>(p->stream);
    gr4pm_status s = GR4PM_OK;
    auto ok = [&](gr4pm_status r) {
        if (s == GR4PM_OK) s = r;
    };
    ok(h->row_var.alloc(row_var.size()));
    ok(h->row_deg.alloc(row_deg.size()));
    ok(h->sched.alloc(sched.size()));
    ok(h->corr.alloc(corr.size()));
    if (s == GR4PM_OK) s = h->row_var.upload(row_var.data(), row_var.size(), h->stream);
    if (s == GR4PM_OK) s = h->row_deg.upload(row_deg.data(), row_deg.size(), h->stream);
    if (s == GR4PM_OK) s = h->sched.upload(sched.data(), sched.size(), h->stream);
    if (s == GR4PM_OK) s = h->corr.upload(corr.data(), corr.size(), h->stream);
    if (s == GR4PM_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = GR4PM_ERR_HIP;
    if (s != GR4PM_OK) {
        delete h;
        return s;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_header_fec_decoder_destroy(gr4pm_header_fec_decoder* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
GR4PM_ABI_CATCH_VOID
gr4pm_status gr4pm_header_fec_decoder_process(gr4pm_header_fec_decoder* h, const float* llrs, size_t n_codewords,
                                              uint8_t* headers, uint8_t* invalid)
try {
    if (!h) return GR4PM_ERR_INVALID;
    if (n_codewords == 0) return GR4PM_OK;
    if (!llrs || !headers || !invalid) {
        set_error("null pointer");
        return GR4PM_ERR_INVALID;
    }
    const size_t hb = (h->n - h->m) / 8;
    const size_t bytes = n_codewords * (hb + 1);
    if (h->d_out.n < bytes) GR4PM_TRY(h->d_out.alloc(bytes * 2));
    if (h->h_out.n < bytes) GR4PM_TRY(h->h_out.alloc(bytes * 2));
    LdpcDev d{ h->row_var.p, h->row_deg.p, h->sched.p, h->corr.p, h->n, h->m, h->n_steps };
    uint8_t* d_headers = h->d_out.p;
    uint8_t* d_invalid = h->d_out.p + n_codewords * hb;
    if (h->arithmetic == 1)
        hipLaunchKernelGGL(k_header_fec<true>, dim3(static_cast<unsigned>(n_codewords)), dim3(64), 0, h->stream, llrs,
                           static_cast<unsigned>(n_codewords), d, h->max_iterations, 2 * h->n, d_headers, d_invalid);
    else
        hipLaunchKernelGGL(k_header_fec<false>, dim3(static_cast<unsigned>(n_codewords)), dim3(64), 0, h->stream, llrs,
                           static_cast<unsigned>(n_codewords), d, h->max_iterations, 2 * h->n, d_headers, d_invalid);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipMemcpyAsync(h->h_out.p, h->d_out.p, bytes, hipMemcpyDeviceToHost, h->stream));
    GR4PM_HIP_TRY(hipStreamSynchronize(h->stream));
    std::copy_n(h->h_out.p, n_codewords * hb, headers);
    std::copy_n(h->h_out.p + n_codewords * hb, n_codewords, invalid);
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

void gr4pm_header_parse(const uint8_t* headers, const uint8_t* invalid, size_t n, gr4pm_header_msg* msgs,
                        int32_t* packet_type)
try {
    for (size_t i = 0; i < n; ++i) { // header_parser.hpp:56-85
        const uint8_t* hd = headers + 4 * i;
        bool valid = !(invalid && invalid[i]);
        const uint64_t packet_length = (static_cast<uint64_t>(hd[0]) << 8) | hd[1];
        if (packet_length == 0) valid = false;
        if (hd[2] != 0x00 && hd[2] != 0x01) valid = false;
        msgs[i].packet_length = valid ? packet_length : 0;
        msgs[i].invalid_header = valid ? 0 : 1;
        if (packet_type) packet_type[i] = valid ? hd[2] : -1;
    }
}
GR4PM_ABI_CATCH_VOID

static unsigned grid1d(size_t n) { return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 65535 * 4))); }

gr4pm_status gr4pm_binary_slicer_process(const float* in, size_t n, uint8_t* out, int invert, void* stream)
try {
    if (n == 0) return GR4PM_OK;
    if (!in || !out) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_slice, dim3(grid1d(n)), dim3(256), 0, s, in, n, out, invert);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
gr4pm_status gr4pm_pack_bits_process(const uint8_t* in, size_t n_out, uint8_t* out, size_t inputs_per_output,
                                     unsigned bits_per_input, int msb_first, void* stream)
try {
    if (inputs_per_output == 0 || bits_per_input == 0 || inputs_per_output * bits_per_input > 8) {
        set_error("inputs_per_output %zu x bits_per_input %u does not fit a byte", inputs_per_output, bits_per_input);
        return GR4PM_ERR_INVALID;
    }
    if (n_out == 0) return GR4PM_OK;
    if (!in || !out) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_pack, dim3(grid1d(n_out)), dim3(256), 0, s, in, n_out, out,
                       static_cast<unsigned>(inputs_per_output), bits_per_input, msb_first);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
} // extern "C"
// (library-internal: csrc/packet_receiver.hip) BinarySlicer + PackBits over [a[0 .. na) | b[0 ..)], no synchronisation
gr4pm_status gr4pm::slice_pack_two(const float* a, size_t na, const float* b, size_t n_out, uint8_t* out, hipStream_t s)
{
    if (n_out == 0) return GR4PM_OK;
    if (na == 0)
        hipLaunchKernelGGL(k_slice_pack, dim3(grid1d(n_out)), dim3(256), 0, s, b, n_out, out);
    else
        hipLaunchKernelGGL(k_slice_pack_two, dim3(grid1d(n_out)), dim3(256), 0, s, a, na, b, n_out, out);
    GR4PM_HIP_TRY(hipGetLastError());
    return GR4PM_OK;
}
// (library-internal: csrc/packet_receiver.hip, the packets_only receiver) the host halves of the descrambler and of
// HeaderPayloadSplit -- the blocks' state advances exactly as in their process() calls, nothing is launched -- and the
// one kernel that does their work together with SyncwordRemove's, the LLR decoder's, the slicer's and the packer's
gr4pm_status gr4pm::scrambler_plan(gr4pm_additive_scrambler* h, size_t n, const uint64_t* reset_index, size_t n_resets,
                                   std::vector<hostlogic::ScrambleRun>& runs)
{
    if (!h || h->item_kind != 1) return GR4PM_ERR_INVALID;
    runs.clear();
    return hostlogic::scramble_runs(*h, n, reset_index, n_resets, runs);
}
gr4pm_status gr4pm::header_payload_split_plan(gr4pm_header_payload_split* h, size_t n, const gr4pm_packet_tag* tags_in,
                                              size_t n_tags_in, gr4pm_packet_tag* header_tags, gr4pm_packet_tag* payload_tags,
                                              size_t tags_cap, hostlogic::HpsReplay& rp)
{
    if (!h) return GR4PM_ERR_INVALID;
    GR4PM_TRY(hostlogic::hps_replay(*h, n, tags_in, n_tags_in, header_tags, payload_tags, tags_cap, rp));
    if (rp.tag_overflow) {
        set_error("tags_cap too small");
        return GR4PM_ERR_OVERFLOW;
    }
    return GR4PM_OK;
}
gr4pm_status gr4pm::tail_fused(gr4pm_additive_scrambler* scr, DevBuf<hostlogic::TailSpan>& table,
                               const std::vector<hostlogic::TailSpan>& spans, const gr4pm_c64* symbols, float scale,
                               float* header_llr, uint8_t* packed, hipStream_t s)
{
    if (spans.empty()) return GR4PM_OK;
    if (table.n < spans.size()) GR4PM_TRY(table.alloc(spans.size() * 2));
    GR4PM_TRY(table.upload_staged(spans.data(), spans.size(), s));
    unsigned longest = 0; // lanes of the longest span: a symbol per lane (header), a byte = four symbols per lane (payload)
    for (const auto& sp : spans) longest = std::max(longest, sp.kind ? sp.n_sym / 4 + 2 : sp.n_sym);
    const unsigned gx = std::max(1u, std::min((longest + 255) / 256, 1024u));
    for (size_t first = 0; first < spans.size(); first += 65535) {
        const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, spans.size() - first));
        hipLaunchKernelGGL(k_tail_fused, dim3(gx, rows), dim3(256), 0, s, table.p + first,
                           reinterpret_cast<const float*>(symbols), scale, scr->seq.p, scr->prefix, scr->period, header_llr,
                           packed);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    return GR4PM_OK;
}
extern "C" {
gr4pm_status gr4pm_slice_pack_process(const float* in, size_t n_out, uint8_t* out, void* stream)
try {
    if (n_out == 0) return GR4PM_OK;
    if (!in || !out) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_slice_pack, dim3(grid1d(n_out)), dim3(256), 0, s, in, n_out, out);
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(final_sync(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_crc_check_create(const gr4pm_crc_check_params* p, gr4pm_crc_check** out)
try {
    if (!p || !out) return GR4PM_ERR_INVALID;
    *out = nullptr;
    if (p->num_bits < 8 || p->num_bits > 64 || p->num_bits % 8 != 0) {
        set_error("CRC number of bits must be a multiple of 8 between 8 and 64"); // crc.hpp:80-83, crc_check.hpp:79-81
        return GR4PM_ERR_INVALID;
    }
    GR4PM_TRY(require_device());
    auto* h = new (std::nothrow) gr4pm_crc_check;
    if (!h) return GR4PM_ERR_NOMEM;
    h->p = *p;
    h->mask = p->num_bits == 64 ? ~0ull : ((1ull << p->num_bits) - 1);
    // table, crc.hpp:84-116
    unsigned long long poly = p->poly;
    h->table[0] = 0;
    if (p->input_reflected) {
        poly = h->reflect(poly);
        unsigned long long crc = 1;
        size_t i = 128;
        do {
            crc = (crc & 1) ? (crc >> 1) ^ poly : crc >> 1;
            for (size_t j = 0; j < 256; j += 2 * i) h->table[i + j] = (crc ^ h->table[j]) & h->mask;
            i >>= 1;
        } while (i > 0);
    } else {
        const unsigned long long msb = 1ull << (p->num_bits - 1);
        unsigned long long crc = msb;
        size_t i = 1;
        do {
            crc = (crc & msb) ? (crc << 1) ^ poly : crc << 1;
            for (size_t j = 0; j < i; ++j) h->table[i + j] = (crc ^ h->table[j]) & h->mask;
            i <<= 1;
        } while (i < 256);
    }
    hipStream_t s = static_cast<hipStream_t>(p->stream);
    gr4pm_status st = h->d_table.alloc(256);
    if (st == GR4PM_OK) st = h->d_table.upload(h->table, 256, s);
    // slice[k][b]: the register after byte b and k zero bytes (k_crc_check_sliced); GR4PM_CRC_BYTEWISE keeps the byte loop
    std::vector<uint32_t> slice;
    h->sliced = !getenv("GR4PM_CRC_BYTEWISE") && p->num_bits <= 32 && (p->input_reflected || p->num_bits == 32);
    if (h->sliced) {
        slice.resize(8 * 256);
        for (unsigned b = 0; b < 256; ++b) {
            uint32_t v = static_cast<uint32_t>(h->table[b]);
            slice[b] = v;
            for (unsigned k = 1; k < 8; ++k) {
                v = p->input_reflected ? static_cast<uint32_t>(h->table[v & 0xff]) ^ (v >> 8)
                                       : static_cast<uint32_t>(h->table[(v >> 24) & 0xff]) ^ (v << 8);
                slice[k * 256 + b] = v;
            }
        }
        if (st == GR4PM_OK) st = h->d_slice.alloc(slice.size());
        if (st == GR4PM_OK) st = h->d_slice.upload(slice.data(), slice.size(), s);
    }
    if (st == GR4PM_OK && hipStreamSynchronize(s) != hipSuccess) st = GR4PM_ERR_HIP;
    if (st != GR4PM_OK) {
        delete h;
        return st;
    }
    *out = h;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH
void gr4pm_crc_check_destroy(gr4pm_crc_check* h)
try {
    if (!h) return;
    (void)hipStreamSynchronize(static_cast<hipStream_t>(h->p.stream));
    delete h;
}
GR4PM_ABI_CATCH_VOID
uint64_t gr4pm_crc_check_compute(const gr4pm_crc_check* h, const uint8_t* data, size_t n)
try {
    unsigned long long rem = h->p.initial_value & h->mask; // crc.hpp:119-156
    if (h->p.input_reflected) {
        for (size_t k = 0; k < n; ++k) rem = h->table[(rem ^ data[k]) & 0xff] ^ (rem >> 8);
    } else {
        for (size_t k = 0; k < n; ++k)
            rem = (h->table[((rem >> (h->p.num_bits - 8)) ^ data[k]) & 0xff] ^ (rem << 8)) & h->mask;
    }
    if ((h->p.input_reflected != 0) != (h->p.result_reflected != 0)) rem = h->reflect(rem);
    return rem ^ (h->p.final_xor & h->mask);
}
GR4PM_ABI_CATCH_RET(0)
gr4pm_status gr4pm_crc_check_process(gr4pm_crc_check* h, const uint8_t* in, const uint64_t* packet_offset,
                                     const uint64_t* packet_len, size_t n_packets, uint8_t* out,
                                     uint64_t* out_len, size_t* n_out_bytes)
try {
    if (!h || !n_out_bytes) return GR4PM_ERR_INVALID;
    *n_out_bytes = 0;
    if (n_packets == 0) return GR4PM_OK;
    if (!in || !out || !packet_offset || !packet_len || !out_len) {
        set_error("null pointer");
        return GR4PM_ERR_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(h->p.stream);
    std::vector<CrcPacket> pk(n_packets);
    for (size_t i = 0; i < n_packets; ++i) {
        if (packet_len[i] == 0) {
            set_error("received packet-length equal to zero"); // crc_check.hpp:131-136
            return GR4PM_ERR_INVALID;
        }
        pk[i] = { packet_offset[i], packet_len[i] };
    }
    if (h->pk.n < n_packets) GR4PM_TRY(h->pk.alloc(n_packets * 2));
    if (h->ok.n < n_packets) GR4PM_TRY(h->ok.alloc(n_packets * 2));
    if (h->ok_host.n < n_packets) GR4PM_TRY(h->ok_host.alloc(n_packets * 2));
    GR4PM_TRY(h->pk.upload_staged(pk.data(), n_packets, s));
    const dim3 crc_grid(static_cast<unsigned>((n_packets + 63) / 64));
    if (h->sliced && h->p.input_reflected)
        hipLaunchKernelGGL(k_crc_check_sliced<true>, crc_grid, dim3(64), 0, s, in, h->pk.p, static_cast<unsigned>(n_packets),
                           h->d_slice.p, h->p.num_bits, h->mask, h->p.initial_value & h->mask, h->p.final_xor & h->mask,
                           h->p.result_reflected ? 1 : 0, h->p.swap_endianness ? 1 : 0, h->p.skip_header_bytes, h->ok.p);
    else if (h->sliced)
        hipLaunchKernelGGL(k_crc_check_sliced<false>, crc_grid, dim3(64), 0, s, in, h->pk.p, static_cast<unsigned>(n_packets),
                           h->d_slice.p, h->p.num_bits, h->mask, h->p.initial_value & h->mask, h->p.final_xor & h->mask,
                           h->p.result_reflected ? 1 : 0, h->p.swap_endianness ? 1 : 0, h->p.skip_header_bytes, h->ok.p);
    else
        hipLaunchKernelGGL(k_crc_check, crc_grid, dim3(64), 0, s, in, h->pk.p,
                           static_cast<unsigned>(n_packets), h->d_table.p, h->p.num_bits, h->mask,
                           h->p.initial_value & h->mask, h->p.final_xor & h->mask, h->p.input_reflected ? 1 : 0,
                           h->p.result_reflected ? 1 : 0, h->p.swap_endianness ? 1 : 0, h->p.skip_header_bytes, h->ok.p);
    GR4PM_HIP_TRY(hipGetLastError());
    // passing packets leave back to back (:180-202): the spans are made on the device (round 5: the verdicts used to go
    // to the host first -- one more wait for a busy chip in the stage that sets the pace of the decode_headers pipeline)
    const size_t crc_bytes = h->p.num_bits / 8;
    if (h->spans.n < n_packets) GR4PM_TRY(h->spans.alloc(n_packets * 2));
    hipLaunchKernelGGL(k_crc_spans, dim3(1), dim3(1024), 0, s, h->pk.p, h->ok.p, static_cast<unsigned>(n_packets),
                       h->p.discard_crc ? static_cast<unsigned>(crc_bytes) : 0u, h->spans.p);
    unsigned long long longest = 0;
    for (size_t i = 0; i < n_packets; ++i) longest = std::max<unsigned long long>(longest, packet_len[i]);
    const unsigned gx = static_cast<unsigned>(std::max<unsigned long long>(
        1, std::min<unsigned long long>((longest + 2047) / 2048, 1024)));
    for (size_t first = 0; first < n_packets; first += 65535) {
        const unsigned rows = static_cast<unsigned>(std::min<size_t>(65535, n_packets - first));
        hipLaunchKernelGGL(k_gather_u8, dim3(gx, rows), dim3(256), 0, s, h->spans.p + first, in, out);
    }
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipMemcpyAsync(h->ok_host.p, h->ok.p, n_packets, hipMemcpyDeviceToHost, s));
    GR4PM_HIP_TRY(hipStreamSynchronize(s));
    size_t opos = 0;
    for (size_t i = 0; i < n_packets; ++i) {
        out_len[i] = 0;
        if (!h->ok_host.p[i]) continue;
        const unsigned long long n = h->p.discard_crc ? packet_len[i] - crc_bytes : packet_len[i];
        out_len[i] = n;
        opos += n;
    }
    *n_out_bytes = opos;
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_mapper_process(const uint8_t* in, size_t n, void* out, const void* map_host, size_t map_size,
                                  int item_kind, void* stream)
try {
    if (map_size == 0 || (map_size & (map_size - 1)) || map_size > 256) {
        set_error("the map size must be a power of 2 (got %zu)", map_size); // mapper.hpp:37-41
        return GR4PM_ERR_INVALID;
    }
    if (n == 0) return GR4PM_OK;
    if (!in || !out || !map_host || (item_kind != 0 && item_kind != 1)) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t item = item_kind == 0 ? 8 : 4;
    DevBuf<uint8_t> dmap;
    GR4PM_TRY(dmap.alloc(map_size * item));
    GR4PM_TRY(dmap.upload(static_cast<const uint8_t*>(map_host), map_size * item, s));
    if (item_kind == 0)
        hipLaunchKernelGGL(k_mapper<float2>, dim3(grid1d(n)), dim3(256), 0, s, in, n,
                           reinterpret_cast<const float2*>(dmap.p), static_cast<unsigned>(map_size - 1),
                           static_cast<float2*>(out));
    else
        hipLaunchKernelGGL(k_mapper<float>, dim3(grid1d(n)), dim3(256), 0, s, in, n,
                           reinterpret_cast<const float*>(dmap.p), static_cast<unsigned>(map_size - 1),
                           static_cast<float*>(out));
    GR4PM_HIP_TRY(hipGetLastError());
    GR4PM_HIP_TRY(hipStreamSynchronize(s));
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

gr4pm_status gr4pm_burst_shaper_process(const void* in, size_t n, void* out, int item_kind, const float* leading_host,
                                        size_t leading_n, const float* trailing_host, size_t trailing_n,
                                        const uint64_t* packet_offset, const uint64_t* packet_len, size_t n_packets,
                                        void* stream)
try {
    if (n == 0) return GR4PM_OK;
    if (!in || !out || (item_kind != 0 && item_kind != 1)) return GR4PM_ERR_INVALID;
    GR4PM_TRY(require_device());
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t item = item_kind == 0 ? 8 : 4;
    if (in != out) GR4PM_HIP_TRY(hipMemcpyAsync(out, in, n * item, hipMemcpyDeviceToDevice, s));
    if (n_packets) {
        std::vector<ShapePacket> pk(n_packets);
        for (size_t i = 0; i < n_packets; ++i) {
            if (packet_len[i] == 0) {
                set_error("received packet-length equal to zero"); // burst_shaper.hpp:84-89
                return GR4PM_ERR_INVALID;
            }
            if (packet_offset[i] + packet_len[i] > n) {
                set_error("packet %zu runs past the %zu items of this call", i, n);
                return GR4PM_ERR_INVALID;
            }
            pk[i] = { packet_offset[i], packet_len[i] };
        }
        DevBuf<ShapePacket> dpk;
        DevBuf<float> dl, dt;
        GR4PM_TRY(dpk.alloc(n_packets));
        GR4PM_TRY(dl.alloc(std::max<size_t>(leading_n, 1)));
        GR4PM_TRY(dt.alloc(std::max<size_t>(trailing_n, 1)));
        GR4PM_TRY(dpk.upload(pk.data(), n_packets, s));
        if (leading_n) GR4PM_TRY(dl.upload(leading_host, leading_n, s));
        if (trailing_n) GR4PM_TRY(dt.upload(trailing_host, trailing_n, s));
        const dim3 grid(static_cast<unsigned>(n_packets));
        if (item_kind == 0)
            hipLaunchKernelGGL(k_burst_edges<float2>, grid, dim3(64), 0, s, static_cast<const float2*>(in),
                               static_cast<float2*>(out), dpk.p, dl.p, static_cast<unsigned>(leading_n), dt.p,
                               static_cast<unsigned>(trailing_n));
        else
            hipLaunchKernelGGL(k_burst_edges<float>, grid, dim3(64), 0, s, static_cast<const float*>(in),
                               static_cast<float*>(out), dpk.p, dl.p, static_cast<unsigned>(leading_n), dt.p,
                               static_cast<unsigned>(trailing_n));
        GR4PM_HIP_TRY(hipGetLastError());
        GR4PM_HIP_TRY(hipStreamSynchronize(s));
    } else {
        GR4PM_HIP_TRY(hipStreamSynchronize(s));
    }
    return GR4PM_OK;
}
GR4PM_ABI_CATCH

} // extern "C"
