// fft2048_wave.hpp -- 2048-point complex FFT executed by ONE 64-lane wavefront, 32 points
// per lane, two LDS exchanges per transform and no workgroup barrier.
//
// Two schedules are provided so that the correlator never re-distributes a spectrum:
//   FFT-1 (16 x 16 x 8): input read straight from HBM as 16-byte loads
//                         (lane l owns samples 2l, 2l+1 (+128 n1)); output X[k] lands with
//                         k = k1 + 16 k2 + 256 k3, (k1, k2) = (l/16 + 4q, l%16), k3 = 0..7.
//   FFT-2 (8 x 16 x 16): takes its input in exactly FFT-1's output distribution (so the
//                         product X * template needs no exchange) and leaves output index
//                         k = l + 64 q + 128 k3: consecutive lanes hold consecutive lags, so
//                         the correlation power is stored to HBM coalesced.
// Replaces: gr::algorithm::FFTw<c64,c64>::compute as used by
//   syncword_detection.hpp:184,239-241,250-251 (forward, un-normalised, e^{-j 2 pi nk/N}).
//
// Every phase is a plain function of (lane, registers, LDS) and is compiled for host too
// (tests/ emulate the 64 lanes on the CPU to check the index algebra without a GPU).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GR4PM_HD __host__ __device__ __forceinline__
#else
#define GR4PM_HD inline
#endif

namespace gr4pm {

// complex<float> as a 2-lane vector so that every complex add / twiddle multiply maps onto
// one or two packed VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with
// op_sel / neg modifiers) instead of scalar pairs plus register shuffles.
#if defined(__clang__)
typedef float cf __attribute__((ext_vector_type(2)));
GR4PM_HD cf mk(float x, float y) { return cf{ x, y }; }
GR4PM_HD cf swap_xy(cf a) { return a.yx; }
GR4PM_HD cf dup_x(cf a) { return a.xx; }
GR4PM_HD cf dup_y(cf a) { return a.yy; }
#else
struct cf {
    float x, y;
};
GR4PM_HD cf mk(float x, float y) { return cf{ x, y }; }
GR4PM_HD cf operator+(cf a, cf b) { return { a.x + b.x, a.y + b.y }; }
GR4PM_HD cf operator-(cf a, cf b) { return { a.x - b.x, a.y - b.y }; }
GR4PM_HD cf operator*(cf a, cf b) { return { a.x * b.x, a.y * b.y }; }
GR4PM_HD cf operator*(float a, cf b) { return { a * b.x, a * b.y }; }
GR4PM_HD cf operator-(cf a) { return { -a.x, -a.y }; }
GR4PM_HD cf swap_xy(cf a) { return { a.y, a.x }; }
GR4PM_HD cf dup_x(cf a) { return { a.x, a.x }; }
GR4PM_HD cf dup_y(cf a) { return { a.y, a.y }; }
#endif
// (x, y) -> (x, -y) and (-x, y)
GR4PM_HD cf neg_y(cf a) { return mk(a.x, -a.y); }
GR4PM_HD cf neg_x(cf a) { return mk(-a.x, a.y); }
GR4PM_HD cf mul_mj(cf a) { return neg_y(swap_xy(a)); } // a * (-j) = (a.y, -a.x)

#if defined(__HIP_DEVICE_COMPILE__)
// The swizzle (op_sel) and per-half sign (neg_lo / neg_hi) source modifiers of the packed
// FP32 instructions make "b +- (-j) a" one instruction and a full complex multiply two; hipcc
// does not fold a per-half negation by itself (it emits v_xor), hence the explicit forms.
// b + (-j) a = (b.x + a.y, b.y - a.x): one v_pk_fma_f32 whose per-half sign is a constant operand (a real
// instruction, not an asm: hipcc pads an s_nop between an asm statement and whatever reads its result)
__device__ __forceinline__ cf add_mj(cf b, cf a) { return __builtin_elementwise_fma(a.yx, cf{ 1.0f, -1.0f }, b); }
// b - (-j) a = (b.x - a.y, b.y + a.x)
__device__ __forceinline__ cf sub_mj(cf b, cf a) { return __builtin_elementwise_fma(a.yx, cf{ -1.0f, 1.0f }, b); }
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x) for a run-time twiddle w
__device__ __forceinline__ cf cmul(cf a, cf w)
{
    // ONE statement for the dependent pair: between two separate asm statements hipcc pads the dependency with
    // an s_nop (it does not model what is inside an asm), and an s_nop costs a whole issue slot -- 80 of them per
    // transform of the correlator; the hardware interlocks a VALU result by itself
    // the product of the first instruction sits in the result register itself: a scratch output would be given the
    // same physical register in consecutive statements, and hipcc pads every register overlap between two asm
    // statements with an s_nop (gfx940 dst-forwarding hazard, assumed for whatever an asm defines)
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "v"(w));
    return r;
}
#else
GR4PM_HD cf add_mj(cf b, cf a) { return b + mul_mj(a); }
GR4PM_HD cf sub_mj(cf b, cf a) { return b - mul_mj(a); }
GR4PM_HD cf cmul(cf a, cf w) { return dup_x(a) * w + dup_y(a) * neg_x(swap_xy(w)); }
#endif
// multiply by a compile-time constant twiddle: plain arithmetic, the compiler folds the signs
GR4PM_HD cf cmulc(cf a, cf w) { return dup_x(a) * w + dup_y(a) * neg_x(swap_xy(w)); }
GR4PM_HD float cnorm(cf a)
{
    const cf s = a * a;
    return s.x + s.y;
}

constexpr int kFftN = 2048;
constexpr int kLanes = 64;
constexpr int kPtsPerLane = 32;
// Row strides (complex items) of the LDS exchange layouts; every exchange runs in two
// half-rounds (8 of 16 rows / 64 of 128 rows at a time) so one wave needs 9 KiB, not 18.
constexpr int kS1 = 136; // FFT-1 exchange 1: 8 rows [k1] of 128 (+8 pad) per half
constexpr int kS2 = 18;  // FFT-1 exchange 2: 64 rows [k1][m] of 16 (+2 pad) per half
constexpr int kSA = 18;  // FFT-2 exchange A: 64 rows [ka][m] of 16 (+2 pad) per half
constexpr int kSB = 18;  // FFT-2 exchange B: 64 rows [k2][ka] of 16 (+2 pad) per half
constexpr int kExchangeItems = 64 * 18; // 1152 complex = 9 KiB per wave

// Per-lane-ordered twiddle tables (built once on the host, double -> float):
//   tw1a[(k1-1)*2 + e][lane] = W2048^((2 lane + e) k1)      k1 = 1..15   (30 x 64)
//   tw1b[(k2-1)][m]          = W128^(m k2)                   k2 = 1..15, m < 8
//   twA[(ka-1)*4 + q][lane]  = W2048^(n2 ka), n2 = lane/16 + 4q + 16 (lane%16), ka = 1..7 (28 x 64)
//   twB[(k2-1)][m]           = W256^(m k2)                   k2 = 1..15, m < 16
constexpr int kTw1aItems = 30 * 64, kTw1bItems = 15 * 8, kTwAItems = 28 * 64, kTwBItems = 15 * 16;
// fills the four tables; w(k) must return exp(-j 2 pi k / 2048) as cf
template <typename W>
inline void build_twiddle_tables(W w, cf* tw1a, cf* tw1b, cf* twA, cf* twB)
{
    for (int k1 = 1; k1 < 16; ++k1)
        for (int e = 0; e < 2; ++e)
            for (int lane = 0; lane < 64; ++lane) tw1a[((k1 - 1) * 2 + e) * 64 + lane] = w(((2 * lane + e) * k1) % kFftN);
    for (int k2 = 1; k2 < 16; ++k2)
        for (int m = 0; m < 8; ++m) tw1b[(k2 - 1) * 8 + m] = w((16 * m * k2) % kFftN);
    for (int ka = 1; ka < 8; ++ka)
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int n2 = (lane >> 4) + 4 * q + 16 * (lane & 15);
                twA[((ka - 1) * 4 + q) * 64 + lane] = w((n2 * ka) % kFftN);
            }
    for (int k2 = 1; k2 < 16; ++k2)
        for (int m = 0; m < 16; ++m) twB[(k2 - 1) * 16 + m] = w((8 * m * k2) % kFftN);
}

// ---- small DFTs, forward sign, natural-order output, everything in registers ----
template <typename T>
GR4PM_HD void dft4(T& a0, T& a1, T& a2, T& a3)
{
    const T t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a1 = add_mj(t1, d); // t1 + (-j) d
    a2 = t0 - t2;
    a3 = sub_mj(t1, d);
}
GR4PM_HD void dft8(cf* v)
{
    constexpr float c = 0.70710678118654752440f;
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    const cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const cf o0 = v[1], o2s = v[5];
    const cf o1 = c * add_mj(v[3], v[3]);    // * W8^1 = c (1 - j)
    const cf o3 = (-c) * sub_mj(v[7], v[7]); // * W8^3 = -c (1 + j)
    v[0] = e0 + o0;
    v[1] = e1 + o1;
    v[2] = add_mj(e2, o2s); // * W8^2 = -j
    v[3] = e3 + o3;
    v[4] = e0 - o0;
    v[5] = e1 - o1;
    v[6] = sub_mj(e2, o2s);
    v[7] = e3 - o3;
}
GR4PM_HD void dft16(cf* v)
{
    constexpr float c8 = 0.70710678118654752440f;
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f; // pi/8
    cf e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = v[2 * i];
        o[i] = v[2 * i + 1];
    }
    dft8(e);
    dft8(o);
    // o[k] *= W16^k = cos(pi k/8) - j sin(pi k/8)
    o[1] = cmulc(o[1], mk(c1, -s1));
    o[2] = c8 * add_mj(o[2], o[2]);
    o[3] = cmulc(o[3], mk(s1, -c1));
    o[5] = cmulc(o[5], mk(-s1, -c1));
    o[6] = (-c8) * sub_mj(o[6], o[6]);
    o[7] = cmulc(o[7], mk(-c1, -s1));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k == 4) { // W16^4 = -j
            v[4] = add_mj(e[4], o[4]);
            v[12] = sub_mj(e[4], o[4]);
        } else {
            v[k] = e[k] + o[k];
            v[k + 8] = e[k] - o[k];
        }
    }
}

// ======================================================================= FFT-1
// r[2*n1 + e] = x[2*lane + e + 128*n1] on entry.
GR4PM_HD void fft1_pass1(int lane, cf* r, const cf* tw1a)
{
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        cf v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = r[2 * n1 + e];
        dft16(v);
        r[e] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) r[2 * k1 + e] = cmul(v[k1], tw1a[((k1 - 1) * 2 + e) * 64 + lane]);
    }
}
// half h: rows k1 in [8h, 8h+8)
GR4PM_HD void fft1_store1(int lane, const cf* r, cf* lds, int h)
{
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k1 = 8 * h + kk;
        lds[kk * kS1 + 2 * lane] = r[2 * k1];
        lds[kk * kS1 + 2 * lane + 1] = r[2 * k1 + 1];
    }
}
// half h fills r[16h .. 16h+16): combo c = lane + 64h, k1 = c/8, m = c%8
GR4PM_HD void fft1_load2(int lane, cf* r, const cf* lds, int h)
{
    const int kk = lane >> 3, m = lane & 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[16 * h + i] = lds[kk * kS1 + m + 8 * i];
}
GR4PM_HD void fft1_pass2(int lane, cf* r, const cf* tw1b)
{
    const int m = lane & 7;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        dft16(r + 16 * q);
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) r[16 * q + k2] = cmul(r[16 * q + k2], tw1b[(k2 - 1) * 8 + m]);
    }
}
// half h: rows (k1*8 + m) with k1 in [8h, 8h+8) == combo c = lane + 64h -> local row = lane
GR4PM_HD void fft1_store2(int lane, const cf* r, cf* lds, int h)
{
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[lane * kS2 + k2] = r[16 * h + k2];
}
// half h fills r[8q .. 8q+8) for q in {2h, 2h+1}: k1 = lane/16 + 4q, k2 = lane%16
GR4PM_HD void fft1_load3(int lane, cf* r, const cf* lds, int h)
{
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * h + qq;
        const int k1loc = (lane >> 4) + 4 * qq, k2 = lane & 15;
#pragma unroll
        for (int m = 0; m < 8; ++m) r[8 * q + m] = lds[(k1loc * 8 + m) * kS2 + k2];
    }
}
// on exit r[8*q + k3] = X[k1 + 16*k2 + 256*k3], (k1, k2) = ((lane + 64 q) / 16, lane % 16)
GR4PM_HD void fft1_pass3(cf* r)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) dft8(r + 8 * q);
}
// index of the spectrum bin held in r[j] after FFT-1 (== input index expected by FFT-2)
GR4PM_HD int fft1_out_index(int lane, int j)
{
    const int q = j >> 3, k3 = j & 7;
    return (lane >> 4) + 4 * q + 16 * (lane & 15) + 256 * k3;
}

// ======================================================================= FFT-2
// r[8*q + n1] = P[n2 + 256*n1], n2 = (lane/16 + 4q) + 16*(lane%16) on entry.
GR4PM_HD void fft2_passA(int lane, cf* r, const cf* twA)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        dft8(r + 8 * q);
#pragma unroll
        for (int ka = 1; ka < 8; ++ka)
#if defined(GR4PM_ABL) && (GR4PM_ABL == 3 || GR4PM_ABL == 4)
            r[8 * q + ka] = cmul(r[8 * q + ka], mk(0.7f + ka, 0.7f - q));
#else
            r[8 * q + ka] = cmul(r[8 * q + ka], twA[((ka - 1) * 4 + q) * 64 + lane]);
#endif
    }
}
// half h: rows (ka*16 + m) with ka in [4h, 4h+4)
GR4PM_HD void fft2_storeA(int lane, const cf* r, cf* lds, int h)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = (lane >> 4) + 4 * q, i = lane & 15;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) lds[(kk * 16 + m) * kSA + i] = r[8 * q + 4 * h + kk];
    }
}
// half h fills r[16h .. 16h+16): combo c = lane + 64h -> local row = lane
GR4PM_HD void fft2_loadB(int lane, cf* r, const cf* lds, int h)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) r[16 * h + i] = lds[lane * kSA + i];
}
GR4PM_HD void fft2_passB(int lane, cf* r, const cf* twB)
{
    const int m = lane & 15;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        dft16(r + 16 * q);
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2)
#if defined(GR4PM_ABL) && (GR4PM_ABL == 3 || GR4PM_ABL == 4)
            r[16 * q + k2] = cmul(r[16 * q + k2], mk(0.5f + k2, 0.25f));
#else
            r[16 * q + k2] = cmul(r[16 * q + k2], twB[(k2 - 1) * 16 + m]);
#endif
    }
}
// half h: rows (k2*8 + ka) with k2 in [8h, 8h+8)
GR4PM_HD void fft2_storeB(int lane, const cf* r, cf* lds, int h)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + 64 * q, ka = c >> 4, m = c & 15;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) lds[(kk * 8 + ka) * kSB + m] = r[16 * q + 8 * h + kk];
    }
}
// half h fills r[16h .. 16h+16): combo c = lane + 64h -> local row = lane
GR4PM_HD void fft2_loadC(int lane, cf* r, const cf* lds, int h)
{
#pragma unroll
    for (int m = 0; m < 16; ++m) r[16 * h + m] = lds[lane * kSB + m];
}
// on exit r[16*q + k3] = C[lane + 64*q + 128*k3]
GR4PM_HD void fft2_passC(cf* r)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) dft16(r + 16 * q);
}
GR4PM_HD int fft2_out_index(int lane, int j)
{
    const int q = j >> 4, k3 = j & 15;
    return lane + 64 * q + 128 * k3;
}

} // namespace gr4pm
