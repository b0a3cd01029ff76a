// fft2048_w64.hpp -- 2048-point complex FFT by ONE 64-lane wavefront with ONE LDS exchange per
// transform (fft2048_wave.hpp needs two).  2048 = 32 (in-lane) x 64 (lanes):
//
//   distribution (input AND output): register j of lane l holds v[l + 64 j], j = 0..31
//   pass A   in-lane DFT-32 over j:           A_l[k1] = sum_j v[l + 64 j] W32^(j k1)
//   exchange lane l writes A_l[k1] to LDS row k1, column l, planar (re[64] | im[64] | 4 pad):
//            64 x ds_write_addtid_b32 (address = M0 + offset + 4 lane: no address VGPR, 128 B/clk,
//            MI355X_MICROARCH.md LDS table) instead of 32 x ds_write2_b64 (79 B/clk) twice
//   mid      lane L = k + 32 q reads the WHOLE row k (both halves of the 64 columns, 32 x
//            ds_read_b128, conflict-free thanks to the 4-dword pad) and forms, for m = 0..31,
//                b[m] = T[m] (A_m[k] + c A_{m+32}[k]),  c = W64^L,  T[m] = W2048^(m L)
//            i.e. the radix-2 step ACROSS lanes (sign (-1)^q and W64^k folded into the lane
//            constant c) and the 2048-point twiddle in one go: 8 scalar FMAs per m
//   pass B   in-lane DFT-32 over m:           X[L + 64 k3] = sum_m b[m] W32^(m k3)
//
// Derivation: n = l + 64 j, k = k1 + 32 k2 gives W2048^(nk) = W32^(j k1) W2048^(l k1) W64^(l k2);
// with l = m + 32 p and k2 = q + 2 k3: W64^(l k2) = W64^(m q) W32^(m k3) (-1)^(p q) and
// W2048^(l k1) = W2048^(m k1) W64^(p k1), so the p-sum is A_m + (-1)^q W64^k1 A_{m+32} and the
// remaining factor is W64^(m q) W2048^(m k1) = W2048^(m (k1 + 32 q)).
//
// Because input and output distributions coincide, the correlator uses the same schedule for
// the forward transform of the samples and for the per-bin transform of X .* template, and the
// output lands with consecutive lags on consecutive lanes (coalesced stores).
// Replaces gr::algorithm::FFTw<c64,c64>::compute as used by syncword_detection.hpp:239-241,250-251.
//
// The phases are plain functions of (lane, registers, LDS image) and compile for the host too
// (tests/fft_w64_emu.cpp runs the 64 lanes on the CPU).
#pragma once
#include "fft2048_wave.hpp"

namespace gr4pm {

constexpr int kW64Row = 132;               // dwords per exchange row: re[64] | im[64] | 4 pad
constexpr int kW64BufDwords = 32 * kW64Row; // 4224 dwords = 16896 B per wave
constexpr int kW64TwFloat4 = 8 * 2 * 64;   // T table: [g = m/4][plane][lane] float4 = 16 KiB

#if defined(__HIPCC__)
using f4 = float4;
GR4PM_HD f4 mkf4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
#else
struct f4 {
    float x, y, z, w;
};
GR4PM_HD f4 mkf4(float a, float b, float c, float d) { return f4{ a, b, c, d }; }
#endif

// tT[(g * 2 + plane) * 64 + lane] = plane(W2048^((4g + e) lane)), e = 0..3; cc[lane] = W64^lane
template <typename W>
inline void build_w64_tables(W w, f4* tT, cf* cc)
{
    for (int g = 0; g < 8; ++g)
        for (int lane = 0; lane < 64; ++lane) {
            float re[4], im[4];
            for (int e = 0; e < 4; ++e) {
                const cf t = w(((4 * g + e) * lane) % kFftN);
                re[e] = t.x;
                im[e] = t.y;
            }
            tT[(g * 2 + 0) * 64 + lane] = mkf4(re[0], re[1], re[2], re[3]);
            tT[(g * 2 + 1) * 64 + lane] = mkf4(im[0], im[1], im[2], im[3]);
        }
    for (int lane = 0; lane < 64; ++lane) cc[lane] = w((32 * lane) % kFftN);
}

// ---- twiddled radix-2 butterflies in three packed instructions -------------------------------------------------
// (u, v) = (a + w b, a - w b) with the twiddle written as  w = g rho (1 + j t),  g in {1, -j},  |t| <= 1:
//     c = b + t (j b)        = (b.x - t b.y, b.y + t b.x)          one v_pk_fma_f32 (swizzled b, one half negated)
//     g = 1 :  u = a + rho c,            v = a - rho c             two v_pk_fma_f32
//     g = -j:  u = a + rho (c.y, -c.x),  v = a - rho (c.y, -c.x)   two v_pk_fma_f32 (swizzled c)
// i.e. the complex multiply (two packed instructions) and the two additions (two more) fold into three
// multiply-adds (Linzer / Feig's form of the butterfly): the in-lane DFT-32 needs 194 packed instructions
// instead of 214.  K = (rho, |t|) lives in an SGPR pair; the sign of t is a source modifier (TNEG).  g = -1 is
// the g = 1 form with u and v exchanged, so every W32^k, k = 1 .. 15, is one of these with
// (rho, |t|) in {(cos, tan)(pi/16), (cos, tan)(pi/8), (cos, tan)(3 pi/16), (sqrt(1/2), 1)}.
#if defined(__clang__)
GR4PM_HD cf vfma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); } // one v_pk_fma_f32, never split
#else
GR4PM_HD cf vfma(cf a, cf b, cf c) { return mk(a.x * b.x + c.x, a.y * b.y + c.y); }
#endif
template <bool MJ, bool TNEG>
GR4PM_HD void bf_tw(cf a, cf b, cf K, cf& u, cf& v)
{
    // plain vector arithmetic: every per-half sign sits in a compile-time constant (an SGPR pair), the swizzles
    // become op_sel, and hipcc schedules real instructions -- between inline-asm statements it pads every register
    // overlap with an s_nop (gfx940's dst-forwarding hazard is assumed for anything an asm defines).  Explicit
    // fma: a * b + c forms would be re-associated into one shared product and two additions (four instructions).
    const float t = TNEG ? -K.y : K.y;
    const cf c = vfma(swap_xy(b), mk(-t, t), b);
    if (MJ) {
        const cf d = swap_xy(c);
        u = vfma(d, mk(K.x, -K.x), a);
        v = vfma(d, mk(-K.x, K.x), a);
    } else {
        u = vfma(c, mk(K.x, K.x), a);
        v = vfma(c, mk(-K.x, -K.x), a);
    }
}
// (rho, |t|) of the four twiddle magnitudes
#define GR4PM_K1 mk(0.98078528040323044913f, 0.19891236737965800691f) /* pi/16 */
#define GR4PM_K2 mk(0.92387953251128675613f, 0.41421356237309504880f) /* pi/8 */
#define GR4PM_K3 mk(0.83146961230254523708f, 0.66817863791929891999f) /* 3 pi/16 */
#define GR4PM_K4 mk(0.70710678118654752440f, 1.0f)                    /* pi/4 */

// ---- two complex numbers side by side, planar: r = (re0, re1), i = (im0, im1) -----------------------------------
// What the mid stage's ds_read_b128 deliver is planar (four consecutive re, four consecutive im): with this type the
// per-bin pass B works on (re[m], re[m + 1]) pairs as they come -- no v_pk_mov_b32 to re-pair them into (re, im) --,
// a multiplication by -j is a renaming, every sign is a whole-vector negation or sits in a constant, and the power
// of two outputs is one v_pk_mul_f32 + one v_pk_fma_f32.  The two halves run the SAME small DFT on two different
// sequences (even and odd items); only the last radix-2 stage combines the halves of one register.
struct pc {
    cf r, i;
};
GR4PM_HD pc operator+(pc a, pc b) { return pc{ a.r + b.r, a.i + b.i }; }
GR4PM_HD pc operator-(pc a, pc b) { return pc{ a.r - b.r, a.i - b.i }; }
GR4PM_HD pc add_mj(pc b, pc a) { return pc{ b.r + a.i, b.i - a.r }; } // b + (-j) a
GR4PM_HD pc sub_mj(pc b, pc a) { return pc{ b.r - a.i, b.i + a.r }; } // b - (-j) a
template <bool MJ, bool TNEG>
GR4PM_HD void bf_tw(pc a, pc b, cf K, pc& u, pc& v)
{
    const float t = TNEG ? -K.y : K.y, rho = K.x;
    const cf cr = vfma(b.i, mk(-t, -t), b.r), ci = vfma(b.r, mk(t, t), b.i); // c = b + t (j b)
    if (MJ) { // w b = rho (c.i, -c.r)
        u = pc{ vfma(ci, mk(rho, rho), a.r), vfma(cr, mk(-rho, -rho), a.i) };
        v = pc{ vfma(ci, mk(-rho, -rho), a.r), vfma(cr, mk(rho, rho), a.i) };
    } else {
        u = pc{ vfma(cr, mk(rho, rho), a.r), vfma(ci, mk(rho, rho), a.i) };
        v = pc{ vfma(cr, mk(-rho, -rho), a.r), vfma(ci, mk(-rho, -rho), a.i) };
    }
}
// last radix-2 stage of the planar DFT-32: x = (E, O) side by side; out = (E + w O, E - w O) side by side, i.e.
// outputs k (lo) and k + 16 (hi); w = g rho (1 + j t), g = 1 / -j (MJ) / -1 (NEG)
template <bool MJ, bool TNEG, bool NEG>
GR4PM_HD pc bf_last(pc x, cf K)
{
    const float t = TNEG ? -K.y : K.y, rho = NEG ? -K.x : K.x;
    const cf cr = vfma(x.i, mk(-t, -t), x.r), ci = vfma(x.r, mk(t, t), x.i); // hi halves: c = O + t (j O)
    const cf er = dup_x(x.r), ei = dup_x(x.i);
    if (MJ) return pc{ vfma(dup_y(ci), mk(rho, -rho), er), vfma(dup_y(cr), mk(-rho, rho), ei) };
    return pc{ vfma(dup_y(cr), mk(rho, -rho), er), vfma(dup_y(ci), mk(rho, -rho), ei) };
}

// W = exp(-j theta): theta in (0, pi/4]: g = 1, t = -tan(theta); around pi/2: g = -j, t = -tan(theta - pi/2);
// theta in [3 pi/4, pi): g = -1 (u, v exchanged), t = -tan(theta - pi)
template <typename T>
GR4PM_HD void dft8f(T* v)
{
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    const T e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const T o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    bf_tw<false, true>(e1, o1, GR4PM_K4, v[1], v[5]);  // W8^1: theta = pi/4
    v[2] = add_mj(e2, o2);                             // W8^2 = -j
    v[6] = sub_mj(e2, o2);
    bf_tw<false, false>(e3, o3, GR4PM_K4, v[7], v[3]); // W8^3: theta = 3 pi/4 -> g = -1, t = +1
}
template <typename T>
GR4PM_HD void dft16f(T* v)
{
    T e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = v[2 * i];
        o[i] = v[2 * i + 1];
    }
    dft8f(e);
    dft8f(o);
    v[0] = e[0] + o[0];
    v[8] = e[0] - o[0];
    bf_tw<false, true>(e[1], o[1], GR4PM_K2, v[1], v[9]);    // theta = pi/8
    bf_tw<false, true>(e[2], o[2], GR4PM_K4, v[2], v[10]);   // pi/4
    bf_tw<true, false>(e[3], o[3], GR4PM_K2, v[3], v[11]);   // 3 pi/8 = pi/2 - pi/8: g = -j, t = +tan(pi/8)
    v[4] = add_mj(e[4], o[4]);                               // -j
    v[12] = sub_mj(e[4], o[4]);
    bf_tw<true, true>(e[5], o[5], GR4PM_K2, v[5], v[13]);    // 5 pi/8 = pi/2 + pi/8: g = -j, t = -tan(pi/8)
    bf_tw<false, false>(e[6], o[6], GR4PM_K4, v[14], v[6]);  // 3 pi/4: g = -1, t = +1
    bf_tw<false, false>(e[7], o[7], GR4PM_K2, v[15], v[7]);  // 7 pi/8 = pi - pi/8: g = -1, t = +tan(pi/8)
}

// in-lane 32-point DFT, forward sign, natural order in and out; `done(k)` is called as soon as v[k] and
// v[k + 16] are final (the correlator issues their exchange stores there, between the butterflies of the
// last stage, instead of 64 stores in one burst)
struct NoDone {
    GR4PM_HD void operator()(int) const {}
};
template <typename Done = NoDone>
GR4PM_HD void dft32(cf* v, Done done = Done{})
{
    cf e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        e[i] = v[2 * i];
        o[i] = v[2 * i + 1];
    }
    dft16f(e);
    dft16f(o);
    // v[k], v[k + 16] = e[k] +- W32^k o[k], theta = pi k / 16
    v[0] = e[0] + o[0];
    v[16] = e[0] - o[0];
    done(0);
    bf_tw<false, true>(e[1], o[1], GR4PM_K1, v[1], v[17]);
    done(1);
    bf_tw<false, true>(e[2], o[2], GR4PM_K2, v[2], v[18]);
    done(2);
    bf_tw<false, true>(e[3], o[3], GR4PM_K3, v[3], v[19]);
    done(3);
    bf_tw<false, true>(e[4], o[4], GR4PM_K4, v[4], v[20]);
    done(4);
    bf_tw<true, false>(e[5], o[5], GR4PM_K3, v[5], v[21]);   // pi/2 - 3 pi/16
    done(5);
    bf_tw<true, false>(e[6], o[6], GR4PM_K2, v[6], v[22]);   // pi/2 - pi/8
    done(6);
    bf_tw<true, false>(e[7], o[7], GR4PM_K1, v[7], v[23]);   // pi/2 - pi/16
    done(7);
    v[8] = add_mj(e[8], o[8]);
    v[24] = sub_mj(e[8], o[8]);
    done(8);
    bf_tw<true, true>(e[9], o[9], GR4PM_K1, v[9], v[25]);     // pi/2 + pi/16
    done(9);
    bf_tw<true, true>(e[10], o[10], GR4PM_K2, v[10], v[26]);
    done(10);
    bf_tw<true, true>(e[11], o[11], GR4PM_K3, v[11], v[27]);
    done(11);
    bf_tw<false, false>(e[12], o[12], GR4PM_K4, v[28], v[12]); // 3 pi/4: g = -1
    done(12);
    bf_tw<false, false>(e[13], o[13], GR4PM_K3, v[29], v[13]); // pi - 3 pi/16
    done(13);
    bf_tw<false, false>(e[14], o[14], GR4PM_K2, v[30], v[14]);
    done(14);
    bf_tw<false, false>(e[15], o[15], GR4PM_K1, v[31], v[15]);
    done(15);
}

// planar in-lane DFT-32: b[i] = (v[2i], v[2i + 1]) on entry, b[k] = (V[k], V[k + 16]) on exit.  148 + 60 packed
// instructions (the interleaved form: 194) -- the 14 extra ones buy the mid stage without its 32 v_pk_mov_b32 and a
// power tail of 2 instead of 3 instructions per output.
GR4PM_HD void dft32p(pc* b)
{
    dft16f(b); // lo halves: DFT-16 of the even items, hi halves: DFT-16 of the odd items
    const cf one_m = mk(1.0f, -1.0f);
    {   // k = 0: (E + O, E - O)
        const pc x = b[0];
        b[0] = pc{ vfma(dup_y(x.r), one_m, dup_x(x.r)), vfma(dup_y(x.i), one_m, dup_x(x.i)) };
    }
    b[1] = bf_last<false, true, false>(b[1], GR4PM_K1);
    b[2] = bf_last<false, true, false>(b[2], GR4PM_K2);
    b[3] = bf_last<false, true, false>(b[3], GR4PM_K3);
    b[4] = bf_last<false, true, false>(b[4], GR4PM_K4);
    b[5] = bf_last<true, false, false>(b[5], GR4PM_K3);
    b[6] = bf_last<true, false, false>(b[6], GR4PM_K2);
    b[7] = bf_last<true, false, false>(b[7], GR4PM_K1);
    {   // k = 8: w = -j: (E_r + O_i, E_r - O_i), (E_i - O_r, E_i + O_r)
        const pc x = b[8];
        b[8] = pc{ vfma(dup_y(x.i), one_m, dup_x(x.r)), vfma(dup_y(x.r), -one_m, dup_x(x.i)) };
    }
    b[9] = bf_last<true, true, false>(b[9], GR4PM_K1);
    b[10] = bf_last<true, true, false>(b[10], GR4PM_K2);
    b[11] = bf_last<true, true, false>(b[11], GR4PM_K3);
    b[12] = bf_last<false, false, true>(b[12], GR4PM_K4);
    b[13] = bf_last<false, false, true>(b[13], GR4PM_K3);
    b[14] = bf_last<false, false, true>(b[14], GR4PM_K2);
    b[15] = bf_last<false, false, true>(b[15], GR4PM_K1);
}

// mid stage, planar out: the pair (m, m + 1) from the row reads as they are.  ur = a0r + c.x a1r - c.y a1i ;
// ui = a0i + c.x a1i + c.y a1r ; b = T u
GR4PM_HD pc w64_mid_pair_p(cf a0r, cf a1r, cf a0i, cf a1i, cf tr, cf ti, cf c)
{
    const cf cx = dup_x(c), cy = dup_y(c);
    const cf ur = vfma(a1i, -cy, vfma(a1r, cx, a0r));
    const cf ui = vfma(a1r, cy, vfma(a1i, cx, a0i));
    return pc{ vfma(ti, -ui, tr * ur), vfma(ti, ur, tr * ui) };
}
GR4PM_HD void w64_mid_group_p(const f4& r0, const f4& r1, const f4& i0, const f4& i1, const f4& tr, const f4& ti, cf c,
                              pc* b)
{
    b[0] = w64_mid_pair_p(mk(r0.x, r0.y), mk(r1.x, r1.y), mk(i0.x, i0.y), mk(i1.x, i1.y), mk(tr.x, tr.y), mk(ti.x, ti.y), c);
    b[1] = w64_mid_pair_p(mk(r0.z, r0.w), mk(r1.z, r1.w), mk(i0.z, i0.w), mk(i1.z, i1.w), mk(tr.z, tr.w), mk(ti.z, ti.w), c);
}
// host reference of the planar mid stage (the device version keeps its reads two groups ahead: correlate_w64.hpp)
GR4PM_HD void w64_mid_p(int lane, const float* xb, const f4* tT, cf c, pc* b)
{
    const f4* row = reinterpret_cast<const f4*>(xb + (lane & 31) * kW64Row);
#pragma unroll
    for (int g = 0; g < 8; ++g)
        w64_mid_group_p(row[g], row[8 + g], row[16 + g], row[24 + g], tT[(g * 2 + 0) * 64 + lane],
                        tT[(g * 2 + 1) * 64 + lane], c, b + 2 * g);
}

// exchange image of one wave (host emulation / reference of what the addtid stores do)
GR4PM_HD void w64_store_ref(int lane, const cf* r, float* xb)
{
    for (int k1 = 0; k1 < 32; ++k1) {
        xb[k1 * kW64Row + lane] = r[k1].x;
        xb[k1 * kW64Row + 64 + lane] = r[k1].y;
    }
}

// one group of four m: reads of row k (r0 = re[4g..], r1 = re[32 + 4g..], i0, i1 likewise), twiddles
GR4PM_HD void w64_mid_group(const f4& r0, const f4& r1, const f4& i0, const f4& i1, const f4& tr, const f4& ti, cf c,
                            cf* b)
{
    const float a0r[4] = { r0.x, r0.y, r0.z, r0.w }, a1r[4] = { r1.x, r1.y, r1.z, r1.w };
    const float a0i[4] = { i0.x, i0.y, i0.z, i0.w }, a1i[4] = { i1.x, i1.y, i1.z, i1.w };
    const float Tr[4] = { tr.x, tr.y, tr.z, tr.w }, Ti[4] = { ti.x, ti.y, ti.z, ti.w };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ur = a0r[e] + c.x * a1r[e] - c.y * a1i[e];
        const float ui = a0i[e] + c.x * a1i[e] + c.y * a1r[e];
        b[e] = mk(Tr[e] * ur - Ti[e] * ui, Tr[e] * ui + Ti[e] * ur);
    }
}

// mid stage of lane `lane` from the wave's exchange image xb and the twiddle table tT
GR4PM_HD void w64_mid(int lane, const float* xb, const f4* tT, cf c, cf* b)
{
    const f4* row = reinterpret_cast<const f4*>(xb + (lane & 31) * kW64Row);
#pragma unroll
    for (int g = 0; g < 8; ++g)
        w64_mid_group(row[g], row[8 + g], row[16 + g], row[24 + g], tT[(g * 2 + 0) * 64 + lane],
                      tT[(g * 2 + 1) * 64 + lane], c, b + 4 * g);
}

// ---- the same exchange with (re, im) interleaved: row k1 = 64 complex (512 B) + 16 B pad, written with
// ds_write_b64, read back two complex per ds_read_b128; twiddles as tC[i * 64 + lane] = (T[2i], T[2i + 1])
template <typename W>
inline void build_w64_tables_c(W w, f4* tC)
{
    for (int i = 0; i < 16; ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const cf t0 = w(((2 * i) * lane) % kFftN), t1 = w(((2 * i + 1) * lane) % kFftN);
            tC[i * 64 + lane] = mkf4(t0.x, t0.y, t1.x, t1.y);
        }
}
GR4PM_HD void w64c_store_ref(int lane, const cf* r, cf* xb)
{
    for (int k1 = 0; k1 < 32; ++k1) xb[k1 * (kW64Row / 2) + lane] = r[k1];
}
// b = T (a0 + c a1)
GR4PM_HD cf w64c_mid1(cf a0, cf a1, cf t, cf c) { return cmul(a0 + cmul(a1, c), t); }
GR4PM_HD void w64c_mid(int lane, const cf* xb, const f4* tC, cf c, cf* b)
{
    const f4* row = reinterpret_cast<const f4*>(xb + (lane & 31) * (kW64Row / 2));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f4 lo = row[i], hi = row[16 + i], t = tC[i * 64 + lane];
        b[2 * i] = w64c_mid1(mk(lo.x, lo.y), mk(hi.x, hi.y), mk(t.x, t.y), c);
        b[2 * i + 1] = w64c_mid1(mk(lo.z, lo.w), mk(hi.z, hi.w), mk(t.z, t.w), c);
    }
}

// index held in register j of lane `lane`, before and after the transform
GR4PM_HD int w64_index(int lane, int j) { return lane + 64 * j; }

} // namespace gr4pm
