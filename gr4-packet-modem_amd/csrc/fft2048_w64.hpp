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

// in-lane 32-point DFT, forward sign, natural order in and out; `done(k)` is called as soon as v[k] and
// v[k + 16] are final (the correlator issues their exchange stores there, between the butterflies of the
// last stage, instead of 64 stores in one burst)
struct NoDone {
    GR4PM_HD void operator()(int) const {}
};
template <typename Done = NoDone>
GR4PM_HD void dft32(cf* v, Done done = Done{})
{
    constexpr float c8 = 0.70710678118654752440f;
    // cos / sin of pi k / 16
    constexpr float c1 = 0.98078528040323044913f, s1 = 0.19509032201612826785f;
    constexpr float c2 = 0.92387953251128675613f, s2 = 0.38268343236508977173f;
    constexpr float c3 = 0.83146961230254523708f, s3 = 0.55557023301960222474f;
    cf e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        e[i] = v[2 * i];
        o[i] = v[2 * i + 1];
    }
    dft16(e);
    dft16(o);
    // o[k] *= W32^k = cos(pi k/16) - j sin(pi k/16)
    o[1] = cmulc(o[1], mk(c1, -s1));
    o[2] = cmulc(o[2], mk(c2, -s2));
    o[3] = cmulc(o[3], mk(c3, -s3));
    o[4] = c8 * add_mj(o[4], o[4]);
    o[5] = cmulc(o[5], mk(s3, -c3));
    o[6] = cmulc(o[6], mk(s2, -c2));
    o[7] = cmulc(o[7], mk(s1, -c1));
    // o[8] * (-j) is folded into the combination below
    o[9] = cmulc(o[9], mk(-s1, -c1));
    o[10] = cmulc(o[10], mk(-s2, -c2));
    o[11] = cmulc(o[11], mk(-s3, -c3));
    o[12] = (-c8) * sub_mj(o[12], o[12]);
    o[13] = cmulc(o[13], mk(-c3, -s3));
    o[14] = cmulc(o[14], mk(-c2, -s2));
    o[15] = cmulc(o[15], mk(-c1, -s1));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k == 8) {
            v[8] = add_mj(e[8], o[8]);
            v[24] = sub_mj(e[8], o[8]);
        } else {
            v[k] = e[k] + o[k];
            v[k + 16] = e[k] - o[k];
        }
        done(k);
    }
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
