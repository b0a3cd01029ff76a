// glibc >= 2.28 sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c:
// double range reduction + double polynomial, one rounding) restated; exhaustive comparison with the host libm.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#ifndef FMA
#define FMA 1
#endif
#if FMA
#define MADD(a, b, c) fma((a), (b), (c))
#else
#define MADD(a, b, c) ((a) * (b) + (c))
#endif
static const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
static const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                    C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                    S3 = -0x1.994eb3774cf24p-13;
// n even: sine polynomial of x; n odd: cosine polynomial; `neg` = second table (cosine coefficients negated)
static inline float poly(double x, double x2, int n, int neg)
{
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = MADD(x2, S3, S2);
        double x7 = x3 * x2;
        double s = MADD(x3, S1, x);
        return (float)MADD(x7, s1, s);
    } else {
        double sg = neg ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = MADD(x2, sg * C4, sg * C3);
        double c1 = MADD(x2, sg * C1, sg * C0);
        double x6 = x4 * x2;
        double c = MADD(x4, sg * C2, c1);
        return (float)MADD(x6, c2, c);
    }
}
static inline uint32_t abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
static void my_sincosf(float y, float* sp, float* cp)
{
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { // pi/4
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) { *sp = y; *cp = 1.0f; return; }
        *sp = poly(x, x2, 0, 0);
        *cp = poly(x, x2, 1, 0);
        return;
    }
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = MADD(-(double)n, hpi, x);
    static const double sign[4] = { 1.0, -1.0, -1.0, 1.0 };
    double s = sign[n & 3];
    int neg = (n & 2) != 0;
    *sp = poly(x * s, x * x, n, neg);
    *cp = poly(x * s, x * x, n ^ 1, neg);
}
typedef struct { uint32_t lo, hi; long bad_s, bad_c; uint32_t first_bad; } job;
static void* run(void* a)
{
    job* j = (job*)a;
    for (uint32_t u = j->lo; u < j->hi; ++u)
        for (int sgn = 0; sgn < 2; ++sgn) {
            uint32_t v = u | ((uint32_t)sgn << 31);
            float y; memcpy(&y, &v, 4);
            float s, c;
            my_sincosf(y, &s, &c);
            float rs = sinf(y), rc = cosf(y);
            if (memcmp(&s, &rs, 4)) { if (!j->bad_s && !j->bad_c) j->first_bad = v; j->bad_s++; }
            if (memcmp(&c, &rc, 4)) { if (!j->bad_s && !j->bad_c) j->first_bad = v; j->bad_c++; }
        }
    return 0;
}
int main()
{
    float lim = 3.2f; uint32_t top; memcpy(&top, &lim, 4);
    enum { T = 8 };
    pthread_t th[T]; job jb[T];
    for (int i = 0; i < T; ++i) {
        jb[i] = (job){ (uint32_t)((uint64_t)top * i / T), (uint32_t)((uint64_t)top * (i + 1) / T), 0, 0, 0 };
        pthread_create(&th[i], 0, run, &jb[i]);
    }
    long bs = 0, bc = 0;
    for (int i = 0; i < T; ++i) { pthread_join(th[i], 0); bs += jb[i].bad_s; bc += jb[i].bad_c; if (jb[i].first_bad) printf("first bad 0x%08x\n", jb[i].first_bad); }
    printf("FMA=%d: %u floats x 2 signs: sin mismatches %ld, cos mismatches %ld\n", FMA, top, bs, bc);
    return 0;
}
