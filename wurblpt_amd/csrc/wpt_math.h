/*
 * wpt_math.h -- the transcendental functions of the path, written once for the host (g++, x86-64: the test oracle) and
 * the device (hipcc, gfx950: the kernels), and giving on both THE BITS OF THIS IMAGE'S C LIBRARY.
 *
 * Why: the reference calls libm through gvm.hpp:118-146 (`using std::sin` ...).  One flipped last bit can send a path
 * down another branch ("Tolerance vs chaos"), so "the reference's result" is only defined together
 * with its libm -- here glibc 2.35 on x86-64 with FMA, which is what the reference's golden vectors
 * (tests/golden/ref_golden.json, written by oracle/ref_probe.cpp from the reference's own headers) were computed with.
 * Round 1 evaluated these functions in double and rounded once (correctly rounded results; glibc's differ from those in
 * 0.06 - 16 % of the inputs by one ulp), which left a hop between "GPU == restatement" and "restatement with libm ==
 * reference".  This header closes it: each function below is glibc's own algorithm --
 *
 *   sinf cosf expf powf   the double-precision table / polynomial algorithms glibc took over from Arm's optimized
 *                         routines (sysdeps/ieee754/flt-32/{s_sinf,s_cosf,e_expf,e_powf}.c; tables read from this image's
 *                         libm.so.6), with every multiply-add FUSED where the FMA build that glibc's ifunc selects on an
 *                         FMA machine fuses it (including both uses of x * InvLn2N in expf: found by the exhaustive test)
 *   asinf acosf           glibc's float forms (e_asinf.c: five-term polynomial; e_acosf.c: fdlibm's rational form)
 *   atanf atan2f          fdlibm's float forms (s_atanf.c with the 2^25 cut-off, e_atan2f.c); no FMA (these have no
 *                         ifunc variants and the baseline x86-64 build has none)
 *
 * -- and tests/test_math_exact.py checks, in the build container, sinf cosf expf asinf acosf atanf against the C
 * library for ALL 2^32 arguments and powf atan2f for 4e8 argument pairs plus the special values: 0 differences (12e9
 * pairs were run once, /tmp: 0).  asin_d, the double arc sine behind the measured-BRDF model's `2 * asin(x)`
 * (powitacq_rgb), stays the double evaluation of round 1: float(2 * asin_d(x)) equals float(2 * asin(x)) of the C library
 * for every float x in [-1, 1] (same test).  The device executes the same IEEE operations (v_fma_f64, correctly rounded
 * f32 divide / sqrt, no contraction of anything not written as fma_d), tests/test_gpu_parity.py compares it with the host.
 *
 * The algorithms are glibc's (LGPL 2.1+; the Arm routines MIT / Apache-2.0 WITH LLVM-exception; fdlibm: Copyright (C)
 * 1993 by Sun Microsystems, freely redistributable); what is here is a restatement from their published descriptions,
 * checked against the binary, with no file of theirs included.  See LICENSE.
 */
#ifndef WPT_MATH_H
#define WPT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define WPT_HD __host__ __device__ __forceinline__
#else
#define WPT_HD inline
#endif

namespace wptm {

WPT_HD double bits_to_double(uint64_t u)
{
    union { uint64_t u; double d; } c;
    c.u = u;
    return c.d;
}

WPT_HD uint64_t double_to_bits(double d)
{
    union { uint64_t u; double d; } c;
    c.d = d;
    return c.u;
}

WPT_HD uint32_t float_to_bits(float f)
{
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}

WPT_HD float bits_to_float(uint32_t u)
{
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

WPT_HD double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }

#if defined(__HIP_DEVICE_COMPILE__)
#define WPT_TABLE static __device__ const
#else
#define WPT_TABLE static const
#endif
/* The two tables of expf / powf (32 + 2 x 16 doubles, 512 bytes).  A kernel whose translation unit defines
 * WPT_MATH_TABLES_IN_LDS copies them to the first 512 bytes of its dynamic LDS (wptm::tables_to_lds) and the functions read
 * them there: a data-dependent look-up in global memory is a dependent memory round trip in the middle of a material
 * evaluation (measured on the 10 M triangle scene, which is bound by memory latency: 50 against 56 Msamples/s). */
#define WPT_MATH_TABLE_WORDS 64

/* ---- double arc tangent and arc sine (asin_d only, see above) ---- */
WPT_HD double atan_d(double x)
{
    const double atanhi0 = 4.63647609000806093515e-01;
    const double atanhi1 = 7.85398163397448278999e-01;
    const double atanhi2 = 9.82793723247329054082e-01;
    const double atanhi3 = 1.57079632679489655800e+00;
    const double atanlo0 = 2.26987774529616870924e-17;
    const double atanlo1 = 3.06161699786838301793e-17;
    const double atanlo2 = 1.39033110312309984516e-17;
    const double atanlo3 = 6.12323399573676603587e-17;
    const double aT0 = 3.33333333333329318027e-01;
    const double aT1 = -1.99999999998764832476e-01;
    const double aT2 = 1.42857142725034663711e-01;
    const double aT3 = -1.11111104054623557880e-01;
    const double aT4 = 9.09088713343650656196e-02;
    const double aT5 = -7.69187620504482999495e-02;
    const double aT6 = 6.66107313738753120669e-02;
    const double aT7 = -5.83357013379057348645e-02;
    const double aT8 = 4.97687799461593236017e-02;
    const double aT9 = -3.65315727442169155270e-02;
    const double aT10 = 1.62858201153657823623e-02;
    if (x != x)
        return x;
    bool neg = x < 0.0;
    double a = neg ? -x : x;
    if (a >= 7.378697629483821e19) { /* 2^66 */
        double z = atanhi3 + atanlo3;
        return neg ? -z : z;
    }
    int id;
    double t;
    if (a < 0.4375) {
        if (a < 7.450580596923828e-09) /* 2^-27 */
            return x;
        id = -1;
        t = a;
    } else if (a < 1.1875) {
        if (a < 0.6875) {
            id = 0;
            t = (2.0 * a - 1.0) / (2.0 + a);
        } else {
            id = 1;
            t = (a - 1.0) / (a + 1.0);
        }
    } else {
        if (a < 2.4375) {
            id = 2;
            t = (a - 1.5) / (1.0 + 1.5 * a);
        } else {
            id = 3;
            t = -1.0 / a;
        }
    }
    double z = t * t;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    double r;
    if (id < 0) {
        r = t - t * (s1 + s2);
    } else {
        double hi = id == 0 ? atanhi0 : id == 1 ? atanhi1 : id == 2 ? atanhi2 : atanhi3;
        double lo = id == 0 ? atanlo0 : id == 1 ? atanlo1 : id == 2 ? atanlo2 : atanlo3;
        r = hi - ((t * (s1 + s2) - lo) - t);
    }
    return neg ? -r : r;
}

WPT_HD double atan2_d(double y, double x)
{
    const double pi = 3.1415926535897931160e+00;
    const double pi_lo = 1.2246467991473531772e-16;
    const double pi_o_2 = 1.5707963267948965580e+00;
    const double pi_o_4 = 7.8539816339744827900e-01;
    if (x != x || y != y)
        return x + y;
    bool xneg = (double_to_bits(x) >> 63) != 0;
    bool yneg = (double_to_bits(y) >> 63) != 0;
    if (y == 0.0) {
        if (!xneg)
            return y; /* +-0 */
        return yneg ? -pi : pi;
    }
    if (x == 0.0)
        return yneg ? -pi_o_2 : pi_o_2;
    bool xinf = (x - x != 0.0);
    bool yinf = (y - y != 0.0);
    if (xinf) {
        if (yinf) {
            double r = xneg ? 3.0 * pi_o_4 : pi_o_4;
            return yneg ? -r : r;
        }
        double r = xneg ? pi : 0.0;
        return yneg ? -r : r;
    }
    if (yinf)
        return yneg ? -pi_o_2 : pi_o_2;
    double ay = yneg ? -y : y;
    double ax = xneg ? -x : x;
    double q = ay / ax;
    double z;
    if (q > 1.152921504606847e18) /* 2^60 */
        z = pi_o_2 + 0.5 * pi_lo;
    else if (xneg && q < 8.673617379884035e-19) /* 2^-60 */
        z = 0.0;
    else
        z = atan_d(q);
    if (!xneg)
        return yneg ? -z : z;
    double r = pi - (z - pi_lo);
    return yneg ? -r : r;
}


/* asin in double (for callers that round later); domain as asinf_ */
WPT_HD double asin_d(double xd)
{
    if (xd != xd)
        return xd;
    if (xd > 1.0 || xd < -1.0)
        return (double)bits_to_float(0x7fc00000u);
    double c = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
    return atan2_d(xd, c);
}


/* ---- glibc 2.35, x86-64, FMA builds ---- */
WPT_TABLE uint64_t exp2f_tab[32] = {
0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };

WPT_TABLE double powf_log2_tab[16][2] = {
    { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 }, { 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2 },
    { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 }, { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3 },
    { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 }, { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 },
    { 0x1.0000000000000p+0, 0x0.0p+0 }, { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3 },
    { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 }, { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },
    { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 } };

#if defined(__HIP_DEVICE_COMPILE__) && defined(WPT_MATH_TABLES_IN_LDS)
extern __shared__ uint64_t wpt_math_lds[]; /* the kernel's dynamic LDS; its first WPT_MATH_TABLE_WORDS words are the tables */
WPT_HD uint64_t exp2f_entry(uint64_t i) { return wpt_math_lds[i]; }
WPT_HD double log2_entry(int i, int k) { return bits_to_double(wpt_math_lds[32 + 2 * i + k]); }
/* called by every thread of the workgroup before the first use, followed by a barrier */
WPT_HD void tables_to_lds(unsigned int thread)
{
    if (thread < 32)
        wpt_math_lds[thread] = exp2f_tab[thread];
    else if (thread < 64)
        wpt_math_lds[thread] = double_to_bits(powf_log2_tab[(thread - 32) >> 1][thread & 1]);
}
#else
WPT_HD uint64_t exp2f_entry(uint64_t i) { return exp2f_tab[i]; }
WPT_HD double log2_entry(int i, int k) { return powf_log2_tab[i][k]; }
#endif

WPT_HD uint32_t top12(float x) { return float_to_bits(x) >> 20; }

WPT_HD float expf_(float x)
{
    const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    double xd = (double)x;
    uint32_t abstop = top12(x) & 0x7ff;
    if (abstop >= top12(88.0f)) {
        if (float_to_bits(x) == float_to_bits(-bits_to_float(0x7f800000u)))
            return 0.0f;
        if (abstop >= top12(bits_to_float(0x7f800000u)))
            return x + x;
        if (x > 0x1.62e42ep6f)
            return 0x1p97f * 0x1p97f; /* overflow */
        if (x < -0x1.9fe368p6f)
            return 0x1p-95f * 0x1p-95f; /* underflow */
    }
    double kd = fma_d(InvLn2N, xd, SHIFT);
    uint64_t ki = double_to_bits(kd);
    kd -= SHIFT;
    double r = fma_d(InvLn2N, xd, -kd);
    uint64_t t = exp2f_entry(ki % 32);
    t += ki << (52 - 5);
    double s = bits_to_double(t);
    double z = fma_d(C0, r, C1);
    double r2 = r * r;
    double y = fma_d(C2, r, 1.0);
    y = fma_d(z, r2, y);
    y = y * s;
    return (float)y;
}

/* sinf / cosf */
/* glibc keeps two sets of polynomial coefficients, the second with the cosine's negated (__sincosf_table[2]), and picks by
 * quadrant; rounding is symmetric in the sign, so negating the coefficients negates every intermediate and the result,
 * bit for bit.  Here the coefficients are literals and `negate` flips the cosine's result: no table, no loads. */
WPT_TABLE uint32_t inv_pio4[24] = { 0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27,
    0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43,
    0x993c4390, 0x3c439041 };

WPT_HD uint32_t abstop12(float x) { return (float_to_bits(x) >> 20) & 0x7ff; }

WPT_HD float sinf_poly(double x, double x2, bool negate, int n)
{
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double t1 = fma_d(x2, s3, s2);
        double x7 = x3 * x2;
        double t = fma_d(x3, s1, x);
        return (float)fma_d(x7, t1, t);
    } else {
        double x4 = x2 * x2;
        double t2 = fma_d(x2, c4, c3);
        double t1 = fma_d(x2, c1, c0);
        double x6 = x4 * x2;
        double t = fma_d(x4, c2, t1);
        const double r = fma_d(x6, t2, t);
        return (float)(negate ? -r : r);
    }
}
/* the sign of quadrant q: 1, -1, -1, 1 */
WPT_HD double quadrant_sign(int q) { return ((q ^ (q >> 1)) & 1) ? -1.0 : 1.0; }
WPT_HD double reduce_fast(double x, int* np)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma_d(-(double)n, hpi, x);
}
WPT_HD double reduce_large(uint32_t xi, int* np)
{
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = xi * arr[0];
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}
WPT_HD float sinf_(float y)
{
    double x = y, s;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f))
            return y;
        return sinf_poly(x, s, false, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        s = quadrant_sign(n & 3);
        return sinf_poly(x * s, x * x, (n & 2) != 0, n);
    } else if (abstop12(y) < abstop12(bits_to_float(0x7f800000u))) {
        uint32_t xi = float_to_bits(y);
        int sign = xi >> 31;
        x = reduce_large(xi, &n);
        s = quadrant_sign((n + sign) & 3);
        return sinf_poly(x * s, x * x, ((n + sign) & 2) != 0, n);
    }
    return (y - y) / (y - y);
}
WPT_HD float cosf_(float y)
{
    double x = y, s;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f))
            return 1.0f;
        return sinf_poly(x, x2, false, 1);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        s = quadrant_sign(n & 3);
        return sinf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    } else if (abstop12(y) < abstop12(bits_to_float(0x7f800000u))) {
        uint32_t xi = float_to_bits(y);
        int sign = xi >> 31;
        x = reduce_large(xi, &n);
        s = quadrant_sign((n + sign) & 3);
        return sinf_poly(x * s, x * x, ((n + sign) & 2) != 0, n ^ 1);
    }
    return (y - y) / (y - y);
}

/* powf */


WPT_HD double log2_inline(uint32_t ix)
{
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
    uint32_t tmp = ix - 0x3f330000;
    int i = (tmp >> (23 - 4)) % 16;
    uint32_t top = tmp & 0xff800000;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc = log2_entry(i, 0), logc = log2_entry(i, 1);
    double z = (double)bits_to_float(iz);
    double r = fma_d(z, invc, -1.0);
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = fma_d(A0, r, A1);
    double p = fma_d(A2, r, A3);
    double r4 = r2 * r2;
    double q = fma_d(A4, r, y0);
    q = fma_d(p, r2, q);
    y = fma_d(y, r4, q);
    return y;
}
WPT_HD float exp2_inline(double xd, uint32_t sign_bias)
{
    const double SHIFT = 0x1.8p+47, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double kd = xd + SHIFT;
    uint64_t ki = double_to_bits(kd);
    kd -= SHIFT;
    double r = xd - kd;
    uint64_t t = exp2f_entry(ki % 32);
    uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    double s = bits_to_double(t);
    double z = fma_d(C0, r, C1);
    double r2 = r * r;
    double y = fma_d(C2, r, 1.0);
    y = fma_d(z, r2, y);
    y = y * s;
    return (float)y;
}
WPT_HD int checkint(uint32_t iy)
{
    int e = iy >> 23 & 0xff;
    if (e < 0x7f)
        return 0;
    if (e > 0x7f + 23)
        return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1))
        return 0;
    if (iy & (1u << (0x7f + 23 - e)))
        return 1;
    return 2;
}
WPT_HD int zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }
WPT_HD float powf_(float x, float y)
{
    const uint32_t SIGN_BIAS = 1u << (5 + 11);
    uint32_t sign_bias = 0;
    uint32_t ix = float_to_bits(x), iy = float_to_bits(y);
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000 || zeroinfnan(iy)) {
        if (zeroinfnan(iy)) {
            if (2 * iy == 0)
                return 1.0f;
            if (ix == 0x3f800000)
                return 1.0f;
            if (2 * ix > 2u * 0x7f800000 || 2 * iy > 2u * 0x7f800000)
                return x + y;
            if (2 * ix == 2 * 0x3f800000)
                return 1.0f;
            if ((2 * ix < 2 * 0x3f800000) == !(iy & 0x80000000))
                return 0.0f;
            return y * y;
        }
        if (zeroinfnan(ix)) {
            float x2 = x * x;
            if (ix & 0x80000000 && checkint(iy) == 1) {
                x2 = -x2;
                sign_bias = 1;
            }
            if (2 * ix == 0 && iy & 0x80000000)
                return sign_bias ? -bits_to_float(0x7f800000u) : bits_to_float(0x7f800000u);
            return iy & 0x80000000 ? 1 / x2 : x2;
        }
        if (ix & 0x80000000) {
            int yint = checkint(iy);
            if (yint == 0)
                return (x - x) / (x - x);
            if (yint == 1)
                sign_bias = SIGN_BIAS;
            ix &= 0x7fffffff;
        }
        if (ix < 0x00800000) {
            ix = float_to_bits(x * 0x1p23f);
            ix &= 0x7fffffff;
            ix -= 23 << 23;
        }
    }
    double logx = log2_inline(ix);
    double ylogx = (double)y * logx;
    if ((double_to_bits(ylogx) >> 47 & 0xffff) >= double_to_bits(126.0) >> 47) {
        if (ylogx > 0x1.fffffffd1d571p+6)
            return sign_bias ? -(0x1p97f * 0x1p97f) : 0x1p97f * 0x1p97f;
        if (ylogx <= -150.0)
            return sign_bias ? -(0x1p-95f * 0x1p-95f) : 0x1p-95f * 0x1p-95f;
    }
    return exp2_inline(ylogx, sign_bias);
}

/* asinf, acosf: glibc's float forms */
WPT_HD float asinf_(float x)
{
    const float one = 1.0f, huge = 1.0e30f, pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f,
                pio4_hi = 0.785398185253143310546875f, p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f,
                p4 = 4.216630880e-2f;
    float t, w, p, q, c, r, s;
    int32_t hx = (int32_t)float_to_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000)
        return x * pio2_hi + x * pio2_lo;
    else if (ix > 0x3f800000)
        return (x - x) / (x - x);
    else if (ix < 0x3f000000) {
        if (ix < 0x32000000) {
            if (huge + x > one)
                return x;
        } else {
            t = x * x;
            w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
            return x + x * w;
        }
    }
    w = one - __builtin_fabsf(x);
    t = w * 0.5f;
    p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    s = __builtin_sqrtf(t);
    if (ix >= 0x3F79999A) {
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        int32_t iw;
        w = s;
        iw = (int32_t)float_to_bits(w);
        w = bits_to_float((uint32_t)iw & 0xfffff000);
        c = (t - w * w) / (s + w);
        r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    if (hx > 0)
        return t;
    else
        return -t;
}
WPT_HD float acosf_(float x)
{
    const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
                pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    float z, p, q, r, w, s, c, df;
    int32_t hx = (int32_t)float_to_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) {
        if (hx > 0)
            return 0.0f;
        else
            return pi + 2.0f * pio2_lo;
    } else if (ix > 0x3f800000) {
        return (x - x) / (x - x);
    }
    if (ix < 0x3f000000) {
        if (ix <= 0x23000000)
            return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = __builtin_sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {
        int32_t idf;
        z = (one - x) * 0.5f;
        s = __builtin_sqrtf(z);
        df = s;
        idf = (int32_t)float_to_bits(df);
        df = bits_to_float((uint32_t)idf & 0xfffff000);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0f * (df + w);
    }
}

/* atanf / atan2f: fdlibm float forms */
WPT_HD float atanf_(float x)
{
    const float atanhi[] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    const float atanlo[] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    const float aT[] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
        6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    const float one = 1.0f, huge = 1.0e30f;
    float w, s1, s2, z;
    int32_t ix, hx, id;
    hx = (int32_t)float_to_bits(x);
    ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000)
            return x + x;
        if (hx > 0)
            return atanhi[3] + atanlo[3];
        else
            return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x31000000) { /* |x| < 2^-29 */
            if (huge + x > one)
                return x;
        }
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) { /* |x| < 1.1875 */
            if (ix < 0x3f300000) { /* 7/16 <=|x|<11/16 */
                id = 0;
                x = (2.0f * x - one) / (2.0f + x);
            } else { /* 11/16<=|x|< 19/16 */
                id = 1;
                x = (x - one) / (x + one);
            }
        } else {
            if (ix < 0x401c0000) { /* |x| < 2.4375 */
                id = 2;
                x = (x - 1.5f) / (one + 1.5f * x);
            } else { /* 2.4375 <= |x| < 2^66 */
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0)
        return x - x * (s1 + s2);
    else {
        z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
        return (hx < 0) ? -z : z;
    }
}
WPT_HD float atan2f_(float y, float x)
{
    const float tiny = 1.0e-30f, zero = 0.0f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    int32_t k, m, hx, hy, ix, iy;
    hx = (int32_t)float_to_bits(x);
    ix = hx & 0x7fffffff;
    hy = (int32_t)float_to_bits(y);
    iy = hy & 0x7fffffff;
    if ((ix > 0x7f800000) || (iy > 0x7f800000))
        return x + y;
    if (hx == 0x3f800000)
        return atanf_(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
        case 0:
        case 1: return y;
        case 2: return pi + tiny;
        case 3: return -pi - tiny;
        }
    }
    if (ix == 0)
        return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            case 3: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return zero;
            case 1: return -zero;
            case 2: return pi + tiny;
            case 3: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000)
        return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = atanf_(__builtin_fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: {
        uint32_t zh = float_to_bits(z);
        return bits_to_float(zh ^ 0x80000000);
    }
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}


WPT_HD void sincosf_(float x, float* s, float* c)
{
    /* glibc's sincosf evaluates the same two polynomials on the same reduced argument as sinf and cosf */
    *s = sinf_(x);
    *c = cosf_(x);
}

} /* namespace wptm */

#endif
