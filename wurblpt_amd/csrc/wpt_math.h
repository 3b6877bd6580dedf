/*
 * wpt_math.h -- transcendental functions that give the SAME bits on the host
 * (g++, x86-64) and on the device (hipcc, gfx950).
 *
 * Why: the reference calls libm through gvm.hpp:118-146 (`using std::sin` ...).
 * glibc's and ROCm's float functions differ in the last bit for some inputs,
 * and one flipped bit can send a path down another branch (SURVEY section 7,
 * "Tolerance vs chaos").  The kernel therefore must not call either library.
 *
 * How: every function evaluates in double precision using only + - * / sqrt,
 * comparisons and bit moves -- operations that IEEE 754 defines exactly and that
 * both compilers emit unfused under -ffp-contract=off -- and rounds once to
 * float.  The double results are good to a few 1e-16, so the float result is the
 * correctly rounded one except when the exact value lies within ~1e-9 ulp of
 * a rounding boundary.  glibc's float functions are themselves within 0.5x ulp of
 * exact, so both agree except in rare last-bit cases; tests/test_math.py
 * measures the rate against libm.
 *
 * The polynomial kernels are the classic published minimax sets of Sun's fdlibm
 * (k_sin.c, k_cos.c, e_exp.c, e_log.c, s_atan.c; freely redistributable),
 * restated here; they are mathematical constants, not reference code.
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

WPT_HD bool is_nan(double x) { return x != x; }

/* round to nearest integer, ties to even: an exact IEEE operation on both sides
 * (roundsd / libm rint on x86-64, v_rndne_f64 on gfx950) */
WPT_HD double round_nearest(double x)
{
    return __builtin_rint(x);
}

/* sin and cos of r in [-pi/4, pi/4] (with a little slack) */
WPT_HD double kernel_sin(double r)
{
    const double S1 = -1.66666666666666324348e-01;
    const double S2 = 8.33333333332248946124e-03;
    const double S3 = -1.98412698298579493134e-04;
    const double S4 = 2.75573137070700676789e-06;
    const double S5 = -2.50507602534068634195e-08;
    const double S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double v = z * r;
    double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return r + v * (S1 + z * p);
}

WPT_HD double kernel_cos(double r)
{
    const double C1 = 4.16666666666666019037e-02;
    const double C2 = -1.38888888888741095749e-03;
    const double C3 = 2.48015872894767294178e-05;
    const double C4 = -2.75573143513906633035e-07;
    const double C5 = 2.08757232129817482790e-09;
    const double C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return 1.0 - (0.5 * z - z * p);
}

/* Argument reduction x = k*pi/2 + r.  Exact for |x| up to ~1e6 (two-part pi/2 with a
 * 33-bit head); beyond that it stays deterministic but loses accuracy, which is
 * irrelevant on this path (arguments are angles of a few pi). */
WPT_HD double reduce_pio2(double x, int* quadrant)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double pio2_1t = 6.07710050650619224932e-11; /* pi/2 - pio2_1 */
    double k = round_nearest(x * invpio2);
    double r = (x - k * pio2_1) - k * pio2_1t;
    *quadrant = (int)((long long)k & 3);
    return r;
}

WPT_HD void sincos_d(double x, double* s, double* c)
{
    if (!(x == x) || x - x != 0.0) { /* NaN or infinity */
        *s = x - x;
        *c = x - x;
        return;
    }
    int q;
    double r = reduce_pio2(x, &q);
    double sr = kernel_sin(r);
    double cr = kernel_cos(r);
    switch (q) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

WPT_HD float sinf_(float x)
{
    double s, c;
    sincos_d((double)x, &s, &c);
    return (float)s;
}

WPT_HD float cosf_(float x)
{
    double s, c;
    sincos_d((double)x, &s, &c);
    return (float)c;
}

WPT_HD void sincosf_(float x, float* s, float* c)
{
    double sd, cd;
    sincos_d((double)x, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
}

/* exp for arguments whose result fits the double range comfortably (|x| < 700) */
WPT_HD double exp_d(double x)
{
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (x != x)
        return x;
    if (x > 700.0)
        return bits_to_double(0x7ff0000000000000ull); /* +inf */
    if (x < -700.0)
        return 0.0;
    double k = round_nearest(x * invln2);
    double hi = x - k * ln2HI;
    double lo = k * ln2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    long long ki = (long long)k;
    double scale = bits_to_double((uint64_t)(ki + 1023) << 52); /* 2^k, |k| <= 1010 */
    return y * scale;
}

/* natural log of a positive, finite, normal double */
WPT_HD double log_pos_d(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01;
    const double Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01;
    const double Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01;
    const double Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    uint64_t u = double_to_bits(x);
    int k = (int)(u >> 52) - 1023;
    uint64_t mant = u & 0x000fffffffffffffull;
    /* choose m in [sqrt(1/2), sqrt(2)) */
    if (mant >= 0x6a09e667f3bcdull) { /* mantissa of sqrt(2) */
        k += 1;
        u = mant | 0x3fe0000000000000ull; /* m in [sqrt(2)/2, 1) */
    } else {
        u = mant | 0x3ff0000000000000ull; /* m in [1, sqrt(2)) */
    }
    double f = bits_to_double(u) - 1.0;
    double dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

WPT_HD float expf_(float x)
{
    return (float)exp_d((double)x);
}

WPT_HD float logf_(float x)
{
    if (x != x)
        return x;
    if (x < 0.0f)
        return bits_to_float(0x7fc00000u);
    if (x == 0.0f)
        return bits_to_float(0xff800000u); /* -inf */
    if (x - x != 0.0f)
        return x; /* +inf */
    return (float)log_pos_d((double)x);
}

/* pow with the C99 special cases that can occur for finite float arguments */
WPT_HD float powf_(float x, float y)
{
    if (y == 0.0f || x == 1.0f)
        return 1.0f;
    if (x != x || y != y)
        return x + y;
    const float inf = bits_to_float(0x7f800000u);
    float ax = x < 0.0f ? -x : x;
    float ay = y < 0.0f ? -y : y;
    /* classify y: 0 = not an integer, 1 = odd integer, 2 = even integer */
    int yint = 0;
    if (ay >= 16777216.0f) {
        yint = 2;
    } else {
        float fl = (float)(long long)ay; /* ay < 2^24: truncation is exact */
        if (fl == ay)
            yint = (((long long)ay) & 1) ? 1 : 2;
    }
    bool xneg = (float_to_bits(x) >> 31) != 0;
    if (ay == inf) {
        if (ax == 1.0f)
            return 1.0f;
        return ((ax > 1.0f) == (y > 0.0f)) ? inf : 0.0f;
    }
    if (ax == 0.0f || ax == inf) {
        float r = ((ax == 0.0f) == (y > 0.0f)) ? 0.0f : inf;
        return (xneg && yint == 1) ? -r : r;
    }
    if (xneg && yint == 0)
        return bits_to_float(0x7fc00000u);
    double l = log_pos_d((double)ax);
    double r = exp_d((double)y * l);
    if (xneg && yint == 1)
        r = -r;
    return (float)r;
}

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

WPT_HD float atanf_(float x)
{
    return (float)atan_d((double)x);
}

WPT_HD float atan2f_(float y, float x)
{
    return (float)atan2_d((double)y, (double)x);
}

WPT_HD float asinf_(float x)
{
    if (x != x)
        return x;
    double xd = (double)x;
    if (xd > 1.0 || xd < -1.0)
        return bits_to_float(0x7fc00000u);
    double c = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
    return (float)atan2_d(xd, c);
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

WPT_HD float acosf_(float x)
{
    if (x != x)
        return x;
    double xd = (double)x;
    if (xd > 1.0 || xd < -1.0)
        return bits_to_float(0x7fc00000u);
    double s = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
    return (float)atan2_d(s, xd);
}

} /* namespace wptm */

#endif
