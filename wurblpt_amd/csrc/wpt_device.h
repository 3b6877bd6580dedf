/*
 * wpt_device.h -- device-side building blocks of the gfx950 path tracer.
 *
 * Everything here is scalar-per-lane code: one lane owns one pixel and walks that pixel's
 * sample sequence, because the reference seeds one Prng per pixel and consumes it serially
 * over all samples (reference wurblpt.hpp:342-366).  The arithmetic follows the reference
 * operation by operation (each function cites its source), is compiled with
 * -ffp-contract=off and IEEE division / square root, and calls no math library: the
 * transcendentals are the double-evaluated ones of wpt_math.h.  Together that makes the
 * frame a pure function of the scene bytes, identical on the host and on the device.
 *
 * Layout notes for MI355X: the BVH node is 32 B (two dwordx4 loads), the intersection
 * stream 48 B per triangle (three dwordx4 loads), the shading stream 96 B per triangle and is
 * touched once per ray, after traversal (the reference builds the full HitRecord for every
 * accepted candidate, hitable_triangle.hpp:277-324; deferring it to the final candidate does
 * not change any value).  Traversal needs no stack: every node carries the index of the first
 * node behind its subtree (wpt_pathtrace.inc.h).
 */
#ifndef WPT_DEVICE_H
#define WPT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wurblpt_hip.h"
#include "wpt_math.h"
#include "wpt_anim.h"
#include "wpt_rgl.h"

namespace wptd {

#define WPT_D __device__ __forceinline__
/* code that few lanes run and that is large: a real call, so that it does not take registers from the common path */
#define WPT_CALL static inline __device__ __attribute__((noinline))

constexpr float k_pi = 3.1415926535897932384626433832795029L;
constexpr float k_pi_2 = 1.5707963267948966192313216916397514L;
constexpr float k_pi_4 = 0.7853981633974483096156608458198757L;
constexpr float k_inv_pi = 0.3183098861837906715377675267450287L;
constexpr float k_maxval = 3.402823466e+38f;
constexpr float k_epsilon = 1.1920928955078125e-07f;
constexpr float k_ldeps = 1.084202172485504434e-19f; /* float(epsilon of long double), hitable_triangle.hpp:240 */

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

/* comparison-based min / max, NaN behaviour of gvm.hpp:88,93 */
WPT_D float fminr(float x, float y) { return x < y ? x : y; }
WPT_D float fmaxr(float x, float y) { return x > y ? x : y; }
WPT_D float clampr(float x, float lo, float hi) { return fminr(hi, fmaxr(lo, x)); }
WPT_D float mixr(float x, float y, float a) { return x + a * (y - x); }

WPT_D f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
WPT_D f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
WPT_D f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
WPT_D f4 ld4(const float* p) { return mk4(p[0], p[1], p[2], p[3]); }
WPT_D f3 add(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
WPT_D f3 sub(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
WPT_D f3 mul(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
WPT_D f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
WPT_D f3 scl(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }    /* s * v */
WPT_D f3 sclr(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }   /* v * s */
WPT_D f3 divs(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
WPT_D f4 add(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
WPT_D f4 sub(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
WPT_D f4 mul(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
WPT_D f4 scl(float s, f4 a) { return mk4(s * a.x, s * a.y, s * a.z, s * a.w); }
WPT_D f4 sclr(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
WPT_D f4 divs(f4 a, float s) { return mk4(a.x / s, a.y / s, a.z / s, a.w / s); }
WPT_D float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
WPT_D float comp(f4 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w)); }

/* dot products accumulate from zero like gvm.hpp:1183-1189 (keeps -0 behaviour) */
WPT_D float dot(f2 a, f2 b) { float d = 0.0f; d += a.x * b.x; d += a.y * b.y; return d; }
WPT_D float dot(f3 a, f3 b) { float d = 0.0f; d += a.x * b.x; d += a.y * b.y; d += a.z * b.z; return d; }
WPT_D f3 normalize(f3 v) { return divs(v, __builtin_sqrtf(dot(v, v))); }
WPT_D f3 cross(f3 v, f3 w) { return mk3(v.y * w.z - v.z * w.y, v.z * w.x - v.x * w.z, v.x * w.y - v.y * w.x); }
WPT_D f3 reflect(f3 i, f3 n) { return sub(i, scl(2.0f * dot(n, i), n)); } /* gvm.hpp:1213 */
WPT_D f3 refract(f3 i, f3 n, float eta)                                  /* gvm.hpp:1218 */
{
    const float d = dot(n, i);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k <= 0.0f)
        return mk3(0.0f, 0.0f, 0.0f);
    return sub(sclr(i, eta), sclr(n, eta * d + __builtin_sqrtf(k)));
}
WPT_D float max4(f4 a)
{
    float r = a.x;
    if (a.y > r) r = a.y;
    if (a.z > r) r = a.z;
    if (a.w > r) r = a.w;
    return r;
}
WPT_D float average3(f3 a)
{
    const float inv_N = 1.0f / 3.0f;
    float sum = 0.0f;
    sum += a.x; sum += a.y; sum += a.z;
    return inv_N * sum;
}

/* column-major 3x3 times vector, accumulation order of gvm.hpp:1481-1491 */
WPT_D f3 mat3mul(const float* m, f3 w)
{
    f3 r;
    r.x = 0.0f; r.x += m[0] * w.x; r.x += m[3] * w.y; r.x += m[6] * w.z;
    r.y = 0.0f; r.y += m[1] * w.x; r.y += m[4] * w.y; r.y += m[7] * w.z;
    r.z = 0.0f; r.z += m[2] * w.x; r.z += m[5] * w.y; r.z += m[8] * w.z;
    return r;
}
WPT_D f3 mat4mulPoint(const float* m, f3 p)
{
    f3 r;
    r.x = 0.0f; r.x += m[0] * p.x; r.x += m[4] * p.y; r.x += m[8] * p.z; r.x += m[12] * 1.0f;
    r.y = 0.0f; r.y += m[1] * p.x; r.y += m[5] * p.y; r.y += m[9] * p.z; r.y += m[13] * 1.0f;
    r.z = 0.0f; r.z += m[2] * p.x; r.z += m[6] * p.y; r.z += m[10] * p.z; r.z += m[14] * 1.0f;
    return r;
}
/* quaternion (x, y, z, w) rotates v, gvm.hpp:1713-1720 */
WPT_D f3 quatRotate(const float* q, f3 v)
{
    f3 s = mk3(q[0], q[1], q[2]);
    f3 t = scl(2.0f, cross(s, v));
    return add(add(v, scl(q[3], t)), cross(s, t));
}

/* ---- Prng: xoshiro128+ seeded by splitmix64 (prng.hpp:47-101) ---- */
struct Prng {
    uint32_t s0, s1, s2, s3;
};
WPT_D uint64_t splitmix64(uint64_t x)
{
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
WPT_D void prngSeed(Prng& p, uint32_t pixelIndex)
{
    uint64_t seed = (uint64_t)pixelIndex + 42ull;
    uint64_t s01 = splitmix64(seed);
    uint64_t s23 = splitmix64(s01);
    p.s0 = (uint32_t)(s01 >> 32);
    p.s1 = (uint32_t)(s01 & 0xffffffffull);
    p.s2 = (uint32_t)(s23 >> 32);
    p.s3 = (uint32_t)(s23 & 0xffffffffull);
}
WPT_D float in01(Prng& p)
{
    const uint32_t result = p.s0 + p.s3;
    const uint32_t t = p.s1 << 9;
    p.s2 ^= p.s0;
    p.s3 ^= p.s1;
    p.s1 ^= p.s2;
    p.s0 ^= p.s3;
    p.s2 ^= t;
    p.s3 = (p.s3 << 11) | (p.s3 >> 21);
    return (float)(result >> 8) * 5.9604644775390625e-08f; /* exact: 24-bit integer times 2^-24 */
}
/* prng.hpp:97-100 as compiled by the reference compiler: .y is drawn first */
WPT_D f2 in01x2(Prng& p)
{
    f2 r;
    r.y = in01(p);
    r.x = in01(p);
    return r;
}

/* ---- Sampler (sampler.hpp:39-109) ---- */
WPT_D f2 inUnitDisk(f2 u)
{
    float ox = 2.0f * u.x - 1.0f;
    float oy = 2.0f * u.y - 1.0f;
    f2 r;
    if (ox == 0.0f && oy == 0.0f) {
        r.x = 0.0f;
        r.y = 0.0f;
    } else {
        float theta, rad;
        if (__builtin_fabsf(ox) > __builtin_fabsf(oy)) {
            rad = ox;
            theta = k_pi_4 * (oy / ox);
        } else {
            rad = oy;
            theta = k_pi_2 - k_pi_4 * (ox / oy);
        }
        float s, c;
        wptm::sincosf_(theta, &s, &c);
        r.x = rad * c;
        r.y = rad * s;
    }
    return r;
}
WPT_D f3 inTriangle(f2 u)
{
    float su0 = __builtin_sqrtf(u.x);
    float b0 = 1.0f - su0;
    float b1 = u.y * su0;
    return mk3(b0, b1, 1.0f - b0 - b1);
}
WPT_D f3 cosineDirection(f2 u)
{
    f2 d = inUnitDisk(u);
    float z = __builtin_sqrtf(fmaxr(0.0f, 1.0f - dot(d, d)));
    return mk3(d.x, d.y, z);
}

/* ---- TangentSpace (tangentspace.hpp:46-136) ---- */
struct Frame {
    f3 n, t, b;
};
WPT_D Frame frameFromNormal(f3 n) /* Duff et al., tangentspace.hpp:57-74 */
{
    Frame f;
    f.n = n;
    float sign = __builtin_copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    f.t = mk3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    f.b = mk3(b, sign + n.y * n.y * a, -n.y);
    return f;
}
WPT_D Frame frameFromNormal(f3 n);
WPT_D f3 toWorld(const Frame& f, f3 v);
/* Sampler::onUnitSphere (sampler.hpp:69-77) */
WPT_D f3 onUnitSphere(f2 u)
{
    const float z = 1.0f - 2.0f * u.x;
    const float r = __builtin_sqrtf(fmaxr(0.0f, 1.0f - z * z));
    const float phi = 2.0f * k_pi * u.y;
    return mk3(r * wptm::cosf_(phi), r * wptm::sinf_(phi), z);
}
/* Sampler::toSphere (sampler.hpp:112-120) */
WPT_D f3 toSphere(f3 direction, float cosThetaMax, f2 u)
{
    const float cosTheta = (1.0f - u.x) + u.x * cosThetaMax;
    const float sinTheta = __builtin_sqrtf(fmaxr(0.0f, 1.0f - cosTheta * cosTheta));
    const float phi = u.y * 2.0f * k_pi;
    const f3 vectorAroundZ = mk3(wptm::cosf_(phi) * sinTheta, wptm::sinf_(phi) * sinTheta, cosTheta);
    return normalize(toWorld(frameFromNormal(direction), vectorAroundZ));
}

WPT_D Frame frameFromNT(f3 n, f3 t)
{
    Frame f;
    f.n = n;
    f.t = t;
    f.b = cross(n, t);
    return f;
}
WPT_D f3 toTangent(const Frame& f, f3 v) /* rows are t, b, n */
{
    f3 r;
    r.x = 0.0f; r.x += f.t.x * v.x; r.x += f.t.y * v.y; r.x += f.t.z * v.z;
    r.y = 0.0f; r.y += f.b.x * v.x; r.y += f.b.y * v.y; r.y += f.b.z * v.z;
    r.z = 0.0f; r.z += f.n.x * v.x; r.z += f.n.y * v.y; r.z += f.n.z * v.z;
    return r;
}
WPT_D f3 toWorld(const Frame& f, f3 v) /* columns are t, b, n */
{
    f3 r;
    r.x = 0.0f; r.x += f.t.x * v.x; r.x += f.b.x * v.y; r.x += f.n.x * v.z;
    r.y = 0.0f; r.y += f.t.y * v.x; r.y += f.b.y * v.y; r.y += f.n.y * v.z;
    r.z = 0.0f; r.z += f.t.z * v.x; r.z += f.b.z * v.y; r.z += f.n.z * v.z;
    return r;
}

/* ---- rays and hit records ---- */
struct Ray {
    f3 o, d;
    f4 ri; /* refractiveIndex */
};
/* RayIntersectionHelper (hitable.hpp:66-113) */
/* Kept per ray in registers while it traverses, so it is small: the axis permutation is
 * packed into one word (kx | ky << 2 | kz << 4) and S.z, which equals inv[kz], is not stored. */
struct RayAux {
    f3 inv;
    int k;
    float Sx, Sy;
};
WPT_D int auxKx(const RayAux& h) { return h.k & 3; }
WPT_D int auxKy(const RayAux& h) { return (h.k >> 2) & 3; }
WPT_D int auxKz(const RayAux& h) { return (h.k >> 4) & 3; }
/* SHEAR_ONLY: for triangle tests alone (the pdf of a light), which read the reciprocal of the direction's largest
 * component and nothing else of `inv`: that one division instead of three, the same bits */
template<bool SHEAR_ONLY = false> WPT_D RayAux rayAux(f3 dir)
{
    RayAux h;
    if (!SHEAR_ONLY)
        h.inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    float ax = __builtin_fabsf(dir.x), ay = __builtin_fabsf(dir.y), az = __builtin_fabsf(dir.z);
    int kx, ky, kz;
    if (az >= ay && az >= ax)
        kz = 2;
    else if (ay >= ax)
        kz = 1;
    else
        kz = 0;
    kx = kz + 1;
    if (kx == 3)
        kx = 0;
    ky = kx + 1;
    if (ky == 3)
        ky = 0;
    if (comp(dir, kz) < 0.0f) {
        int tmp = kx;
        kx = ky;
        ky = tmp;
    }
    float invz;
    if (SHEAR_ONLY) {
        invz = 1.0f / comp(dir, kz);
        h.inv = mk3(invz, invz, invz);
    } else {
        invz = comp(h.inv, kz);
    }
    h.Sx = comp(dir, kx) * invz;
    h.Sy = comp(dir, ky) * invz;
    h.k = kx | (ky << 2) | (kz << 4);
    return h;
}

/* what survives of a triangle candidate: enough to rebuild the HitRecord later */
struct Candidate {
    uint32_t prim; /* 0xffffffff = no hit */
    float a, invDet, U, V, W; /* det itself is not kept: its sign is the sign of invDet */
};

/* Watertight test (hitable_triangle.hpp:189-271).  Returns true and fills c when accepted. */
WPT_D bool triangleTest(f3 v0, f3 v1, f3 v2, f3 org, const RayAux& h, float amin, float amax, Candidate& c)
{
    const f3 A = sub(v0, org);
    const f3 B = sub(v1, org);
    const f3 C = sub(v2, org);
    const int kx = auxKx(h), ky = auxKy(h), kz = auxKz(h);
    const float Sz = comp(h.inv, kz);
    const float Akz = comp(A, kz), Bkz = comp(B, kz), Ckz = comp(C, kz);
    const float Ax = comp(A, kx) - h.Sx * Akz;
    const float Ay = comp(A, ky) - h.Sy * Akz;
    const float Bx = comp(B, kx) - h.Sx * Bkz;
    const float By = comp(B, ky) - h.Sy * Bkz;
    const float Cx = comp(C, kx) - h.Sx * Ckz;
    const float Cy = comp(C, ky) - h.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (__builtin_fabsf(U) < k_ldeps || __builtin_fabsf(V) < k_ldeps || __builtin_fabsf(W) < k_ldeps) {
        double CxBy = (double)Cx * (double)By;
        double CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy;
        double AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay;
        double ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    float det = U + V + W;
    if (det == 0.0f)
        return false;
    const float Az = Sz * Akz;
    const float Bz = Sz * Bkz;
    const float Cz = Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const uint32_t sgn = wptm::float_to_bits(det) & 0x80000000u;
    const float Ts = wptm::bits_to_float(wptm::float_to_bits(T) ^ sgn);
    const float ds = wptm::bits_to_float(wptm::float_to_bits(det) ^ sgn);
    if (Ts < amin * ds || Ts > amax * ds)
        return false;
    const float invDet = 1.0f / det;
    c.a = invDet * T;
    c.invDet = invDet;
    c.U = U;
    c.V = V;
    c.W = W;
    return true;
}

/* AABB::mayHit (aabb.hpp:70-86) written out with the reference's comparison chains, whose
 * results for NaN slab distances (0 * inf: origin on a slab plane, direction parallel to it)
 * depend on the operand order */
WPT_D bool boxTestChains(float t0x, float t0y, float t0z, float t1x, float t1y, float t1z, float amin, float amax)
{
    float mnx = fminr(t0x, t1x), mny = fminr(t0y, t1y), mnz = fminr(t0z, t1z);
    float mxx = fmaxr(t0x, t1x), mxy = fmaxr(t0y, t1y), mxz = fmaxr(t0z, t1z);
    float tmin = amin;
    if (mnx > tmin) tmin = mnx;
    if (mny > tmin) tmin = mny;
    if (mnz > tmin) tmin = mnz;
    float tmax = amax;
    if (mxx < tmax) tmax = mxx;
    if (mxy < tmax) tmax = mxy;
    if (mxz < tmax) tmax = mxz;
    return tmin <= tmax;
}

/* AABB::mayHit.  When none of the six slab distances is NaN the comparison chains are plain
 * minima and maxima (the sign of a zero cannot change the final comparison), which the hardware
 * has as single instructions (v_min_f32 / v_max3_f32) where a compare + select pair costs three
 * issue slots; the rare lanes with a NaN take the chains as written.
 * CHECK = false leaves the test for NaN out: for rays whose slab distances are numbers whatever the box (rayMayNan below). */
typedef float v2f __attribute__((ext_vector_type(2)));
template<bool CHECK = true> WPT_D bool boxTest(f3 lo, f3 hi, f3 org, f3 inv, float amin, float amax)
{
    /* the six distances as three pairs, the way the node record holds the bounds (nodeLo / nodeHi): one packed subtract and
     * one packed multiply per pair, each element the same IEEE operation as (bound - origin) * reciprocal */
    const v2f tx = (v2f { lo.x, hi.x } - v2f { org.x, org.x }) * v2f { inv.x, inv.x };
    const v2f t0yz = (v2f { lo.y, lo.z } - v2f { org.y, org.z }) * v2f { inv.y, inv.z };
    const v2f t1yz = (v2f { hi.y, hi.z } - v2f { org.y, org.z }) * v2f { inv.y, inv.z };
    const float t0x = tx.x, t1x = tx.y, t0y = t0yz.x, t0z = t0yz.y, t1y = t1yz.x, t1z = t1yz.y;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
            __builtin_fmaxf(__builtin_fminf(t0z, t1z), amin));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
            __builtin_fminf(__builtin_fmaxf(t0z, t1z), amax));
    bool hit = tmin <= tmax;
    if (CHECK && __builtin_expect(__builtin_isunordered(t0x, t1x) || __builtin_isunordered(t0y, t1y) || __builtin_isunordered(t0z, t1z), 0))
        hit = boxTestChains(t0x, t0y, t0z, t1x, t1y, t1z, amin, amax);
    return hit;
}

/* Can a slab distance (box - origin) * (1 / direction) of this ray be NaN for a box whose coordinates are numbers?  Only as
 * 0 * inf or inf - inf: a reciprocal that is zero, infinite or NaN (a direction component that is infinite, zero -- also a
 * denormal whose reciprocal overflows -- or NaN), or an origin that is not finite.  For every other ray the six distances are
 * numbers (possibly infinite), and the walks skip the test for NaN per box.  (A box coordinate that is NaN itself: SceneView::
 * boxesMayBeNan, found at upload, keeps the test for every ray.) */
constexpr int RAY_MAY_NAN = 0x40; /* in RayAux::k, above the three axes */
WPT_D bool rayMayNan(f3 org, f3 inv)
{
    const float inf = __builtin_inff();
    const bool numbers = __builtin_fabsf(inv.x) < inf && __builtin_fabsf(inv.y) < inf && __builtin_fabsf(inv.z) < inf
            && inv.x != 0.0f && inv.y != 0.0f && inv.z != 0.0f
            && __builtin_fabsf(org.x) < inf && __builtin_fabsf(org.y) < inf && __builtin_fabsf(org.z) < inf;
    return !numbers;
}

/* the device's node record, two quadwords: lo.x hi.x lo.y lo.z | hi.y hi.z skip word -- the x bounds side by side, the y and z
 * bounds as pairs, so that the packed subtract / multiply of the box test take them as they come (three register moves per
 * node less than lo.xyz hi.x | hi.yz) */
WPT_D f3 nodeLo(float4 n0, float4 n1) { (void)n1; return mk3(n0.x, n0.z, n0.w); }
WPT_D f3 nodeHi(float4 n0, float4 n1) { return mk3(n0.y, n1.x, n1.y); }

/* feature bits: what a kernel instantiation can evaluate */
enum {
    FEAT_TEXTURES = 1,   /* any texture, normal maps */
    FEAT_MODPHONG = 2,   /* MaterialModPhong */
    FEAT_ENVMAP = 4,     /* environment map radiance and importance sampling */
    FEAT_LENS = 8,       /* thin lens camera */
    FEAT_TWOSIDED = 16,  /* MaterialTwoSided */
    FEAT_GGX = 32,       /* MaterialGGX */
    FEAT_GLASS = 64,     /* MaterialGlass, MaterialMirror */
    FEAT_SPHERES = 128,  /* HitableSphere leaves and sphere hot spots */
    FEAT_RGL = 256,      /* MaterialRGL, measured BRDFs (wpt_rgl.h) */
    FEAT_ANIM = 512      /* exposure interval t0 != t1 and / or animated instances (wpt_anim.h) */
};

/* a primitive index with this bit is a sphere (index in the low bits), otherwise a triangle */
constexpr uint32_t PRIM_SPHERE = 0x80000000u;
constexpr float k_cosOrthoAngleTolerance = 0.0003f; /* constants.hpp:41 */

/* full HitRecord (hitable.hpp:39-64) */
struct Hit {
    float a;
    f3 p, n, t;
    f2 tc;
    bool backside;
    uint32_t prim;
    uint32_t material;
};

struct SceneView {
    const float4* nodes;     /* 2 x float4 per node: lo.x hi.x lo.y lo.z | hi.y hi.z skip word (device form, nodeLo / nodeHi) */
    const float4* triGeom;   /* 3 x float4 per triangle */
    const float4* triAttr;   /* 6 x float4 per triangle */
    const wpt_instance* instances;
    const wpt_material* materials;
    uint32_t materialCount;
    const wpt_texture* textures;
    const float4* texels4; /* decoded RGBA texels of all image textures */
    const wpt_hotspot* hotspots;
    float invHotspotCount;     /* 1.0f / (float)hotspotCount, divided once on the host (IEEE, the same bits) */
    const float4* hotspotFace; /* per triangle hot spot: unit face normal, face area (wpt_hotspot_face_kernel, at upload) */
    const wpt_sphere* spheres;
    const wpt_rgl_brdf* rglBrdfs; /* measured BRDFs and the pool their tables live in */
    const float* rglData;
    const uint32_t* rglRgbl; /* per measured BRDF: offset of its interleaved colour + luminance table in rglData, or WPT_RGL_NONE (wpt_rgl.h) */
    const float* envM;
    const int32_t* envMs;
    const float* envMcs;
    uint32_t nodeCount, triCount;
    uint32_t boxesMayBeNan;       /* 1: a box coordinate of the tree is NaN (found at upload): every box test checks its slab distances */
    uint32_t hotspotCount;
    uint32_t envType, envCompat;
    int32_t envTex, envN;
    int32_t envLog2N; /* log2(envN) if envN is a power of two, else -1 */
    int32_t envCube[6]; /* WPT_ENV_CUBE: textures +x -x +y -y +z -z */
    const int32_t* envLut; /* envLutSize + 1 entries: first bin whose cumulative importance reaches k / envLutSize, or NULL */
    uint32_t envLutSize;   /* a power of two */
    uint32_t sphereCount;
    const wpt_animation* animations; /* key frame animations of instances and camera */
    const wpt_keyframe* keyframes;
    /* The binary tree collapsed by one level, or NULL where the scene has no wide form (wpt_capi.hip builds it, wpt_pathtrace.inc.h
     * walks it).  8 quadwords per wide node: lo.x, lo.y, lo.z, hi.x, hi.y, hi.z of up to four entries (the node's grandchildren, a
     * child that is a leaf standing for itself, in the reference's order), their references (NODE_CHILD | wide node, a triangle
     * index, PRIM_SPHERE | sphere index, 0xffffffff = no entry), one spare.  Wide node 0 is the root's. */
    const float4* wideNodes;
};

/* ---- animations at a ray's time (wpt_anim.h) ---- */
struct DeviceAnimMath {
    static WPT_D float acos(float x) { return wptm::acosf_(x); }
    static WPT_D float sin(float x) { return wptm::sinf_(x); }
    static WPT_D float sqrt(float x) { return __builtin_sqrtf(x); }
};
/* AnimationCache::get(ai) for a cache at time t (animation.hpp:61-117); recomputed where it is needed */
WPT_D wptanim::Trs animationAt(const SceneView& sv, int ai, float t)
{
    const wpt_animation a = sv.animations[ai];
    return wptanim::at<DeviceAnimMath>(sv.keyframes + a.first_keyframe, a.keyframe_count, t);
}
WPT_D f3 animatePoint(const float* M16, f3 p)
{
    const float in[3] = { p.x, p.y, p.z };
    float out[3];
    wptanim::mulPoint(M16, in, out);
    return mk3(out[0], out[1], out[2]);
}

/* HitableSphere::hit, candidate part (hitable_sphere.hpp:104-147): the nearer root inside
 * (amin, amax), both bounds exclusive, computed without cancellation */
WPT_D bool sphereTest(const wpt_sphere& sp, f3 org, f3 dir, float amin, float amax, float& a)
{
    const f3 oc = sub(org, ld3(sp.center));
    const float ocd = dot(oc, dir);
    const float aq = -ocd;
    const f3 tmp = sub(oc, scl(ocd, dir));
    const float discriminant = sp.radius * sp.radius - dot(tmp, tmp);
    bool hit = false;
    if (discriminant > 0.0f) {
        float a1, a2;
        const float root = __builtin_sqrtf(discriminant);
        if (aq < 0.0f) {
            a2 = aq - root;
            a1 = 2.0f * aq - a2;
        } else {
            a1 = aq + root;
            a2 = 2.0f * aq - a1;
        }
        if (a2 > amin && a2 < amax) {
            a = a2;
            hit = true;
        } else if (a1 > amin && a1 < amax) {
            a = a1;
            hit = true;
        }
    }
    return hit;
}

/* An animated sphere as hit() and direction() see it at a time (hitable_sphere.hpp:118-127,196-203): the animation's
 * translation is added to the centre, its largest scaling multiplies the radius, its rotation follows the sphere's */
WPT_D wpt_sphere sphereMoved(const wpt_sphere& sp, const wptanim::Trs& T)
{
    wpt_sphere r = sp;
    float m = T.s[0];
    if (T.s[1] > m)
        m = T.s[1];
    if (T.s[2] > m)
        m = T.s[2];
    for (int k = 0; k < 3; k++)
        r.center[k] = sp.center[k] + T.t[k];
    r.radius = sp.radius * m;
    /* quaternion product sp.rotation * T.rotation (gvm.hpp:1687-1695) */
    const float x = sp.rotation[0], y = sp.rotation[1], z = sp.rotation[2], w = sp.rotation[3];
    const float qx = T.q[0], qy = T.q[1], qz = T.q[2], qw = T.q[3];
    r.rotation[0] = w * qx + x * qw + y * qz - z * qy;
    r.rotation[1] = w * qy + y * qw + z * qx - x * qz;
    r.rotation[2] = w * qz + z * qw + x * qy - y * qx;
    r.rotation[3] = w * qw - x * qx - y * qy - z * qz;
    return r;
}
template<uint32_t F> WPT_D wpt_sphere sphereAt(const SceneView& sv, const wpt_sphere& sp, float time)
{
    if ((F & FEAT_ANIM) && sp.animation >= 0)
        return sphereMoved(sp, animationAt(sv, sp.animation, time));
    return sp;
}
/* What the out-of-line sphere code needs of the scene, by value: a reference to the scene view, which lives in the
 * kernel's arguments, would make the compiler keep a private copy of ALL kernel arguments in scratch memory as soon as
 * the callee reads more than a field or two of it (the kernels for moving scenes: 670 - 930 bytes per lane). */
struct SphereScene {
    const wpt_sphere* spheres;
    const wpt_animation* animations;
    const wpt_keyframe* keyframes;
};
WPT_D SphereScene sphereScene(const SceneView& sv)
{
    SphereScene v;
    v.spheres = sv.spheres;
    v.animations = sv.animations;
    v.keyframes = sv.keyframes;
    return v;
}
/* ... and as pdfValue() places it (:161-166): the whole transformation applied to the centre */
WPT_D wpt_sphere sphereMovedForPdf(const wpt_sphere& sp, const wptanim::Trs& T)
{
    wpt_sphere r = sp;
    float m = T.s[0];
    if (T.s[1] > m)
        m = T.s[1];
    if (T.s[2] > m)
        m = T.s[2];
    wptanim::applyTrs(T, sp.center, r.center);
    r.radius = sp.radius * m;
    return r;
}

/* HitableSphere::constructHitRecord (hitable_sphere.hpp:42-75) */
/* out of line in the path tracing kernels, where few lanes run it and its registers would be taken from the common path;
 * a kernel that passes it a scene view living in its arguments defines WPT_SPHERE_HIT_INLINE: a real call would need the
 * arguments' address, and the compiler then keeps a private copy of ALL kernel arguments in scratch memory (the ground
 * truth kernel: 880 bytes per lane) */
#ifdef WPT_SPHERE_HIT_INLINE
#define WPT_SPHERE_HIT WPT_D
#else
#define WPT_SPHERE_HIT static __device__ __attribute__((noinline))
#endif
template<uint32_t F = 0> WPT_SPHERE_HIT Hit finishSphereHit(SphereScene sv, Candidate c, f3 org, f3 dir, float time = 0.0f)
{
    wpt_sphere sp = sv.spheres[c.prim & ~PRIM_SPHERE];
    if ((F & FEAT_ANIM) && sp.animation >= 0) {
        const wpt_animation a = sv.animations[sp.animation];
        sp = sphereMoved(sp, wptanim::at<DeviceAnimMath>(sv.keyframes + a.first_keyframe, a.keyframe_count, time));
    }
    Hit h;
    h.a = c.a;
    h.prim = c.prim;
    h.material = sp.material;
    h.p = add(org, scl(c.a, dir));
    f3 n = normalize(sub(h.p, ld3(sp.center)));
    const f3 rn = quatRotate(sp.rotation, n);
    const float alpha = wptm::atan2f_(rn.x, rn.z);
    const float beta = wptm::asinf_(clampr(rn.y, -1.0f, +1.0f));
    h.tc.x = 0.5f * k_inv_pi * (alpha + k_pi);
    h.tc.y = k_inv_pi * (beta + 0.5f * k_pi);
    f3 t = mk3(wptm::cosf_(alpha), 0.0f, -wptm::sinf_(alpha));
    if (__builtin_fabsf(dot(rn, t)) >= k_cosOrthoAngleTolerance)
        t = mk3(0.0f, 0.0f, 0.0f); /* at the poles */
    h.backside = false;
    if (dot(n, neg(dir)) < 0.0f) {
        h.backside = true;
        n = neg(n);
    }
    h.n = n;
    h.t = t;
    return h;
}

/* Rebuilds the HitRecord of the surviving candidate (hitable_triangle.hpp:273-324). */
/* where the walk found the triangles' positions (LDS or HBM), the words behind them are fetched from as well */
struct TriGeomFromScene {
    const float4* triGeom;
    WPT_D float4 operator()(uint32_t i) const { return triGeom[i]; }
};
template<uint32_t F = 0, class Tri4 = TriGeomFromScene>
WPT_D Hit finishHit(const SceneView& sv, const Candidate& c, f3 org, f3 dir, float time, Tri4 tri4)
{
    if ((F & FEAT_SPHERES) && (c.prim & PRIM_SPHERE))
        return finishSphereHit<F>(sphereScene(sv), c, org, dir, time);
    Hit h;
    h.a = c.a;
    h.prim = c.prim;
    const float4 g0 = tri4(3 * c.prim + 0);
    const float4 g1 = tri4(3 * c.prim + 1);
    const float4 g2 = tri4(3 * c.prim + 2);
    const uint32_t instance = __float_as_uint(g0.w);
    h.material = __float_as_uint(g1.w);
    const uint32_t flags = __float_as_uint(g2.w);
    const bool backfacing = c.invDet < 0.0f; /* det < 0 (hitable_triangle.hpp:274) */
    const float bx = c.invDet * c.U, by = c.invDet * c.V, bz = c.invDet * c.W;
    h.p = add(org, scl(c.a, dir));
    const float4* at = sv.triAttr + 6 * (size_t)c.prim;
    const float4 a0 = at[0], a1 = at[1], a2 = at[2], a3 = at[3], a4 = at[4], a5 = at[5];
    /* 24 floats: n0 n1 n2 (9) tc0 tc1 tc2 (6) t0 t1 t2 (9) */
    const f3 n0 = mk3(a0.x, a0.y, a0.z), n1 = mk3(a0.w, a1.x, a1.y), n2 = mk3(a1.z, a1.w, a2.x);
    f3 nrm = add(add(scl(bx, n0), scl(by, n1)), scl(bz, n2));
    const bool transform = (flags & WPT_TRI_TRANSFORM) != 0;
    const float* N = sv.instances[instance].N;
    if (transform)
        nrm = mat3mul(N, nrm);
    float animationN[9];
    const bool animate = (F & FEAT_ANIM) && (flags & WPT_TRI_ANIMATE);
    if (animate) { /* hitable_triangle.hpp:213,296-297 */
        const wptanim::Trs T = animationAt(sv, sv.instances[instance].animation, time);
        wptanim::toMat3(T.q, animationN);
        nrm = mat3mul(animationN, nrm);
    }
    nrm = normalize(nrm);
    if (backfacing)
        nrm = neg(nrm);
    h.n = nrm;
    h.tc.x = 0.0f;
    h.tc.y = 0.0f;
    if (flags & WPT_TRI_HAVE_TEXCOORDS) {
        /* tc0 = (a2.y, a2.z) tc1 = (a2.w, a3.x) tc2 = (a3.y, a3.z) */
        h.tc.x = bx * a2.y + by * a2.w + bz * a3.y;
        h.tc.y = bx * a2.z + by * a3.x + bz * a3.z;
    }
    f3 tan = mk3(0.0f, 0.0f, 0.0f);
    if (flags & WPT_TRI_HAVE_TANGENTS) {
        const f3 t0 = mk3(a3.w, a4.x, a4.y), t1 = mk3(a4.z, a4.w, a5.x), t2 = mk3(a5.y, a5.z, a5.w);
        tan = add(add(scl(bx, t0), scl(by, t1)), scl(bz, t2));
        if (dot(tan, tan) > 0.0f) {
            if (transform)
                tan = mat3mul(N, tan);
            if (animate)
                tan = mat3mul(animationN, tan);
            tan = normalize(sub(tan, scl(dot(nrm, tan), nrm)));
        }
    }
    h.t = tan;
    h.backside = backfacing;
    return h;
}
template<uint32_t F = 0>
WPT_D Hit finishHit(const SceneView& sv, const Candidate& c, f3 org, f3 dir, float time = 0.0f)
{
    TriGeomFromScene fromScene;
    fromScene.triGeom = sv.triGeom;
    return finishHit<F>(sv, c, org, dir, time, fromScene);
}

/* ---- textures (texture.hpp:160-246, texture_image.hpp:85-212, color.hpp:275-294) ---- */
WPT_D float srgbToRgb(float x)
{
    return (x <= 0.04045f ? (x * (1.0f / 12.92f)) : wptm::powf_((x + 0.055f) * (1.0f / 1.055f), 2.4f));
}

/* TextureImage texel decode (texture_image.hpp:85-140): component type, sRGB linearisation of
 * the colour components, grey / grey+alpha / RGB / RGBA expansion.  Runs once per texel when a
 * scene is uploaded (wpt_expand_texels_kernel): the device texel pool holds the decoded RGBA
 * float4 of every texel, so a lookup in the path tracer is one 16-byte load instead of up to
 * four component loads and three pow() evaluations.  Decoding is a pure function of the texel,
 * so the values are the ones the per-lookup decode would give. */
WPT_D f4 imageTexelDecode(const uint8_t* pool, const wpt_texture& t, size_t x, size_t y)
{
    const uint8_t* base = pool + t.texel_offset;
    const size_t idx = (y * (size_t)t.width + x) * t.comps;
    float d[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    const bool lin = t.linearize_srgb != 0;
    for (uint32_t k = 0; k < t.comps; k++) {
        float raw;
        if (t.texel_type == WPT_TEXEL_U8)
            raw = (float)base[idx + k] / 255.0f;
        else if (t.texel_type == WPT_TEXEL_U16)
            raw = (float)((const uint16_t*)base)[idx + k] / 65535.0f;
        else
            raw = ((const float*)base)[idx + k];
        const bool isAlpha = (t.comps == 2 && k == 1) || (t.comps == 4 && k == 3);
        d[k] = (lin && !isAlpha) ? srgbToRgb(raw) : raw;
    }
    if (t.comps == 3)
        return mk4(d[0], d[1], d[2], 1.0f);
    if (t.comps == 4)
        return mk4(d[0], d[1], d[2], d[3]);
    if (t.comps == 1)
        return mk4(d[0], d[0], d[0], 1.0f);
    return mk4(d[0], d[0], d[0], d[1]);
}

/* texel fetch with the reference's clamp at the far edges (texture_image.hpp:142-150); in the
 * device copy of a texture record texel_offset counts float4 texels of the decoded pool */
WPT_D f4 imageTexel(const SceneView& sv, const wpt_texture& t, size_t x, size_t y)
{
    if (x >= t.width)
        x = t.width - 1;
    if (y >= t.height)
        y = t.height - 1;
    const float4 v = sv.texels4[t.texel_offset + y * (size_t)t.width + x];
    return mk4(v.x, v.y, v.z, v.w);
}

WPT_D f4 mix4(f4 a, f4 b, float k)
{
    return mk4(mixr(a.x, b.x, k), mixr(a.y, b.y, k), mixr(a.z, b.z, k), mixr(a.w, b.w, k));
}

/* Texture::value.  TextureTransformer chains are unrolled iteratively: the coordinate
 * transforms apply on the way down, the value transforms on the way back up. */
WPT_D f4 textureValue(const SceneView& sv, int tex, f2 tc)
{
    /* way down */
    int chain[4];
    int depth = 0;
    const wpt_texture* t = sv.textures + tex;
    while (t->type == WPT_TEX_TRANSFORMER && depth < 4) {
        tc.x = t->coord_factor[0] * tc.x + t->coord_offset[0];
        tc.y = t->coord_factor[1] * tc.y + t->coord_offset[1];
        chain[depth++] = tex;
        tex = t->child;
        t = sv.textures + tex;
    }
    f4 val;
    if (t->type == WPT_TEX_CONSTANT) {
        val = ld4(t->a);
    } else if (t->type == WPT_TEX_CHECKER) {
        int row = (int)(tc.y * (float)(int)t->height);
        int col = (int)(tc.x * (float)(int)t->width);
        val = (row % 2 == col % 2) ? ld4(t->a) : ld4(t->b);
    } else {
        float cx = t->coord_factor[0] * tc.x + t->coord_offset[0];
        float cy = t->coord_factor[1] * tc.y + t->coord_offset[1];
        float u = cx - __builtin_floorf(cx);
        float v = cy - __builtin_floorf(cy);
        float uvs = fmaxr(0.0f, (u * (float)t->width) - 0.5f);
        float uvt = fmaxr(0.0f, (v * (float)t->height) - 0.5f);
        size_t x0 = (size_t)uvs;
        size_t y0 = (size_t)uvt;
        float alpha = uvs - (float)x0;
        float beta = uvt - (float)y0;
        f4 v00 = imageTexel(sv, *t, x0, y0);
        f4 v10 = imageTexel(sv, *t, x0 + 1, y0);
        f4 v01 = imageTexel(sv, *t, x0, y0 + 1);
        f4 v11 = imageTexel(sv, *t, x0 + 1, y0 + 1);
        f4 a = mix4(v00, v10, alpha);
        f4 b = mix4(v01, v11, alpha);
        val = add(mul(ld4(t->a), mix4(a, b, beta)), ld4(t->b));
    }
    /* way up */
    while (depth > 0) {
        const wpt_texture* p = sv.textures + chain[--depth];
        val = add(mul(ld4(p->a), val), ld4(p->b));
    }
    return val;
}

/* ---- environment map (envmap.hpp:55-247) ---- */
WPT_D f2 envM(f3 d)
{
    float lat = wptm::asinf_(clampr(d.y, -1.0f, +1.0f));
    float lon = wptm::atan2f_(-d.x, d.z);
    float r = wptm::sinf_(0.5f * (k_pi_2 - lat));
    float alpha = lon - k_pi_2;
    float u, v;
    if (alpha < -k_pi_4)
        alpha += 2.0f * k_pi;
    if (alpha < k_pi_4) {
        u = r;
        v = alpha * u / k_pi_4;
    } else if (alpha < k_pi_2 + k_pi_4) {
        v = r;
        u = -(alpha - k_pi_2) * v / k_pi_4;
    } else if (alpha < k_pi + k_pi_4) {
        u = -r;
        v = (alpha - k_pi) * u / k_pi_4;
    } else {
        v = -r;
        u = -(alpha - (k_pi + k_pi_2)) * v / k_pi_4;
    }
    f2 o;
    o.x = 0.5f * (u + 1.0f);
    o.y = 0.5f * (v + 1.0f);
    return o;
}
WPT_D f3 envInvM(f2 uv)
{
    float u = 2.0f * uv.x - 1.0f;
    float v = 2.0f * uv.y - 1.0f;
    float r, alpha;
    if (u * u > v * v) {
        r = u;
        alpha = k_pi_4 * v / u;
    } else {
        r = v;
        if (__builtin_fabsf(v) > 0.0f)
            alpha = k_pi_2 - k_pi_4 * u / v;
        else
            alpha = 0.0f;
    }
    float lat = k_pi_2 - 2.0f * wptm::asinf_(r);
    float lon = alpha + k_pi_2;
    float slat, clat, slon, clon;
    wptm::sincosf_(lat, &slat, &clat);
    wptm::sincosf_(lon, &slon, &clon);
    return normalize(mk3(-clat * slon, slat, clat * clon));
}
WPT_D f4 textureValue(const SceneView& sv, int tex, f2 tc);
WPT_D f4 envL(const SceneView& sv, f3 dir)
{
    if (sv.envType == WPT_ENV_CUBE) {
        /* EnvironmentMapCube::L (envmap.hpp:265-284) */
        const float ax = __builtin_fabsf(dir.x), ay = __builtin_fabsf(dir.y), az = __builtin_fabsf(dir.z);
        int side;
        f2 tc;
        if (ax > ay && ax > az) {
            tc.x = 0.5f * (dir.z / -dir.x + 1.0f);
            tc.y = 0.5f * (dir.y / ax + 1.0f);
            side = 0 + (__builtin_signbit(dir.x) ? 1 : 0);
        } else if (ay > az) {
            tc.x = 0.5f * (dir.x / ay + 1.0f);
            tc.y = 0.5f * (dir.z / -dir.y + 1.0f);
            side = 2 + (__builtin_signbit(dir.y) ? 1 : 0);
        } else {
            tc.x = 0.5f * (dir.x / dir.z + 1.0f);
            tc.y = 0.5f * (dir.y / az + 1.0f);
            side = 4 + (__builtin_signbit(dir.z) ? 1 : 0);
        }
        return textureValue(sv, sv.envCube[side], tc);
    }
    float y = wptm::asinf_(clampr(dir.y, -1.0f, 1.0f));
    float x = wptm::atan2f_(-dir.x, dir.z);
    if (sv.envCompat == WPT_ENV_COMPAT_MITSUBA) {
        x -= k_pi;
        if (x < 0.0f)
            x += 2.0f * k_pi;
    }
    x *= 0.5f * k_inv_pi;
    y = y * k_inv_pi + 0.5f;
    f2 tc;
    tc.x = x;
    tc.y = y;
    return textureValue(sv, sv.envTex, tc);
}
WPT_D float envP(const SceneView& sv, f3 dir)
{
    const int N = sv.envN;
    f2 uv = envM(dir);
    int x = (int)(uv.x * (float)N);
    int y = (int)(uv.y * (float)N);
    if (x >= N)
        x = N - 1;
    if (y >= N)
        y = N - 1;
    float q = sv.envM[y * N + x];
    float invBinSizeOnSphere = (float)(N * N) * 0.25f * k_inv_pi;
    return q * invBinSizeOnSphere;
}
WPT_D f3 envD(const SceneView& sv, Prng& prng)
{
    const int N = sv.envN;
    float r = in01(prng);
    int bin;
    if (sv.envLut) {
        /* The reference's bisection over all N*N cumulative values (envmap.hpp:171-183) ends at the first
         * bin whose cumulative importance is >= r (the last bin if there is none).  The table is
         * non-decreasing, so that bin lies between the answers for k/K and (k+1)/K with k = floor(r*K),
         * which are tabulated: 2-3 dependent loads instead of 2*log2(N). */
        const uint32_t k = (uint32_t)(r * (float)sv.envLutSize);
        int lo = sv.envLut[k], hi = sv.envLut[k + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sv.envMcs[mid] < r)
                lo = mid + 1;
            else
                hi = mid;
        }
        bin = lo;
    } else {
        int a = 0;
        int b = N * N - 1;
        while (b > a + 1) {
            int c = (a + b) / 2;
            if (sv.envMcs[c] < r)
                a = c;
            else
                b = c;
        }
        bin = (sv.envMcs[a] >= r ? a : b);
    }
    bin = sv.envMs[bin];
    int x, y;
    if (sv.envLog2N >= 0) { /* bin >= 0: mask and shift are % and / */
        x = bin & (N - 1);
        y = bin >> sv.envLog2N;
    } else {
        x = bin % N;
        y = bin / N;
    }
    f2 uv;
    uv.x = ((float)x + in01(prng)) / (float)N;
    uv.y = ((float)y + in01(prng)) / (float)N;
    return envInvM(uv);
}

/* ---- materials ---- */
/* transcendentals of the measured-BRDF model on the device (wpt_rgl.h) */
struct DeviceRglMath {
    static WPT_D float sin(float x) { return wptm::sinf_(x); }
    static WPT_D float cos(float x) { return wptm::cosf_(x); }
    static WPT_D float atan2(float y, float x) { return wptm::atan2f_(y, x); }
    static WPT_D float twiceAsin(float x) { return (float)(2.0 * wptm::asin_d((double)x)); }
};

enum { SCATTER_NONE = 0, SCATTER_EXPLICIT = 1, SCATTER_RANDOM = 2 };
struct Scatter {
    int type;
    f3 dir;
    f4 att;
    float pdf;
    f4 ri;
};
WPT_D Scatter scatterNone()
{
    Scatter s;
    s.type = SCATTER_NONE;
    s.dir = mk3(0.0f, 0.0f, 0.0f);
    s.att = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    s.pdf = 0.0f;
    s.ri = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    return s;
}
WPT_D Scatter scatterMake(int type, f3 dir, f4 att, float pdf, f4 ri)
{
    Scatter s;
    s.type = type;
    s.dir = dir;
    s.att = att;
    s.pdf = pdf;
    s.ri = ri;
    return s;
}

template<uint32_t F> WPT_D f4 texOrConst(const SceneView& sv, int tex, const float* c, f2 tc)
{
    if ((F & FEAT_TEXTURES) && tex >= 0)
        return textureValue(sv, tex, tc);
    return ld4(c);
}

/* What a material reads from its textures at a hit, kept from Material::scatter for the scatterToDirection that the
 * next-event estimation asks of the same material at the same hit (wurblpt.hpp:157 and :199 / :229): the texel of the
 * normal map, albedo / diffuse / specular colours, shininess, roughness.  They are pure functions of material and
 * texture coordinates, so the second evaluation can take the first one's values: the same bits, four bilinear look-ups
 * (sixteen texel loads) fewer per ModPhong hit.  Kernels without textures do not use it. */
struct MatCache {
    bool haveNormalTexel, haveColours, haveAlbedo, haveRoughness, haveRgl;
    wptrgl::RglIncident rgl; /* measured BRDFs: what the model derives from the incident direction alone */
    f4 normalTexel;
    f4 kd, ks;   /* ModPhong: diffuse, specular; Lambertian / GGX: kd = albedo */
    float shininess, rx, ry;
};
WPT_D MatCache matCacheEmpty()
{
    MatCache mc;
    mc.haveNormalTexel = mc.haveColours = mc.haveAlbedo = mc.haveRoughness = mc.haveRgl = false;
    mc.normalTexel = mc.kd = mc.ks = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    mc.shininess = mc.rx = mc.ry = 0.0f;
    return mc;
}
WPT_D f4 normalMapTexel(const SceneView& sv, const wpt_material& m, const Hit& h, MatCache& mc)
{
    if (!mc.haveNormalTexel) {
        mc.normalTexel = textureValue(sv, m.normal_tex, h.tc);
        mc.haveNormalTexel = true;
    }
    return mc.normalTexel;
}

/* Material::normalAt / tangentSpaceAt (material.hpp:195-228) */
template<uint32_t F> WPT_D f3 normalAt(const SceneView& sv, const wpt_material& m, const Hit& h, MatCache& mc)
{
    f3 n = h.n;
    if ((F & FEAT_TEXTURES) && m.normal_tex >= 0) {
        f4 v = normalMapTexel(sv, m, h, mc);
        n = sub(scl(2.0f, mk3(v.x, v.y, v.z)), mk3(1.0f, 1.0f, 1.0f));
        n = normalize(toWorld(frameFromNT(h.n, h.t), n));
    }
    return n;
}
template<uint32_t F> WPT_D Frame tangentSpaceAt(const SceneView& sv, const wpt_material& m, const Hit& h, MatCache& mc)
{
    Frame ts;
    if (dot(h.t, h.t) > k_epsilon) {
        ts = frameFromNT(h.n, h.t);
        if ((F & FEAT_TEXTURES) && m.normal_tex >= 0) {
            f4 v = normalMapTexel(sv, m, h, mc);
            f3 n = sub(scl(2.0f, mk3(v.x, v.y, v.z)), mk3(1.0f, 1.0f, 1.0f));
            n = normalize(toWorld(ts, n));
            f3 t = normalize(sub(h.t, scl(dot(n, h.t), n)));
            ts = frameFromNT(n, t);
        }
    } else {
        ts = frameFromNormal(h.n);
    }
    return ts;
}

template<uint32_t F> WPT_D Frame tangentSpaceAt(const SceneView& sv, const wpt_material& m, const Hit& h)
{
    MatCache mc = matCacheEmpty();
    return tangentSpaceAt<F>(sv, m, h, mc);
}

template<uint32_t F> WPT_D f4 withNir(const wpt_material& m, f4 a)
{
    if (!(m.flags & WPT_MATF_HAVE_NIR))
        a.w = average3(mk3(a.x, a.y, a.z));
    return a;
}

/* GGX helpers (material_ggx.hpp:89-171) */
WPT_D float ggxLambda(f3 v, float rx, float ry)
{
    float a2x = rx * rx, a2y = ry * ry;
    float vx2 = v.x * v.x, vy2 = v.y * v.y, vz2 = v.z * v.z;
    float discriminant = 1.0f + (a2x * vx2 + a2y * vy2) / vz2;
    return 0.5f * (-1.0f + __builtin_sqrtf(discriminant));
}
WPT_D float ggxD(f3 hv, float rx, float ry)
{
    float a2x = rx * rx, a2y = ry * ry;
    float t = (hv.x * hv.x) / a2x + (hv.y * hv.y) / a2y + hv.z * hv.z;
    return 1.0f / (k_pi * rx * ry * t * t);
}
WPT_D float ggxDV(f3 hv, f3 v, float dotVH, float rx, float ry)
{
    float g1 = 1.0f / (1.0f + ggxLambda(v, rx, ry));
    return g1 * dotVH * ggxD(hv, rx, ry) / v.z;
}
WPT_D f4 fresnelSchlick4(f4 r0, float cosTheta) /* fresnel.hpp:48-53 */
{
    float t = 1.0f - cosTheta;
    float t2 = t * t;
    f4 one = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    return add(r0, sclr(sclr(sclr(sub(one, r0), t2), t2), t));
}
WPT_D float fresnelUnpolarized(float cosI, float cosT, float n1, float n2) /* fresnel.hpp:57-72 */
{
    float Fs = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
    Fs *= Fs;
    float Fp = (n1 * cosT - n2 * cosI) / (n1 * cosT + n2 * cosI);
    Fp *= Fp;
    return 0.5f * (Fs + Fp);
}
WPT_D f4 ggxAttenuation(f3 tsH, f3 tsV, f3 tsL, float dotVH, float dotNV, f4 albedo, float rx, float ry)
{
    float Dval = ggxD(tsH, rx, ry);
    f4 Fval = fresnelSchlick4(albedo, dotVH);
    float Gval = 1.0f / (1.0f + ggxLambda(tsV, rx, ry) + ggxLambda(tsL, rx, ry));
    return divs(sclr(scl(Dval, Fval), Gval), 4.0f * dotNV);
}

/* ModPhong helpers (material_modphong.hpp:136-239) */
template<uint32_t F> WPT_D f4 mpDiffuseAt(const SceneView& sv, const wpt_material& m, f2 tc)
{
    return withNir<F>(m, texOrConst<F>(sv, m.tex[0], m.v[0], tc));
}
template<uint32_t F> WPT_D f4 mpSpecularAt(const SceneView& sv, const wpt_material& m, f2 tc)
{
    f4 ks = texOrConst<F>(sv, m.tex[1], m.v[1], tc);
    if (m.flags & WPT_MATF_SPECULAR_TEX_HAS_ALPHA)
        ks = mk4(mixr(ks.x, m.v[1][0], ks.w), mixr(ks.y, m.v[1][1], ks.w), mixr(ks.z, m.v[1][2], ks.w), ks.w);
    return withNir<F>(m, ks);
}
template<uint32_t F> WPT_D float mpShininessAt(const SceneView& sv, const wpt_material& m, f2 tc)
{
    float s = m.f[0];
    if ((F & FEAT_TEXTURES) && m.tex[2] >= 0)
        s *= textureValue(sv, m.tex[2], tc).x;
    return s;
}
WPT_D f4 mpAttenuation(f3 n, f3 v, f3 l, f4 kd, f4 ks, float s, float cosTheta)
{
    f3 r = reflect(neg(l), n);
    float cosRV = fmaxr(dot(r, v), 0.0f);
    f4 spec = sclr(sclr(scl(0.5f, ks), s + 2.0f), wptm::powf_(cosRV, s));
    return sclr(sclr(add(kd, spec), k_inv_pi), fminr(cosTheta, 1.0f));
}
WPT_D float mpSpecularProbability(f4 kd, f4 ks)
{
    float skd = kd.x + kd.y + kd.z + kd.w;
    float sks = ks.x + ks.y + ks.z + ks.w;
    float sum = skd + sks + 1e-4f;
    return clampr(sks / sum, 0.1f, 0.9f);
}
WPT_D float mpPdfValue(f3 n, f3 v, f3 l, float s, float cosTheta, float specProb)
{
    float diffusePdfValue = cosTheta * k_inv_pi;
    f3 r = reflect(neg(v), n);
    float cosRL = fmaxr(dot(r, l), 0.0f);
    float specularPdfValue = 0.5f * k_inv_pi * (s + 1.0f) * wptm::powf_(cosRL, s);
    return mixr(diffusePdfValue, specularPdfValue, specProb);
}

/* resolves MaterialTwoSided (material.hpp:273-320): returns the effective material and clears
 * `backside` when the back material takes over */
template<uint32_t F> WPT_D const wpt_material& resolveMaterial(const SceneView& sv, uint32_t mat, Hit& h)
{
    const wpt_material* m = sv.materials + mat;
    if (F & FEAT_TWOSIDED) {
        for (int guard = 0; guard < 4 && m->type == WPT_MAT_TWOSIDED; guard++) {
            if (h.backside) {
                m = sv.materials + m->tex[1];
                h.backside = false;
            } else {
                m = sv.materials + m->tex[0];
            }
        }
    }
    return *m;
}

/* Material::scatter.  `h` must already be resolved through resolveMaterial. */
template<uint32_t F> WPT_D Scatter materialScatter(const SceneView& sv, const wpt_material& m, const Ray& ray, const Hit& h, Prng& prng, MatCache& mc)
{
    switch (m.type) {
    case WPT_MAT_LAMBERTIAN: { /* material_lambertian.hpp:61-84 */
        if (h.backside)
            return scatterNone();
        f3 cd = cosineDirection(in01x2(prng));
        float cosTheta = cd.z;
        Frame ts = tangentSpaceAt<F>(sv, m, h, mc);
        f3 dir = normalize(toWorld(ts, cd));
        float p = cosTheta * k_inv_pi;
        mc.kd = withNir<F>(m, texOrConst<F>(sv, m.tex[0], m.v[0], h.tc));
        mc.haveAlbedo = true;
        f4 att = sclr(mc.kd, p);
        return scatterMake(SCATTER_RANDOM, dir, att, p, ray.ri);
    }
    case WPT_MAT_RGL: { /* material_rgl.hpp:59-80 */
        if (!(F & FEAT_RGL) || h.backside)
            return scatterNone();
        Frame ts = tangentSpaceAt<F>(sv, m, h, mc);
        const f3 wi = toTangent(ts, neg(ray.d));
        const f2 u = in01x2(prng);
        wptrgl::V3 pwo;
        float p;
        wptrgl::V2 uu;
        uu.x = u.x;
        uu.y = u.y;
        wptrgl::V3 wwi;
        wwi.x = wi.x;
        wwi.y = wi.y;
        wwi.z = wi.z;
        wptrgl::V3 a;
        a.x = a.y = a.z = 0.0f;
        pwo = a;
        p = 0.0f;
        if (!(wwi.z <= 0.0f)) { /* BRDF::sample's own first test */
            wptrgl::rglIncidentCall<DeviceRglMath>(sv.rglBrdfs[m.tex[0]], sv.rglData, wwi, mc.rgl, sv.rglRgbl[m.tex[0]]);
            mc.haveRgl = true;
            a = wptrgl::rglSampleCall<DeviceRglMath>(sv.rglBrdfs[m.tex[0]], sv.rglData, mc.rgl, uu, wwi, pwo, p);
        }
        const f3 wo = mk3(pwo.x, pwo.y, pwo.z);
        if (dot(wo, wo) <= 0.0f)
            return scatterNone();
        const f3 attenuation = mk3(a.x, a.y, a.z);
        f4 att = sclr(mk4(attenuation.x, attenuation.y, attenuation.z, average3(attenuation)), p); /* undo the division by the pdf */
        const f3 dir = normalize(toWorld(ts, wo));
        return scatterMake(SCATTER_RANDOM, dir, att, p, ray.ri);
    }
    case WPT_MAT_GGX: { /* material_ggx.hpp:173-225 */
        if (!(F & FEAT_GGX) || h.backside)
            return scatterNone();
        f3 view = neg(ray.d);
        float rx = m.f[0], ry = m.f[1];
        if ((F & FEAT_TEXTURES) && m.tex[1] >= 0) {
            f4 r = textureValue(sv, m.tex[1], h.tc);
            rx = r.x;
            ry = r.y;
        }
        mc.rx = rx;
        mc.ry = ry;
        mc.haveRoughness = true;
        Frame ts = tangentSpaceAt<F>(sv, m, h, mc);
        f3 tsV = toTangent(ts, view);
        /* sampleVNDF (material_ggx.hpp:138-171) */
        float U1 = in01(prng);
        float U2 = in01(prng);
        f3 Vh = normalize(mk3(rx * tsV.x, ry * tsV.y, tsV.z));
        float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
        f3 T1 = mk3(1.0f, 0.0f, 0.0f);
        if (lensq > 0.0f)
            T1 = sclr(mk3(-Vh.y, Vh.x, 0.0f), 1.0f / __builtin_sqrtf(lensq));
        f3 T2 = cross(Vh, T1);
        float r = __builtin_sqrtf(U1);
        float phi = 2.0f * k_pi * U2;
        float sphi, cphi;
        wptm::sincosf_(phi, &sphi, &cphi);
        float t1 = r * cphi;
        float t2 = r * sphi;
        float s = 0.5f * (1.0f + Vh.z);
        t2 = (1.0f - s) * __builtin_sqrtf(1.0f - t1 * t1) + s * t2;
        f3 Nh = add(add(scl(t1, T1), scl(t2, T2)), scl(__builtin_sqrtf(fmaxr(0.0f, 1.0f - t1 * t1 - t2 * t2)), Vh));
        f3 tsH = normalize(mk3(rx * Nh.x, ry * Nh.y, fmaxr(0.0f, Nh.z)));
        f3 tsL = reflect(neg(tsV), tsH);
        f3 light = toWorld(ts, tsL);
        float l = dot(light, light);
        if (l < k_epsilon)
            return scatterNone();
        f3 dir = divs(light, __builtin_sqrtf(l));
        float dotVH = dot(tsV, tsH);
        float p = ggxDV(tsH, tsV, dotVH, rx, ry) / (4.0f * dotVH);
        if (!__builtin_isfinite(p) || p < 0.0f)
            return scatterNone();
        f4 att = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        float dotNL = dot(ts.n, light);
        float dotNV = dot(ts.n, view);
        if (dotNL > 0.0f && dotNV > 0.0f) {
            f4 albedo = texOrConst<F>(sv, m.tex[0], m.v[0], h.tc);
            mc.kd = albedo;
            mc.haveAlbedo = true;
            att = ggxAttenuation(tsH, tsV, tsL, dotVH, dotNV, albedo, rx, ry);
        }
        return scatterMake(SCATTER_RANDOM, dir, att, p, ray.ri);
    }
    case WPT_MAT_GLASS: { /* material_glass.hpp:91-152 */
        if (!(F & FEAT_GLASS))
            return scatterNone();
        f4 att = mk4(1.0f, 1.0f, 1.0f, 1.0f);
        f4 ourRI = ld4(m.v[1]);
        f4 theirRI = ld4(m.v[2]);
        int riIndex = 0;
        if (m.flags & WPT_MATF_CHROMATIC_DISPERSION) {
            riIndex = (int)(in01(prng) * 4.0f);
            att = mk4(riIndex == 0 ? 4.0f : 0.0f, riIndex == 1 ? 4.0f : 0.0f, riIndex == 2 ? 4.0f : 0.0f, riIndex == 3 ? 4.0f : 0.0f);
        }
        if (h.backside) {
            f4 tmp = ourRI;
            ourRI = theirRI;
            theirRI = tmp;
            float dist = h.a;
            /* exp(-absorption * dist) per channel; channels with equal absorption (the usual grey glass)
             * share one evaluation: same argument, same value */
            const float a0 = m.v[0][0], a1 = m.v[0][1], a2 = m.v[0][2], a3 = m.v[0][3];
            f4 e;
            e.x = wptm::expf_(-a0 * dist);
            e.y = a1 == a0 ? e.x : wptm::expf_(-a1 * dist);
            e.z = a2 == a0 ? e.x : (a2 == a1 ? e.y : wptm::expf_(-a2 * dist));
            e.w = a3 == a0 ? e.x : (a3 == a1 ? e.y : (a3 == a2 ? e.z : wptm::expf_(-a3 * dist)));
            att = mul(att, e);
        }
        f3 n = normalAt<F>(sv, m, h, mc);
        float ours = comp(ourRI, riIndex), theirs = comp(theirRI, riIndex);
        f3 refracted = refract(ray.d, n, theirs / ours);
        bool doReflection = true;
        if (dot(refracted, refracted) > 0.0f) {
            float cosIncident = dot(neg(ray.d), n);
            float cosTransmitted = -dot(refracted, n);
            float fresnel = fresnelUnpolarized(cosIncident, cosTransmitted, theirs, ours);
            doReflection = in01(prng) < fresnel;
        }
        if (doReflection)
            return scatterMake(SCATTER_EXPLICIT, normalize(reflect(ray.d, n)), att, 0.0f, theirRI);
        return scatterMake(SCATTER_EXPLICIT, normalize(refracted), att, 0.0f, ourRI);
    }
    case WPT_MAT_MIRROR: { /* material_mirror.hpp:53-62 */
        if (!(F & FEAT_GLASS) || h.backside)
            return scatterNone();
        f3 reflected = reflect(ray.d, normalAt<F>(sv, m, h, mc));
        f4 att = withNir<F>(m, texOrConst<F>(sv, m.tex[0], m.v[0], h.tc));
        return scatterMake(SCATTER_EXPLICIT, normalize(reflected), att, 0.0f, ray.ri);
    }
    case WPT_MAT_MODPHONG: { /* material_modphong.hpp:241-308 */
        if (!(F & FEAT_MODPHONG))
            return scatterNone();
        float opa;
        if ((F & FEAT_TEXTURES) && m.tex[3] >= 0)
            opa = textureValue(sv, m.tex[3], h.tc).x;
        else if ((F & FEAT_TEXTURES) && (m.flags & WPT_MATF_DIFFUSE_TEX_HAS_ALPHA))
            opa = textureValue(sv, m.tex[0], h.tc).w;
        else
            opa = m.f[1];
        bool transparent = (opa < 1.0f && opa < in01(prng));
        if (transparent) {
            float ourRI = m.f[2];
            float theirRI = 1.0f;
            if (h.backside) {
                float tmp = ourRI;
                ourRI = theirRI;
                theirRI = tmp;
            }
            f3 n = normalize(normalAt<F>(sv, m, h, mc));
            f3 refracted = refract(ray.d, n, theirRI / ourRI);
            float l = dot(refracted, refracted);
            if (l < k_epsilon)
                return scatterNone();
            refracted = divs(refracted, __builtin_sqrtf(l));
            f4 att = withNir<F>(m, ld4(m.v[2]));
            return scatterMake(SCATTER_EXPLICIT, refracted, att, 0.0f, mk4(ourRI, ourRI, ourRI, ourRI));
        }
        if (h.backside)
            return scatterNone();
        f4 kd = mpDiffuseAt<F>(sv, m, h.tc);
        f4 ks = mpSpecularAt<F>(sv, m, h.tc);
        float s = mpShininessAt<F>(sv, m, h.tc);
        mc.kd = kd;
        mc.ks = ks;
        mc.shininess = s;
        mc.haveColours = true;
        float specProb = mpSpecularProbability(kd, ks);
        f3 dir, n;
        float cosTheta;
        if (in01(prng) < specProb) {
            float r1 = in01(prng);
            float r2 = in01(prng);
            float cosThetaSpec = wptm::powf_(1.0f - r1, 1.0f / (1.0f + s));
            float discriminant = fmaxr(1.0f - cosThetaSpec * cosThetaSpec, 0.0f);
            float sinThetaSpec = __builtin_sqrtf(discriminant);
            float phi = 2.0f * k_pi * r2;
            float sphi, cphi;
            wptm::sincosf_(phi, &sphi, &cphi);
            float x = cphi * sinThetaSpec;
            float y = sphi * sinThetaSpec;
            float z = cosThetaSpec;
            n = normalAt<F>(sv, m, h, mc);
            Frame specTS = frameFromNormal(reflect(ray.d, n));
            dir = normalize(toWorld(specTS, mk3(x, y, z)));
            cosTheta = fmaxr(dot(dir, n), 0.0f);
        } else {
            Frame ts = tangentSpaceAt<F>(sv, m, h, mc);
            n = ts.n;
            f3 cd = cosineDirection(in01x2(prng));
            cosTheta = cd.z;
            dir = normalize(toWorld(ts, cd));
        }
        f4 att = mpAttenuation(n, neg(ray.d), dir, kd, ks, s, cosTheta);
        float p = mpPdfValue(n, neg(ray.d), dir, s, cosTheta, specProb);
        return scatterMake(SCATTER_RANDOM, dir, att, p, ray.ri);
    }
    default:
        return scatterNone();
    }
}

/* Material::scatterToDirection: attenuation and pdf for a given direction */
template<uint32_t F> WPT_D void materialEval(const SceneView& sv, const wpt_material& m, const Ray& ray, const Hit& h, f3 direction,
        f4& att, float& p, MatCache& mc)
{
    att = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    p = 0.0f;
    switch (m.type) {
    case WPT_MAT_LAMBERTIAN: { /* material_lambertian.hpp:86-102 */
        float cosTheta = dot(normalAt<F>(sv, m, h, mc), direction);
        if (cosTheta > 0.0f) {
            p = cosTheta * k_inv_pi;
            if (!((F & FEAT_TEXTURES) && mc.haveAlbedo))
                mc.kd = withNir<F>(m, texOrConst<F>(sv, m.tex[0], m.v[0], h.tc));
            att = sclr(mc.kd, p);
        }
        break;
    }
    case WPT_MAT_RGL: { /* material_rgl.hpp:82-98 */
        if (!(F & FEAT_RGL))
            break;
        Frame ts = tangentSpaceAt<F>(sv, m, h, mc);
        if (dot(ts.n, direction) > 0.0f) {
            const f3 wo = toTangent(ts, direction);
            const f3 wi = toTangent(ts, neg(ray.d));
            wptrgl::V3 wwi, wwo;
            wwi.x = wi.x;
            wwi.y = wi.y;
            wwi.z = wi.z;
            wwo.x = wo.x;
            wwo.y = wo.y;
            wwo.z = wo.z;
            const wpt_rgl_brdf& b = sv.rglBrdfs[m.tex[0]];
            /* BRDF::eval and BRDF::pdf share the half vector's part; what depends on the incident direction alone comes
             * from scatter where it has been there (same arguments, same values) */
            wptrgl::V3 a;
            a.x = a.y = a.z = 0.0f;
            p = 0.0f;
            if (!(wwi.z <= 0.0f || wwo.z <= 0.0f)) {
                if (!mc.haveRgl) {
                    wptrgl::rglIncidentCall<DeviceRglMath>(b, sv.rglData, wwi, mc.rgl, sv.rglRgbl[m.tex[0]]);
                    mc.haveRgl = true;
                }
                wptrgl::rglEvalPdfCall<DeviceRglMath>(b, sv.rglData, mc.rgl, wwi, wwo, a, p);
            }
            const f3 attenuation = mk3(a.x, a.y, a.z);
            att = mk4(attenuation.x, attenuation.y, attenuation.z, average3(attenuation));
        }
        break;
    }
    case WPT_MAT_GGX: { /* material_ggx.hpp:227-257 */
        if (!(F & FEAT_GGX))
            break;
        f3 view = neg(ray.d);
        f3 light = direction;
        Frame ts = tangentSpaceAt<F>(sv, m, h, mc);
        float dotNL = dot(ts.n, light);
        float dotNV = dot(ts.n, view);
        if (dotNL > 0.0f && dotNV > 0.0f) {
            float rx = m.f[0], ry = m.f[1];
            if ((F & FEAT_TEXTURES) && mc.haveRoughness) {
                rx = mc.rx;
                ry = mc.ry;
            } else if ((F & FEAT_TEXTURES) && m.tex[1] >= 0) {
                f4 r = textureValue(sv, m.tex[1], h.tc);
                rx = r.x;
                ry = r.y;
            }
            f3 tsV = toTangent(ts, view);
            f3 tsL = toTangent(ts, light);
            f3 tsH = normalize(add(tsV, tsL));
            float dotVH = dot(tsV, tsH);
            if (dotVH > 0.0f) {
                p = ggxDV(tsH, tsV, dotVH, rx, ry) / (4.0f * dotVH);
                f4 albedo = ((F & FEAT_TEXTURES) && mc.haveAlbedo) ? mc.kd : texOrConst<F>(sv, m.tex[0], m.v[0], h.tc);
                att = ggxAttenuation(tsH, tsV, tsL, dotVH, dotNV, albedo, rx, ry);
            }
        }
        break;
    }
    case WPT_MAT_MODPHONG: { /* material_modphong.hpp:310-327 */
        if (!(F & FEAT_MODPHONG))
            break;
        f3 n = normalAt<F>(sv, m, h, mc);
        float cosTheta = dot(n, direction);
        if (cosTheta > 0.0f) {
            f4 kd, ks;
            float s;
            if ((F & FEAT_TEXTURES) && mc.haveColours) {
                kd = mc.kd;
                ks = mc.ks;
                s = mc.shininess;
            } else {
                kd = mpDiffuseAt<F>(sv, m, h.tc);
                ks = mpSpecularAt<F>(sv, m, h.tc);
                s = mpShininessAt<F>(sv, m, h.tc);
            }
            float specProb = mpSpecularProbability(kd, ks);
            att = mpAttenuation(n, neg(ray.d), direction, kd, ks, s, cosTheta);
            p = mpPdfValue(n, neg(ray.d), direction, s, cosTheta, specProb);
        }
        break;
    }
    default:
        break;
    }
}

/* Material::emitted */
template<uint32_t F> WPT_D f4 materialEmitted(const SceneView& sv, const wpt_material& m, const Hit& h)
{
    if (m.type == WPT_MAT_LIGHT_DIFFUSE) { /* light_diffuse.hpp:50-61 */
        f4 e = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!h.backside) {
            e = ld4(m.v[0]);
            if ((F & FEAT_TEXTURES) && m.tex[0] >= 0) {
                f4 c = textureValue(sv, m.tex[0], h.tc);
                e = mul(e, mk4(c.x, c.y, c.z, average3(mk3(c.x, c.y, c.z))));
            }
        }
        return e;
    }
    if ((F & FEAT_MODPHONG) && m.type == WPT_MAT_MODPHONG) { /* material_modphong.hpp:183-190 */
        f4 e = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!h.backside)
            e = withNir<F>(m, texOrConst<F>(sv, m.tex[4], m.v[3], h.tc));
        return e;
    }
    return mk4(0.0f, 0.0f, 0.0f, 0.0f);
}

WPT_D float powerHeuristicWeight(float f, float g) /* wurblpt.hpp:101-106 */
{
    f *= f;
    g *= g;
    return (f + g > 0.0f ? f / (f + g) : 0.0f);
}

} /* namespace wptd */

#endif
