/*
 * wpt_oracle.cpp -- CPU restatement of WurblPT's per-pixel Monte Carlo integrator.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (wurblpt_amd/csrc)
 * never includes, links or calls anything in this directory.
 *
 * It restates, scalar and in the reference's own operation order, the algorithm of
 * /root/reference/libwurblpt (each function cites the file:line it follows) on the
 * flattened scene of include/wurblpt_hip.h.  Parity status:
 *   - Prng, Sampler, TangentSpace, Fresnel, AABB::mayHit, RayIntersectionHelper,
 *     BVH::hit order, Camera::getRay, reflect/refract, quaternion rotate:
 *     PINNED bit-for-bit against the reference's own headers compiled from
 *     /root/reference (oracle/ref_probe.cpp -> tests/golden/ref_*.json).
 *   - triangle hit/pdfValue/direction, materials, textures, environment map, tracePath,
 *     mcpt: those reference headers include <tgd/array.hpp> (libtgd, an external
 *     library that is absent here), so they cannot be compiled; they are pinned by the
 *     reference's in-tree Mitsuba render (statistically) and by the work-per-sample
 *     figures recorded in SURVEY.md section 6; see DESIGN.md "Oracle".
 *
 * Two math back ends: default = wpt_math.h (same bits as the GPU); -DWPT_ORACLE_LIBM =
 * std:: functions as the reference uses (gvm.hpp:118-146).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <limits>
#include <algorithm>
#include <vector>
#include <omp.h>

#include "../include/wurblpt_hip.h"
#include "../wurblpt_amd/csrc/wpt_rgl.h"
#include "../wurblpt_amd/csrc/wpt_lens.h"
#include "../wurblpt_amd/csrc/wpt_anim.h"
#include "../wurblpt_amd/csrc/wpt_postproc.h"
#include "../wurblpt_amd/csrc/wpt_math.h"

namespace {

/* ---- math back end (gvm.hpp:118-146) ---- */
#ifdef WPT_ORACLE_LIBM
inline float m_sin(float x) { return std::sin(x); }
inline float m_cos(float x) { return std::cos(x); }
inline float m_exp(float x) { return std::exp(x); }
inline float m_pow(float x, float y) { return std::pow(x, y); }
inline float m_asin(float x) { return std::asin(x); }
inline float m_acos(float x) { return std::acos(x); }
inline float m_atan2(float y, float x) { return std::atan2(y, x); }
#else
inline float m_sin(float x) { return wptm::sinf_(x); }
inline float m_cos(float x) { return wptm::cosf_(x); }
inline float m_exp(float x) { return wptm::expf_(x); }
inline float m_pow(float x, float y) { return wptm::powf_(x, y); }
inline float m_asin(float x) { return wptm::asinf_(x); }
inline float m_acos(float x) { return wptm::acosf_(x); }
inline float m_atan2(float y, float x) { return wptm::atan2f_(y, x); }
#endif

/* the measured-BRDF model (written once, see the header); its transcendentals come from this back end */
struct OracleMath {
    static float sin(float x) { return m_sin(x); }
    static float cos(float x) { return m_cos(x); }
    static float atan2(float y, float x) { return m_atan2(y, x); }
    static float acos(float x) { return m_acos(x); }
    static float sqrt(float x) { return std::sqrt(x); }
#ifdef WPT_ORACLE_LIBM
    static float twiceAsin(float x) { return float(2.0 * ::asin(double(x))); }
#else
    static float twiceAsin(float x) { return float(2.0 * wptm::asin_d(double(x))); }
#endif
};

constexpr float k_pi = 3.1415926535897932384626433832795029L;
constexpr float k_pi_2 = 1.5707963267948966192313216916397514L;
constexpr float k_pi_4 = 0.7853981633974483096156608458198757L;
constexpr float k_inv_pi = 0.3183098861837906715377675267450287L;
constexpr float k_maxval = std::numeric_limits<float>::max();
constexpr float k_epsilon = std::numeric_limits<float>::epsilon();

/* gvm.hpp:88-98: min/max/clamp with the `x < y ? x : y` NaN behaviour */
inline float fmin_(float x, float y) { return x < y ? x : y; }
inline float fmax_(float x, float y) { return x > y ? x : y; }
inline float clamp_(float x, float lo, float hi) { return fmin_(hi, fmax_(lo, x)); }
inline float mix_(float x, float y, float a) { return x + a * (y - x); } /* gvm.hpp:166 */

/* ---- vectors (gvm.hpp:979-1025,1183-1232) ---- */
struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
struct V4 {
    float x, y, z, w;
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
    float& at(int i) { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
};

inline V3 v3(float a) { return V3 { a, a, a }; }
inline V3 v3(const float* p) { return V3 { p[0], p[1], p[2] }; }
inline V4 v4(float a) { return V4 { a, a, a, a }; }
inline V4 v4(const float* p) { return V4 { p[0], p[1], p[2], p[3] }; }

inline V2 operator+(V2 a, V2 b) { return V2 { a.x + b.x, a.y + b.y }; }
inline V2 operator-(V2 a, V2 b) { return V2 { a.x - b.x, a.y - b.y }; }
inline V2 operator*(V2 a, V2 b) { return V2 { a.x * b.x, a.y * b.y }; }
inline V2 operator*(float s, V2 a) { return V2 { s * a.x, s * a.y }; }
inline V2 operator*(V2 a, float s) { return V2 { a.x * s, a.y * s }; }

inline V3 operator+(V3 a, V3 b) { return V3 { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return V3 { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator*(V3 a, V3 b) { return V3 { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline V3 operator-(V3 a) { return V3 { -a.x, -a.y, -a.z }; }
inline V3 operator*(float s, V3 a) { return V3 { s * a.x, s * a.y, s * a.z }; }
inline V3 operator*(V3 a, float s) { return V3 { a.x * s, a.y * s, a.z * s }; }
inline V3 operator/(V3 a, float s) { return V3 { a.x / s, a.y / s, a.z / s }; }
inline V3 operator/(float s, V3 a) { return V3 { s / a.x, s / a.y, s / a.z }; }

inline V4 operator+(V4 a, V4 b) { return V4 { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline V4 operator-(V4 a, V4 b) { return V4 { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
inline V4 operator*(V4 a, V4 b) { return V4 { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }
inline V4 operator-(V4 a) { return V4 { -a.x, -a.y, -a.z, -a.w }; }
inline V4 operator*(float s, V4 a) { return V4 { s * a.x, s * a.y, s * a.z, s * a.w }; }
inline V4 operator*(V4 a, float s) { return V4 { a.x * s, a.y * s, a.z * s, a.w * s }; }
inline V4 operator/(V4 a, float s) { return V4 { a.x / s, a.y / s, a.z / s, a.w / s }; }

/* gvm.hpp:1183-1189: d = 0; d += a[i]*b[i] */
inline float dot(V2 a, V2 b) { float d = 0.0f; d += a.x * b.x; d += a.y * b.y; return d; }
inline float dot(V3 a, V3 b) { float d = 0.0f; d += a.x * b.x; d += a.y * b.y; d += a.z * b.z; return d; }
inline float dot(V4 a, V4 b) { float d = 0.0f; d += a.x * b.x; d += a.y * b.y; d += a.z * b.z; d += a.w * b.w; return d; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }           /* gvm.hpp:1191 */
inline V3 normalize(V3 v) { return v / length(v); }                 /* gvm.hpp:1201 */
inline V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }  /* gvm.hpp:1213 */
inline V3 refract(V3 i, V3 n, float eta)                            /* gvm.hpp:1218 */
{
    const float d = dot(n, i);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    return k <= 0.0f ? v3(0.0f) : i * eta - n * (eta * d + std::sqrt(k));
}
inline V3 cross(V3 v, V3 w) /* gvm.hpp:1225 */
{
    return V3 { v.y * w.z - v.z * w.y, v.z * w.x - v.x * w.z, v.x * w.y - v.y * w.x };
}
inline float max4(V4 a) /* gvm.hpp:1282 */
{
    float r = a.x;
    if (a.y > r) r = a.y;
    if (a.z > r) r = a.z;
    if (a.w > r) r = a.w;
    return r;
}
inline float min4(V4 a) /* gvm.hpp:1273 */
{
    float r = a.x;
    if (a.y < r) r = a.y;
    if (a.z < r) r = a.z;
    if (a.w < r) r = a.w;
    return r;
}
inline float average3(V3 a) /* gvm.hpp:1291 */
{
    constexpr float inv_N = 1.0f / 3.0f;
    float sum = 0;
    sum += a.x; sum += a.y; sum += a.z;
    return inv_N * sum;
}
inline V3 rgb(V4 a) { return V3 { a.x, a.y, a.z }; }
inline V4 mix4(V4 x, V4 y, float a) /* gvm.hpp: vector mix = x + alpha * (y - x) */
{
    return V4 { mix_(x.x, y.x, a), mix_(x.y, y.y, a), mix_(x.z, y.z, a), mix_(x.w, y.w, a) };
}
inline V3 mix3(V3 x, V3 y, float a)
{
    return V3 { mix_(x.x, y.x, a), mix_(x.y, y.y, a), mix_(x.z, y.z, a) };
}

/* column-major mat3 * vec3 (gvm.hpp:1481-1491): r[i] = 0; r[i] += m[j][i]*w[j] */
inline V3 mat3_mul(const float* m, V3 w)
{
    float r[3];
    for (int i = 0; i < 3; i++) {
        r[i] = 0.0f;
        r[i] += m[0 * 3 + i] * w.x;
        r[i] += m[1 * 3 + i] * w.y;
        r[i] += m[2 * 3 + i] * w.z;
    }
    return V3 { r[0], r[1], r[2] };
}
inline V3 mat4_mul_point(const float* m, V3 p) /* (M * vec4(p, 1)).xyz() */
{
    float r[3];
    for (int i = 0; i < 3; i++) {
        r[i] = 0.0f;
        r[i] += m[0 * 4 + i] * p.x;
        r[i] += m[1 * 4 + i] * p.y;
        r[i] += m[2 * 4 + i] * p.z;
        r[i] += m[3 * 4 + i] * 1.0f;
    }
    return V3 { r[0], r[1], r[2] };
}
/* quaternion * vec3 (gvm.hpp:1713-1720) */
inline V3 quat_rotate(const float* q, V3 v)
{
    V3 s = V3 { q[0], q[1], q[2] };
    V3 t = 2.0f * cross(s, v);
    return v + q[3] * t + cross(s, t);
}

/* ---- prng.hpp:47-101 ---- */
struct Prng {
    uint32_t s[4];
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    static uint64_t splitmix64(uint64_t x)
    {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    explicit Prng(unsigned int pixelIndex)
    {
        uint64_t seed = pixelIndex;
        seed += 42;
        uint64_t s01 = splitmix64(seed);
        uint64_t s23 = splitmix64(s01);
        s[0] = s01 >> 32;
        s[1] = s01 & 0xffffffffull;
        s[2] = s23 >> 32;
        s[3] = s23 & 0xffffffffull;
    }
    uint32_t next()
    {
        const uint32_t result = s[0] + s[3];
        const uint32_t t = s[1] << 9;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 11);
        return result;
    }
    float in01()
    {
        uint32_t x = next();
        return (x >> 8) * 0x1.0p-24;
    }
    /* prng.hpp:97-100 `vec2(in01(), in01())`: g++ 11.4 evaluates the SECOND argument
     * first, so .y receives the first draw (pinned by ref_probe). */
    V2 in01x2()
    {
        float second = in01();
        float first = in01();
        return V2 { first, second };
    }
};

/* ---- sampler.hpp:39-123 ---- */
inline V2 inUnitDisk(V2 u)
{
    V2 uOffset = 2.0f * u - V2 { 1.0f, 1.0f };
    V2 result;
    if (uOffset.x == 0.0f && uOffset.y == 0.0f) {
        result = V2 { 0.0f, 0.0f };
    } else {
        float theta, r;
        if (std::fabs(uOffset.x) > std::fabs(uOffset.y)) {
            r = uOffset.x;
            theta = k_pi_4 * (uOffset.y / uOffset.x);
        } else {
            r = uOffset.y;
            theta = k_pi_2 - k_pi_4 * (uOffset.x / uOffset.y);
        }
        result = r * V2 { m_cos(theta), m_sin(theta) };
    }
    return result;
}
inline V3 inTriangle(V2 u)
{
    float su0 = std::sqrt(u.x);
    float b0 = 1.0f - su0;
    float b1 = u.y * su0;
    return V3 { b0, b1, 1.0f - b0 - b1 };
}
inline V3 cosineDirection(V2 u)
{
    V2 d = inUnitDisk(u);
    float z = std::sqrt(fmax_(0.0f, 1.0f - dot(d, d)));
    return V3 { d.x, d.y, z };
}

/* ---- tangentspace.hpp:46-136 ---- */
struct TangentSpace {
    V3 normal, tangent, bitangent;
    TangentSpace() {}
    explicit TangentSpace(V3 n) : normal(n)
    {
        float sign = std::copysign(1.0f, normal.z);
        float a = -1.0f / (sign + normal.z);
        float b = normal.x * normal.y * a;
        tangent = V3 { 1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x };
        bitangent = V3 { b, sign + normal.y * normal.y * a, -normal.y };
    }
    TangentSpace(V3 n, V3 t) : normal(n), tangent(t), bitangent(cross(n, t)) {}
    V3 toTangentSpace(V3 v) const
    {
        /* matrixToTangentSpace(): columns (t.x,b.x,n.x), (t.y,b.y,n.y), (t.z,b.z,n.z) */
        float m[9] = { tangent.x, bitangent.x, normal.x, tangent.y, bitangent.y, normal.y,
            tangent.z, bitangent.z, normal.z };
        return mat3_mul(m, v);
    }
    V3 toWorldSpace(V3 v) const
    {
        float m[9] = { tangent.x, tangent.y, tangent.z, bitangent.x, bitangent.y, bitangent.z,
            normal.x, normal.y, normal.z };
        return mat3_mul(m, v);
    }
};

/* sampler.hpp:69-77 */
inline V3 onUnitSphere(V2 u)
{
    float z = 1.0f - 2.0f * u.x;
    float r = std::sqrt(fmax_(0.0f, 1.0f - z * z));
    float phi = 2.0f * k_pi * u.y;
    return V3 { r * m_cos(phi), r * m_sin(phi), z };
}
/* sampler.hpp:112-120 */
inline V3 toSphere(V3 direction, float cosThetaMax, V2 u)
{
    float cosTheta = (1.0f - u.x) + u.x * cosThetaMax;
    float sinTheta = std::sqrt(fmax_(0.0f, 1.0f - cosTheta * cosTheta));
    float phi = u.y * 2.0f * k_pi;
    V3 vectorAroundZ = V3 { m_cos(phi) * sinTheta, m_sin(phi) * sinTheta, cosTheta };
    return normalize(TangentSpace(direction).toWorldSpace(vectorAroundZ));
}

/* ---- fresnel.hpp:48-72 ---- */
inline V4 fresnelSchlick(V4 r0, float cosTheta)
{
    float t = 1.0f - cosTheta;
    float t_squared = t * t;
    return r0 + (v4(1.0f) - r0) * t_squared * t_squared * t;
}
inline float fresnelUnpolarized(float cosI, float cosT, float n1, float n2)
{
    float Fs = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
    Fs *= Fs;
    float Fp = (n1 * cosT - n2 * cosI) / (n1 * cosT + n2 * cosI);
    Fp *= Fp;
    return 0.5f * (Fs + Fp);
}

/* ---- ray.hpp:36-56, hitable.hpp:39-113 ---- */
struct Ray {
    V3 origin, direction;
    float time;
    V4 refractiveIndex;
    V3 at(float a) const { return origin + a * direction; }
};
struct HitRecord {
    bool haveHit = false;
    float a;
    V3 position, normal, tangent;
    V2 texcoords;
    bool backside;
    uint32_t prim; /* stands for `const Hitable* hitable` */
};
struct RayHelper {
    V3 invDirection;
    int kx, ky, kz;
    V3 S;
    explicit RayHelper(const Ray& ray)
    {
        invDirection = 1.0f / ray.direction;
        V3 absdir = V3 { std::fabs(ray.direction.x), std::fabs(ray.direction.y), std::fabs(ray.direction.z) };
        if (absdir.z >= absdir.y && absdir.z >= absdir.x)
            kz = 2;
        else if (absdir.y >= absdir.x)
            kz = 1;
        else
            kz = 0;
        kx = kz + 1;
        if (kx == 3)
            kx = 0;
        ky = kx + 1;
        if (ky == 3)
            ky = 0;
        if (ray.direction[kz] < 0.0f) {
            int tmp = kx;
            kx = ky;
            ky = tmp;
        }
        S.x = ray.direction[kx] * invDirection[kz];
        S.y = ray.direction[ky] * invDirection[kz];
        S.z = invDirection[kz];
    }
};

/* ---- aabb.hpp:70-86 ---- */
inline bool aabbMayHit(const float* lo, const float* hi, const Ray& ray, float amin, float amax, V3 invDir)
{
    V3 t0 = (v3(lo) - ray.origin) * invDir;
    V3 t1 = (v3(hi) - ray.origin) * invDir;
    V4 tmin = V4 { amin, fmin_(t0.x, t1.x), fmin_(t0.y, t1.y), fmin_(t0.z, t1.z) };
    V4 tmax = V4 { amax, fmax_(t0.x, t1.x), fmax_(t0.y, t1.y), fmax_(t0.z, t1.z) };
    return max4(tmin) <= min4(tmax);
}

struct Ctx {
    const wpt_scene_desc* sc;
    const wpt_params* pr;
    wpt_counters cnt;
    float time = 0.0f; /* stands for the thread's AnimationCache: the time of the path being traced (wurblpt.hpp:361) */
};

/* AnimationCache::get / getM / getN of animation `ai` at the path's time (animation.hpp:61-117) */
inline wptanim::Trs animationAt(const Ctx& c, int ai, float t)
{
    const wpt_animation& a = c.sc->animations[ai];
    return wptanim::at<OracleMath>(c.sc->keyframes + a.first_keyframe, a.keyframe_count, t);
}
inline V3 animatePoint(const float* M16, V3 p)
{
    const float in[3] = { p.x, p.y, p.z };
    float out[3];
    wptanim::mulPoint(M16, in, out);
    return V3 { out[0], out[1], out[2] };
}

/* ---- hitable_triangle.hpp:189-325 ---- */
inline float xorf(float a, uint32_t b)
{
    return wptm::bits_to_float(wptm::float_to_bits(a) ^ b);
}

inline HitRecord triangleHit(const Ctx& c, uint32_t prim, const Ray& ray, const RayHelper& rh,
        float amin, float amax, bool fullInfo, V3* v0v1, V3* v0v2)
{
    const wpt_tri_geom& g = c.sc->tri_geom[prim];
    V3 v0 = v3(g.v0), v1 = v3(g.v1), v2 = v3(g.v2); /* TRANSFORM already applied (host) */
    const bool animate = (g.flags & WPT_TRI_ANIMATE) != 0;
    float animationN[9];
    if (animate) { /* hitable_triangle.hpp:209-218 */
        const wptanim::Trs T = animationAt(c, c.sc->instances[g.instance].animation, c.time);
        float animationM[16];
        wptanim::toMat4(T, animationM);
        wptanim::toMat3(T.q, animationN);
        v0 = animatePoint(animationM, v0);
        v1 = animatePoint(animationM, v1);
        v2 = animatePoint(animationM, v2);
    }
    const V3 A = v0 - ray.origin;
    const V3 B = v1 - ray.origin;
    const V3 C = v2 - ray.origin;
    const float Ax = A[rh.kx] - rh.S.x * A[rh.kz];
    const float Ay = A[rh.ky] - rh.S.y * A[rh.kz];
    const float Bx = B[rh.kx] - rh.S.x * B[rh.kz];
    const float By = B[rh.ky] - rh.S.y * B[rh.kz];
    const float Cx = C[rh.kx] - rh.S.x * C[rh.kz];
    const float Cy = C[rh.ky] - rh.S.y * C[rh.kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    const float ldeps = float(std::numeric_limits<long double>::epsilon());
    if (std::fabs(U) < ldeps || std::fabs(V) < ldeps || std::fabs(W) < ldeps) {
        double CxBy = double(Cx) * double(By);
        double CyBx = double(Cy) * double(Bx);
        U = CxBy - CyBx;
        double AxCy = double(Ax) * double(Cy);
        double AyCx = double(Ay) * double(Cx);
        V = AxCy - AyCx;
        double BxAy = double(Bx) * double(Ay);
        double ByAx = double(By) * double(Ax);
        W = BxAy - ByAx;
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return HitRecord();
    float det = U + V + W;
    if (det == 0.0f)
        return HitRecord();
    const float Az = rh.S.z * A[rh.kz];
    const float Bz = rh.S.z * B[rh.kz];
    const float Cz = rh.S.z * C[rh.kz];
    const float T = U * Az + V * Bz + W * Cz;
    uint32_t detSign = uint32_t(std::signbit(det)) << 31;
    if (xorf(T, detSign) < amin * xorf(det, detSign) || xorf(T, detSign) > amax * xorf(det, detSign))
        return HitRecord();
    const float invDet = 1.0f / det;
    const float a = invDet * T;

    HitRecord hr;
    hr.haveHit = true;
    hr.a = a;
    if (!fullInfo) {
        *v0v1 = v1 - v0;
        *v0v2 = v2 - v0;
        return hr;
    }

    bool backfacing = (det < 0.0f);
    const V3 bary = invDet * V3 { U, V, W };
    V3 hitpos = ray.at(a);
    const wpt_tri_attr& at = c.sc->tri_attr[prim];
    const bool transform = (g.flags & WPT_TRI_TRANSFORM) != 0;
    const float* N = c.sc->instances[g.instance].N;
    V3 hitnrm = bary.x * v3(at.n0) + bary.y * v3(at.n1) + bary.z * v3(at.n2);
    if (transform)
        hitnrm = mat3_mul(N, hitnrm);
    if (animate)
        hitnrm = mat3_mul(animationN, hitnrm);
    hitnrm = normalize(hitnrm);
    if (backfacing)
        hitnrm = -hitnrm;
    V2 hittc = V2 { 0.0f, 0.0f };
    if (g.flags & WPT_TRI_HAVE_TEXCOORDS) {
        V2 tc0 = V2 { at.tc0[0], at.tc0[1] }, tc1 = V2 { at.tc1[0], at.tc1[1] }, tc2 = V2 { at.tc2[0], at.tc2[1] };
        hittc = bary.x * tc0 + bary.y * tc1 + bary.z * tc2;
    }
    V3 hittan = v3(0.0f);
    if (g.flags & WPT_TRI_HAVE_TANGENTS) {
        hittan = bary.x * v3(at.t0) + bary.y * v3(at.t1) + bary.z * v3(at.t2);
        if (dot(hittan, hittan) > 0.0f) {
            if (transform)
                hittan = mat3_mul(N, hittan);
            if (animate)
                hittan = mat3_mul(animationN, hittan);
            hittan = normalize(hittan - dot(hitnrm, hittan) * hitnrm);
        }
    }
    hr.position = hitpos;
    hr.normal = hitnrm;
    hr.tangent = hittan;
    hr.texcoords = hittc;
    hr.backside = backfacing;
    hr.prim = prim;
    return hr;
}

/* hitable_triangle.hpp:405-423 */
inline float trianglePdfValue(Ctx& c, uint32_t prim, V3 origin, V3 direction)
{
    c.cnt.pdf_tests++;
    float value = 0.0f;
    Ray ray { origin, direction, c.time, v4(0.0f) };
    V3 v0v1, v0v2;
    HitRecord hr = triangleHit(c, prim, ray, RayHelper(ray), 0.0f, k_maxval, false, &v0v1, &v0v2);
    if (hr.haveHit) {
        V3 edgeCross = cross(v0v1, v0v2);
        float edgeCrossLength = std::sqrt(dot(edgeCross, edgeCross));
        V3 faceNormal = edgeCross / edgeCrossLength;
        float faceArea = 0.5f * edgeCrossLength;
        float cosine = std::fabs(dot(faceNormal, -direction));
        float distance_squared = hr.a * hr.a;
        value = distance_squared / (cosine * faceArea);
    }
    return value;
}

/* hitable_triangle.hpp:425-443 */
inline V3 triangleDirection(const Ctx& c, uint32_t hotspot, V3 origin, Prng& prng)
{
    const wpt_hotspot& h = c.sc->hotspots[hotspot];
    V3 bary = inTriangle(prng.in01x2());
    V3 p = bary.x * v3(h.p0) + bary.y * v3(h.p1) + bary.z * v3(h.p2);
    if (h.transform)
        p = mat4_mul_point(h.M, p);
    if (h.animation >= 0) {
        float animationM[16];
        wptanim::toMat4(animationAt(c, h.animation, c.time), animationM);
        p = animatePoint(animationM, p);
    }
    return normalize(p - origin);
}

/* ---- hitable_sphere.hpp ----
 * A sphere's HitRecord carries prim = PRIM_SPHERE | index (stands for the Hitable pointer). */
constexpr uint32_t PRIM_SPHERE = 0x80000000u;
constexpr float k_cosOrthoAngleTolerance = 0.0003f; /* constants.hpp:41 */

/* hitable_sphere.hpp:42-75 */
inline HitRecord sphereHitRecord(const Ray& ray, float a, V3 center, const float* rotation, uint32_t prim)
{
    HitRecord hr;
    V3 p = ray.at(a);
    V3 n = normalize(p - center);
    V3 rn = quat_rotate(rotation, n);
    float alpha = m_atan2(rn.x, rn.z);
    float beta = m_asin(clamp_(rn.y, -1.0f, +1.0f));
    float u = 0.5f * k_inv_pi * (alpha + k_pi);
    float v = k_inv_pi * (beta + 0.5f * k_pi);
    V3 t = V3 { m_cos(alpha), 0.0f, -m_sin(alpha) };
    if (std::fabs(dot(rn, t)) >= k_cosOrthoAngleTolerance)
        t = v3(0.0f);
    bool backside = false;
    if (dot(n, -ray.direction) < 0.0f) {
        backside = true;
        n = -n;
    }
    hr.haveHit = true;
    hr.a = a;
    hr.position = p;
    hr.normal = n;
    hr.tangent = t;
    hr.texcoords = V2 { u, v };
    hr.backside = backside;
    hr.prim = prim;
    return hr;
}

inline float max3(const float* s) /* gvm.hpp:1284-1291 */
{
    float r = s[0];
    for (int i = 1; i < 3; i++)
        if (s[i] > r)
            r = s[i];
    return r;
}

/* hitable_sphere.hpp:104-147; `c` gives the animations and the time of the path (the AnimationCache) */
inline HitRecord sphereHit(const Ctx& c, const wpt_sphere& sp, uint32_t index, const Ray& ray, float amin, float amax)
{
    V3 center = v3(sp.center);
    float radius = sp.radius;
    float rot[4] = { sp.rotation[0], sp.rotation[1], sp.rotation[2], sp.rotation[3] };
    if (sp.animation >= 0) {
        const wptanim::Trs T = animationAt(c, sp.animation, c.time);
        center = center + v3(T.t);
        radius *= max3(T.s);
        const float x = sp.rotation[0], y = sp.rotation[1], z = sp.rotation[2], w = sp.rotation[3];
        rot[0] = w * T.q[0] + x * T.q[3] + y * T.q[2] - z * T.q[1];
        rot[1] = w * T.q[1] + y * T.q[3] + z * T.q[0] - x * T.q[2];
        rot[2] = w * T.q[2] + z * T.q[3] + x * T.q[1] - y * T.q[0];
        rot[3] = w * T.q[3] - x * T.q[0] - y * T.q[1] - z * T.q[2];
    }
    V3 oc = ray.origin - center;
    float aq = -dot(oc, ray.direction);
    V3 tmp = oc - dot(oc, ray.direction) * ray.direction;
    float discriminant = radius * radius - dot(tmp, tmp);
    HitRecord hr;
    if (discriminant > 0.0f) {
        float a1, a2;
        if (aq < 0.0f) {
            a2 = aq - std::sqrt(discriminant);
            a1 = 2.0f * aq - a2;
        } else {
            a1 = aq + std::sqrt(discriminant);
            a2 = 2.0f * aq - a1;
        }
        if (a2 > amin && a2 < amax)
            hr = sphereHitRecord(ray, a2, center, rot, PRIM_SPHERE | index);
        else if (a1 > amin && a1 < amax)
            hr = sphereHitRecord(ray, a1, center, rot, PRIM_SPHERE | index);
    }
    return hr;
}

/* hitable_sphere.hpp:149-186 */
inline float spherePdfValue(Ctx& c, uint32_t index, V3 origin, V3 direction)
{
    c.cnt.pdf_tests++;
    const wpt_sphere& sp = c.sc->spheres[index];
    V3 center = v3(sp.center);
    float radius = sp.radius;
    if (sp.animation >= 0) { /* :161-166: the whole transformation, unlike hit() and direction() */
        const wptanim::Trs T = animationAt(c, sp.animation, c.time);
        float moved[3];
        wptanim::applyTrs(T, sp.center, moved);
        center = v3(moved);
        radius *= max3(T.s);
    }
    float value = 0.0f;
    V3 cmo = center - origin;
    float distanceSquared = dot(cmo, cmo);
    float radiusSquared = radius * radius;
    if (distanceSquared <= radiusSquared) {
        value = 0.25f * k_inv_pi;
    } else {
        HitRecord hr = sphereHit(c, sp, index, Ray { origin, direction, c.time, v4(0.0f) }, 0.0f, k_maxval);
        if (hr.haveHit) {
            float discriminant = 1.0f - radiusSquared / distanceSquared;
            float cosThetaMax = (discriminant > 0.0f ? std::sqrt(discriminant) : 0.0f);
            float solidAngle = 2.0f * k_pi * (1.0f - cosThetaMax);
            value = 1.0f / solidAngle;
        }
    }
    return value;
}

/* hitable_sphere.hpp:188-219 */
inline V3 sphereDirection(const Ctx& c, uint32_t index, V3 origin, Prng& prng)
{
    const wpt_sphere& sp = c.sc->spheres[index];
    V3 center = v3(sp.center);
    float radius = sp.radius;
    if (sp.animation >= 0) { /* :196-203 */
        const wptanim::Trs T = animationAt(c, sp.animation, c.time);
        center = center + v3(T.t);
        radius *= max3(T.s);
    }
    V3 dir;
    V3 cmo = center - origin;
    float distanceSquared = dot(cmo, cmo);
    float radiusSquared = radius * radius;
    if (distanceSquared <= radiusSquared) {
        dir = onUnitSphere(prng.in01x2());
    } else {
        float discriminant = 1.0f - radiusSquared / distanceSquared;
        float cosThetaMax = (discriminant > 0.0f ? std::sqrt(discriminant) : 0.0f);
        dir = toSphere(normalize(cmo), cosThetaMax, prng.in01x2());
    }
    return dir;
}

/* the virtual Hitable::pdfValue / direction / identity of hot spot i, and Hitable::material() */
inline float hotSpotPdfValue(Ctx& c, size_t i, V3 origin, V3 direction)
{
    const wpt_hotspot& h = c.sc->hotspots[i];
    return h.kind == WPT_HOTSPOT_SPHERE ? spherePdfValue(c, h.prim, origin, direction) : trianglePdfValue(c, h.prim, origin, direction);
}
inline V3 hotSpotDirection(const Ctx& c, size_t i, V3 origin, Prng& prng)
{
    const wpt_hotspot& h = c.sc->hotspots[i];
    return h.kind == WPT_HOTSPOT_SPHERE ? sphereDirection(c, h.prim, origin, prng) : triangleDirection(c, uint32_t(i), origin, prng);
}
inline uint32_t hotSpotPrim(const Ctx& c, size_t i)
{
    const wpt_hotspot& h = c.sc->hotspots[i];
    return h.kind == WPT_HOTSPOT_SPHERE ? (PRIM_SPHERE | h.prim) : h.prim;
}
inline uint32_t materialOfPrim(const Ctx& c, uint32_t prim)
{
    return (prim & PRIM_SPHERE) ? c.sc->spheres[prim & ~PRIM_SPHERE].material : c.sc->tri_geom[prim].material;
}

/* ---- bvh.hpp:277-311 ---- */
/* leafHit(kind, index, amin, amax) stands for the virtual `node.hitable->hit(...)` */
template<typename LeafHit>
inline HitRecord bvhTraverse(const wpt_bvh_node* nodes, wpt_counters& cnt, const Ray& ray, const RayHelper& rh,
        float amin, float amax, LeafHit&& leafHit)
{
    cnt.rays++;
    HitRecord hr;
    size_t toVisitOffset = 0;
    size_t currentNodeIndex = 0;
    uint32_t nodesToVisit[128];
    for (;;) {
        const wpt_bvh_node& node = nodes[currentNodeIndex];
        cnt.node_visits++;
        if (aabbMayHit(node.lo, node.hi, ray, amin, amax, rh.invDirection)) {
            if (node.kind != WPT_NODE_INNER) {
                if (node.kind != WPT_NODE_EMPTY) {
                    cnt.leaf_tests++;
                    HitRecord cur = leafHit(node.kind, node.link, amin, amax);
                    if (cur.haveHit) {
                        hr = cur;
                        amax = hr.a;
                    }
                }
                if (toVisitOffset == 0)
                    break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            } else {
                nodesToVisit[toVisitOffset++] = node.link;
                currentNodeIndex++;
            }
        } else {
            if (toVisitOffset == 0)
                break;
            currentNodeIndex = nodesToVisit[--toVisitOffset];
        }
    }
    return hr;
}

inline HitRecord bvhHit(Ctx& c, const Ray& ray, const RayHelper& rh, float amin, float amax)
{
    return bvhTraverse(c.sc->nodes, c.cnt, ray, rh, amin, amax, [&](uint32_t kind, uint32_t index, float lo, float hi) {
        if (kind == WPT_NODE_SPHERE)
            return sphereHit(c, c.sc->spheres[index], index, ray, lo, hi);
        return triangleHit(c, index, ray, rh, lo, hi, true, nullptr, nullptr);
    });
}

/* ---- textures: texture.hpp:160-246, texture_image.hpp:85-212, color.hpp:275-294 ---- */
inline float srgb_to_rgb_helper(float x)
{
    return (x <= 0.04045f ? (x * (1.0f / 12.92f)) : m_pow((x + 0.055f) * (1.0f / 1.055f), 2.4f));
}

V4 textureValue(const Ctx& c, int tex, V2 texcoords);

inline V4 imageTexel(const Ctx& c, const wpt_texture& t, size_t x, size_t y)
{
    if (x >= t.width)
        x = t.width - 1;
    if (y >= t.height)
        y = t.height - 1;
    const uint8_t* base = c.sc->texels + t.texel_offset;
    size_t idx = (y * size_t(t.width) + x) * t.comps;
    float d[4];
    bool lin = t.linearize_srgb != 0;
    for (uint32_t k = 0; k < t.comps; k++) {
        float raw;
        if (t.texel_type == WPT_TEXEL_U8)
            raw = base[idx + k] / 255.0f;
        else if (t.texel_type == WPT_TEXEL_U16)
            raw = reinterpret_cast<const uint16_t*>(base)[idx + k] / 65535.0f;
        else
            raw = reinterpret_cast<const float*>(base)[idx + k];
        /* colour channels are linearized, alpha (2nd of 2, 4th of 4) never */
        bool isAlpha = (t.comps == 2 && k == 1) || (t.comps == 4 && k == 3);
        d[k] = (lin && !isAlpha) ? srgb_to_rgb_helper(raw) : raw;
    }
    if (t.comps == 3)
        return V4 { d[0], d[1], d[2], 1.0f };
    if (t.comps == 4)
        return V4 { d[0], d[1], d[2], d[3] };
    if (t.comps == 1)
        return V4 { d[0], d[0], d[0], 1.0f };
    return V4 { d[0], d[0], d[0], d[1] };
}

V4 textureValue(const Ctx& c, int tex, V2 texcoords)
{
    const wpt_texture& t = c.sc->textures[tex];
    switch (t.type) {
    case WPT_TEX_CONSTANT:
        return v4(t.a);
    case WPT_TEX_CHECKER: {
        int row = texcoords.y * int(t.height);
        int col = texcoords.x * int(t.width);
        return (row % 2 == col % 2 ? v4(t.a) : v4(t.b));
    }
    case WPT_TEX_TRANSFORMER: {
        V2 cf = V2 { t.coord_factor[0], t.coord_factor[1] }, co = V2 { t.coord_offset[0], t.coord_offset[1] };
        V4 val = textureValue(c, t.child, cf * texcoords + co);
        return v4(t.a) * val + v4(t.b);
    }
    default: {
        V2 cf = V2 { t.coord_factor[0], t.coord_factor[1] }, co = V2 { t.coord_offset[0], t.coord_offset[1] };
        V2 tc = cf * texcoords + co;
        V2 uv = V2 { tc.x - std::floor(tc.x), tc.y - std::floor(tc.y) };
        float uvs = fmax_(0.0f, (uv.x * t.width) - 0.5f);
        float uvt = fmax_(0.0f, (uv.y * t.height) - 0.5f);
        size_t x0 = uvs;
        size_t y0 = uvt;
        size_t x1 = x0 + 1;
        size_t y1 = y0 + 1;
        float alpha = uvs - x0;
        float beta = uvt - y0;
        V4 v00 = imageTexel(c, t, x0, y0);
        V4 v10 = imageTexel(c, t, x1, y0);
        V4 v01 = imageTexel(c, t, x0, y1);
        V4 v11 = imageTexel(c, t, x1, y1);
        V4 a = mix4(v00, v10, alpha);
        V4 b = mix4(v01, v11, alpha);
        V4 val = mix4(a, b, beta);
        return v4(t.a) * val + v4(t.b);
    }
    }
}

/* ---- envmap.hpp:55-247 ---- */
inline V2 envM(V3 d)
{
    float lat = m_asin(clamp_(d.y, -1.0f, +1.0f));
    float lon = m_atan2(-d.x, d.z);
    float r = m_sin(0.5f * (k_pi_2 - lat));
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
    return V2 { 0.5f * (u + 1.0f), 0.5f * (v + 1.0f) };
}
inline V3 envInvM(V2 uv)
{
    float u = 2.0f * uv.x - 1.0f;
    float v = 2.0f * uv.y - 1.0f;
    float r, alpha;
    if (u * u > v * v) {
        r = u;
        alpha = k_pi_4 * v / u;
    } else {
        r = v;
        if (std::fabs(v) > 0.0f)
            alpha = k_pi_2 - k_pi_4 * u / v;
        else
            alpha = 0.0f;
    }
    float lat = k_pi_2 - 2.0f * m_asin(r);
    float lon = alpha + k_pi_2;
    V3 d = V3 { -m_cos(lat) * m_sin(lon), m_sin(lat), m_cos(lat) * m_cos(lon) };
    return normalize(d);
}
inline V4 envL(const Ctx& c, V3 direction)
{
    const wpt_envmap& e = c.sc->envmap;
    if (e.type == WPT_ENV_CUBE) {
        /* EnvironmentMapCube::L (envmap.hpp:265-284) */
        float ax = std::fabs(direction.x);
        float ay = std::fabs(direction.y);
        float az = std::fabs(direction.z);
        int cubeside;
        float u, v;
        if (ax > ay && ax > az) {
            u = 0.5f * (direction.z / -direction.x + 1.0f);
            v = 0.5f * (direction.y / ax + 1.0f);
            cubeside = 0 + (std::signbit(direction.x) ? 1 : 0);
        } else if (ay > az) {
            u = 0.5f * (direction.x / ay + 1.0f);
            v = 0.5f * (direction.z / -direction.y + 1.0f);
            cubeside = 2 + (std::signbit(direction.y) ? 1 : 0);
        } else {
            u = 0.5f * (direction.x / direction.z + 1.0f);
            v = 0.5f * (direction.y / az + 1.0f);
            cubeside = 4 + (std::signbit(direction.z) ? 1 : 0);
        }
        return textureValue(c, e.cube_tex[cubeside], V2 { u, v });
    }
    float y = m_asin(clamp_(direction.y, -1.0f, 1.0f));
    float x = m_atan2(-direction.x, direction.z);
    if (e.compat == WPT_ENV_COMPAT_MITSUBA) {
        x -= k_pi;
        if (x < 0.0f)
            x += 2.0f * k_pi;
    }
    x *= 0.5f * k_inv_pi;
    y = y * k_inv_pi + 0.5f;
    return textureValue(c, e.tex, V2 { x, y });
}
inline float envP(const Ctx& c, V3 direction)
{
    const wpt_envmap& e = c.sc->envmap;
    int N = e.N;
    V2 uv = envM(direction);
    int x = uv.x * N;
    int y = uv.y * N;
    if (x >= N)
        x = N - 1;
    if (y >= N)
        y = N - 1;
    float q = e.M[y * N + x];
    float invBinSizeOnSphere = (N * N) * 0.25f * k_inv_pi;
    return q * invBinSizeOnSphere;
}
inline V3 envD(const Ctx& c, Prng& prng)
{
    const wpt_envmap& e = c.sc->envmap;
    int N = e.N;
    float r = prng.in01();
    int a = 0;
    int b = N * N - 1;
    while (b > a + 1) {
        int cc = (a + b) / 2;
        if (e.Mcs[cc] < r)
            a = cc;
        else
            b = cc;
    }
    int bin = (e.Mcs[a] >= r ? a : b);
    bin = e.Ms[bin];
    int x = bin % N;
    int y = bin / N;
    float u = (x + prng.in01()) / N;
    float v = (y + prng.in01()) / N;
    return envInvM(V2 { u, v });
}

/* ---- materials ---- */
enum { ScatterNone = 0, ScatterExplicit = 1, ScatterRandom = 2 };
struct ScatterRecord {
    int type = ScatterNone;
    V3 direction { 0, 0, 0 };
    V4 attenuation { 0, 0, 0, 0 };
    float pdf = 0.0f;
    V4 refractiveIndex { 0, 0, 0, 0 };
};
inline ScatterRecord srExplicit(V3 dir, V4 att, V4 ri)
{
    ScatterRecord s;
    s.type = ScatterExplicit;
    s.direction = dir;
    s.attenuation = att;
    s.pdf = 0.0f;
    s.refractiveIndex = ri;
    return s;
}
inline ScatterRecord srRandom(V3 dir, V4 att, float p, V4 ri)
{
    ScatterRecord s;
    s.type = ScatterRandom;
    s.direction = dir;
    s.attenuation = att;
    s.pdf = p;
    s.refractiveIndex = ri;
    return s;
}

/* material.hpp:195-228 */
inline V3 normalAt(const Ctx& c, const wpt_material& m, const HitRecord& hit)
{
    V3 n = hit.normal;
    if (m.normal_tex >= 0) {
        n = rgb(textureValue(c, m.normal_tex, hit.texcoords));
        n = 2.0f * n - v3(1.0f);
        n = normalize(TangentSpace(hit.normal, hit.tangent).toWorldSpace(n));
    }
    return n;
}
inline TangentSpace tangentSpaceAt(const Ctx& c, const wpt_material& m, const HitRecord& hit)
{
    TangentSpace ts;
    if (dot(hit.tangent, hit.tangent) > k_epsilon) {
        ts = TangentSpace(hit.normal, hit.tangent);
        if (m.normal_tex >= 0) {
            V3 n = rgb(textureValue(c, m.normal_tex, hit.texcoords));
            n = 2.0f * n - v3(1.0f);
            n = normalize(ts.toWorldSpace(n));
            V3 t = normalize(hit.tangent - dot(n, hit.tangent) * n);
            ts = TangentSpace(n, t);
        }
    } else {
        ts = TangentSpace(hit.normal);
    }
    return ts;
}

/* material_lambertian.hpp:53-102 */
inline V4 lambertianAlbedoAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    V4 a = m.tex[0] >= 0 ? textureValue(c, m.tex[0], tc) : v4(m.v[0]);
    if (!(m.flags & WPT_MATF_HAVE_NIR))
        a.w = average3(rgb(a));
    return a;
}

/* material_ggx.hpp:89-171 */
inline float ggxLambda(V3 tsv, V2 roughness)
{
    V2 a2 = roughness * roughness;
    V3 tsv2 = tsv * tsv;
    float discriminant = 1.0f + (a2.x * tsv2.x + a2.y * tsv2.y) / tsv2.z;
    return 0.5f * (-1.0f + std::sqrt(discriminant));
}
inline float ggxG1(V3 tsv, V2 r) { return 1.0f / (1.0f + ggxLambda(tsv, r)); }
inline float ggxG2(V3 tsV, V3 tsL, V2 r) { return 1.0f / (1.0f + ggxLambda(tsV, r) + ggxLambda(tsL, r)); }
inline float ggxD(V3 tsH, V2 roughness)
{
    V2 a2 = roughness * roughness;
    V3 tsh2 = tsH * tsH;
    float t = tsh2.x / a2.x + tsh2.y / a2.y + tsh2.z;
    float D = 1.0f / (k_pi * roughness.x * roughness.y * t * t);
    return D;
}
inline float ggxDV(V3 tsH, V3 tsV, float dotVH, V2 roughness)
{
    float dotVZ = tsV.z;
    float DV = ggxG1(tsV, roughness) * dotVH * ggxD(tsH, roughness) / dotVZ;
    return DV;
}
inline V3 ggxSampleVNDF(V3 Ve, V2 roughness, Prng& prng)
{
    float alpha_x = roughness.x;
    float alpha_y = roughness.y;
    float U1 = prng.in01();
    float U2 = prng.in01();
    V3 Vh = normalize(V3 { alpha_x * Ve.x, alpha_y * Ve.y, Ve.z });
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0f ? V3 { -Vh.y, Vh.x, 0.0f } * (1.0f / std::sqrt(lensq)) : V3 { 1.0f, 0.0f, 0.0f };
    V3 T2 = cross(Vh, T1);
    float r = std::sqrt(U1);
    float phi = 2.0f * k_pi * U2;
    float t1 = r * m_cos(phi);
    float t2 = r * m_sin(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * std::sqrt(1.0f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + std::sqrt(fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    V3 Ne = normalize(V3 { alpha_x * Nh.x, alpha_y * Nh.y, fmax_(0.0f, Nh.z) });
    return Ne;
}
inline V4 ggxAlbedoAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    return m.tex[0] >= 0 ? textureValue(c, m.tex[0], tc) : v4(m.v[0]);
}
inline V2 ggxRoughnessAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    V2 r = V2 { m.f[0], m.f[1] };
    if (m.tex[1] >= 0) {
        V4 t = textureValue(c, m.tex[1], tc);
        r = V2 { t.x, t.y };
    }
    return r;
}

/* material_modphong.hpp:136-239 */
inline float mpOpacityAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    if (m.tex[3] >= 0)
        return textureValue(c, m.tex[3], tc).x;
    if (m.flags & WPT_MATF_DIFFUSE_TEX_HAS_ALPHA)
        return textureValue(c, m.tex[0], tc).w;
    return m.f[1];
}
inline V4 mpDiffuseAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    V4 kd = m.tex[0] >= 0 ? textureValue(c, m.tex[0], tc) : v4(m.v[0]);
    if (!(m.flags & WPT_MATF_HAVE_NIR))
        kd.w = average3(rgb(kd));
    return kd;
}
inline V4 mpSpecularAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    V4 ks = m.tex[1] >= 0 ? textureValue(c, m.tex[1], tc) : v4(m.v[1]);
    if (m.flags & WPT_MATF_SPECULAR_TEX_HAS_ALPHA) {
        V3 mx = mix3(rgb(ks), rgb(v4(m.v[1])), ks.w);
        ks = V4 { mx.x, mx.y, mx.z, ks.w };
    }
    if (!(m.flags & WPT_MATF_HAVE_NIR))
        ks.w = average3(rgb(ks));
    return ks;
}
inline float mpShininessAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    float s = m.f[0];
    if (m.tex[2] >= 0)
        s *= textureValue(c, m.tex[2], tc).x;
    return s;
}
inline V4 mpEmissiveAt(const Ctx& c, const wpt_material& m, V2 tc)
{
    V4 ke = m.tex[4] >= 0 ? textureValue(c, m.tex[4], tc) : v4(m.v[3]);
    if (!(m.flags & WPT_MATF_HAVE_NIR))
        ke.w = average3(rgb(ke));
    return ke;
}
inline V4 mpAttenuation(V3 n, V3 v, V3 l, V4 kd, V4 ks, float s, float cosTheta)
{
    V3 r = reflect(-l, n);
    float cosRV = fmax_(dot(r, v), 0.0f);
    return (kd + 0.5f * ks * (s + 2.0f) * m_pow(cosRV, s)) * k_inv_pi * fmin_(cosTheta, 1.0f);
}
inline float mpSpecularProbability(V4 kd, V4 ks)
{
    float skd = kd.x + kd.y + kd.z + kd.w;
    float sks = ks.x + ks.y + ks.z + ks.w;
    float sum = skd + sks + 1e-4f;
    float p = sks / sum;
    return clamp_(p, 0.1f, 0.9f);
}
inline float mpPdfValue(V3 n, V3 v, V3 l, float s, float cosTheta, float specProb)
{
    float diffusePdfValue = cosTheta * k_inv_pi;
    V3 r = reflect(-v, n);
    float cosRL = fmax_(dot(r, l), 0.0f);
    float specularPdfValue = 0.5f * k_inv_pi * (s + 1.0f) * m_pow(cosRL, s);
    return mix_(diffusePdfValue, specularPdfValue, specProb);
}

ScatterRecord materialScatter(Ctx& c, uint32_t mat, const Ray& ray, const HitRecord& hit, Prng& prng);
ScatterRecord materialScatterToDirection(const Ctx& c, uint32_t mat, const Ray& ray, const HitRecord& hit, V3 direction);
V4 materialEmitted(const Ctx& c, uint32_t mat, const Ray& ray, const HitRecord& hit);

inline HitRecord toFrontSide(const HitRecord& hit) /* material.hpp:279-282 */
{
    HitRecord h = hit;
    h.backside = false;
    return h;
}

ScatterRecord materialScatter(Ctx& c, uint32_t mat, const Ray& ray, const HitRecord& hit, Prng& prng)
{
    const wpt_material& m = c.sc->materials[mat];
    switch (m.type) {
    case WPT_MAT_LAMBERTIAN: { /* material_lambertian.hpp:61-84 */
        if (hit.backside)
            return ScatterRecord();
        V3 cosineDir = cosineDirection(prng.in01x2());
        float cosTheta = cosineDir.z;
        TangentSpace ts = tangentSpaceAt(c, m, hit);
        V3 dir = normalize(ts.toWorldSpace(cosineDir));
        float p = cosTheta * k_inv_pi;
        V4 att = lambertianAlbedoAt(c, m, hit.texcoords) * p;
        return srRandom(dir, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_MIRROR: { /* material_mirror.hpp:53-62 */
        if (hit.backside)
            return ScatterRecord();
        V3 reflected = reflect(ray.direction, normalAt(c, m, hit));
        V4 att = m.tex[0] >= 0 ? textureValue(c, m.tex[0], hit.texcoords) : v4(m.v[0]);
        if (!(m.flags & WPT_MATF_HAVE_NIR))
            att.w = average3(rgb(att));
        return srExplicit(normalize(reflected), att, ray.refractiveIndex);
    }
    case WPT_MAT_GGX: { /* material_ggx.hpp:173-225 */
        if (hit.backside)
            return ScatterRecord();
        V3 view = -ray.direction;
        V2 roughness = ggxRoughnessAt(c, m, hit.texcoords);
        TangentSpace ts = tangentSpaceAt(c, m, hit);
        V3 tsV = ts.toTangentSpace(view);
        V3 tsH = ggxSampleVNDF(tsV, roughness, prng);
        V3 tsL = reflect(-tsV, tsH);
        V3 light = ts.toWorldSpace(tsL);
        float l = dot(light, light);
        if (l < k_epsilon)
            return ScatterRecord();
        V3 dir = light / std::sqrt(l);
        float dotVH = dot(tsV, tsH);
        float p = ggxDV(tsH, tsV, dotVH, roughness) / (4.0f * dotVH);
        if (!std::isfinite(p) || p < 0.0f)
            return ScatterRecord();
        V4 att = v4(0.0f);
        float dotNL = dot(ts.normal, light);
        float dotNV = dot(ts.normal, view);
        if (dotNL > 0.0f && dotNV > 0.0f) {
            float Dval = ggxD(tsH, roughness);
            V4 albedo = ggxAlbedoAt(c, m, hit.texcoords);
            V4 Fval = fresnelSchlick(albedo, dotVH);
            float Gval = ggxG2(tsV, tsL, roughness);
            att = Dval * Fval * Gval / (4.0f * dotNV);
        }
        return srRandom(dir, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_GLASS: { /* material_glass.hpp:91-152 */
        V4 att = v4(1.0f);
        V4 ourRI = v4(m.v[1]);
        V4 theirRI = v4(m.v[2]);
        int riIndex = 0;
        if (m.flags & WPT_MATF_CHROMATIC_DISPERSION) {
            riIndex = prng.in01() * 4;
            att = v4(0.0f);
            att.at(riIndex) = 4.0f;
        }
        if (hit.backside) {
            V4 tmp = ourRI;
            ourRI = theirRI;
            theirRI = tmp;
            float distInVolume = hit.a;
            V4 e = -v4(m.v[0]) * distInVolume;
            att = att * V4 { m_exp(e.x), m_exp(e.y), m_exp(e.z), m_exp(e.w) };
        }
        V3 n = normalAt(c, m, hit);
        V3 refracted = refract(ray.direction, n, theirRI[riIndex] / ourRI[riIndex]);
        bool doReflection = true;
        if (dot(refracted, refracted) > 0.0f) {
            float cosIncident = dot(-ray.direction, n);
            float cosTransmitted = -dot(refracted, n);
            float fresnel = fresnelUnpolarized(cosIncident, cosTransmitted, theirRI[riIndex], ourRI[riIndex]);
            doReflection = prng.in01() < fresnel;
        }
        if (doReflection) {
            V3 reflected = reflect(ray.direction, n);
            return srExplicit(normalize(reflected), att, theirRI);
        } else {
            return srExplicit(normalize(refracted), att, ourRI);
        }
    }
    case WPT_MAT_MODPHONG: { /* material_modphong.hpp:241-308 */
        float opa = mpOpacityAt(c, m, hit.texcoords);
        bool transparent = (opa < 1.0f && opa < prng.in01());
        if (transparent) {
            float ourRI = m.f[2];
            float theirRI = 1.0f;
            if (hit.backside) {
                float tmp = ourRI;
                ourRI = theirRI;
                theirRI = tmp;
            }
            V3 n = normalize(normalAt(c, m, hit));
            V3 refracted = refract(ray.direction, n, theirRI / ourRI);
            float l = dot(refracted, refracted);
            if (l < k_epsilon)
                return ScatterRecord();
            refracted = refracted / std::sqrt(l);
            V4 att = v4(m.v[2]);
            if (!(m.flags & WPT_MATF_HAVE_NIR))
                att.w = average3(rgb(att));
            return srExplicit(refracted, att, v4(ourRI));
        }
        if (hit.backside)
            return ScatterRecord();
        V4 kd = mpDiffuseAt(c, m, hit.texcoords);
        V4 ks = mpSpecularAt(c, m, hit.texcoords);
        float s = mpShininessAt(c, m, hit.texcoords);
        float specProb = mpSpecularProbability(kd, ks);
        V3 dir, n;
        float cosTheta;
        if (prng.in01() < specProb) {
            float r1 = prng.in01();
            float r2 = prng.in01();
            float cosThetaSpec = m_pow(1.0f - r1, 1.0f / (1.0f + s));
            float discriminant = fmax_(1.0f - cosThetaSpec * cosThetaSpec, 0.0f);
            float sinThetaSpec = std::sqrt(discriminant);
            float phi = 2.0f * k_pi * r2;
            float x = m_cos(phi) * sinThetaSpec;
            float y = m_sin(phi) * sinThetaSpec;
            float z = cosThetaSpec;
            n = normalAt(c, m, hit);
            TangentSpace specTS = TangentSpace(reflect(ray.direction, n));
            dir = normalize(specTS.toWorldSpace(V3 { x, y, z }));
            cosTheta = fmax_(dot(dir, n), 0.0f);
        } else {
            TangentSpace ts = tangentSpaceAt(c, m, hit);
            n = ts.normal;
            V3 cosineDir = cosineDirection(prng.in01x2());
            cosTheta = cosineDir.z;
            dir = normalize(ts.toWorldSpace(cosineDir));
        }
        V4 att = mpAttenuation(n, -ray.direction, dir, kd, ks, s, cosTheta);
        float p = mpPdfValue(n, -ray.direction, dir, s, cosTheta, specProb);
        return srRandom(dir, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_RGL: { /* material_rgl.hpp:59-80 */
        if (hit.backside)
            return ScatterRecord();
        TangentSpace ts = tangentSpaceAt(c, m, hit);
        V3 wi = ts.toTangentSpace(-ray.direction);
        V2 u = prng.in01x2();
        wptrgl::V3 pwo;
        float p;
        wptrgl::V3 a = wptrgl::rglSample<OracleMath>(c.sc->rgl_brdfs[m.tex[0]], c.sc->rgl_data, wptrgl::V2 { u.x, u.y },
                wptrgl::V3 { wi.x, wi.y, wi.z }, pwo, p);
        V3 attenuation = V3 { a.x, a.y, a.z };
        V3 wo = V3 { pwo.x, pwo.y, pwo.z };
        if (dot(wo, wo) <= 0.0f)
            return ScatterRecord();
        V4 att = V4 { attenuation.x, attenuation.y, attenuation.z, average3(attenuation) };
        att = att * p; /* sample() returns f * cos / pdf; the division is undone here */
        V3 dir = normalize(ts.toWorldSpace(wo));
        return srRandom(dir, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_TWOSIDED: /* material.hpp:290-296 */
        return hit.backside ? materialScatter(c, uint32_t(m.tex[1]), ray, toFrontSide(hit), prng)
                            : materialScatter(c, uint32_t(m.tex[0]), ray, hit, prng);
    default: /* material.hpp:158-164: lights and the base class do not scatter */
        return ScatterRecord();
    }
}

ScatterRecord materialScatterToDirection(const Ctx& c, uint32_t mat, const Ray& ray, const HitRecord& hit, V3 direction)
{
    const wpt_material& m = c.sc->materials[mat];
    switch (m.type) {
    case WPT_MAT_LAMBERTIAN: { /* material_lambertian.hpp:86-102 */
        V4 att = v4(0.0f);
        float p = 0.0f;
        float cosTheta = dot(normalAt(c, m, hit), direction);
        if (cosTheta > 0.0f) {
            p = cosTheta * k_inv_pi;
            att = lambertianAlbedoAt(c, m, hit.texcoords) * p;
        }
        return srRandom(direction, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_GGX: { /* material_ggx.hpp:227-257 */
        V4 att = v4(0.0f);
        float p = 0.0f;
        V3 view = -ray.direction;
        V3 light = direction;
        TangentSpace ts = tangentSpaceAt(c, m, hit);
        float dotNL = dot(ts.normal, light);
        float dotNV = dot(ts.normal, view);
        if (dotNL > 0.0f && dotNV > 0.0f) {
            V2 roughness = ggxRoughnessAt(c, m, hit.texcoords);
            V3 tsV = ts.toTangentSpace(view);
            V3 tsL = ts.toTangentSpace(light);
            V3 tsH = normalize(tsV + tsL);
            float dotVH = dot(tsV, tsH);
            if (dotVH > 0.0f) {
                p = ggxDV(tsH, tsV, dotVH, roughness) / (4.0f * dotVH);
                float Dval = ggxD(tsH, roughness);
                V4 albedo = ggxAlbedoAt(c, m, hit.texcoords);
                V4 Fval = fresnelSchlick(albedo, dotVH);
                float Gval = ggxG2(tsV, tsL, roughness);
                att = Dval * Fval * Gval / (4.0f * dotNV);
            }
        }
        return srRandom(direction, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_MODPHONG: { /* material_modphong.hpp:310-327 */
        V4 att = v4(0.0f);
        float p = 0.0f;
        V3 n = normalAt(c, m, hit);
        float cosTheta = dot(n, direction);
        if (cosTheta > 0.0f) {
            V4 kd = mpDiffuseAt(c, m, hit.texcoords);
            V4 ks = mpSpecularAt(c, m, hit.texcoords);
            float s = mpShininessAt(c, m, hit.texcoords);
            float specProb = mpSpecularProbability(kd, ks);
            att = mpAttenuation(n, -ray.direction, direction, kd, ks, s, cosTheta);
            p = mpPdfValue(n, -ray.direction, direction, s, cosTheta, specProb);
        }
        return srRandom(direction, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_RGL: { /* material_rgl.hpp:82-98 */
        V4 att = v4(0.0f);
        float p = 0.0f;
        TangentSpace ts = tangentSpaceAt(c, m, hit);
        if (dot(ts.normal, direction) > 0.0f) {
            V3 wo = ts.toTangentSpace(direction);
            V3 wi = ts.toTangentSpace(-ray.direction);
            const wpt_rgl_brdf& b = c.sc->rgl_brdfs[m.tex[0]];
            wptrgl::V3 a = wptrgl::rglEval<OracleMath>(b, c.sc->rgl_data, wptrgl::V3 { wi.x, wi.y, wi.z }, wptrgl::V3 { wo.x, wo.y, wo.z });
            V3 attenuation = V3 { a.x, a.y, a.z };
            att = V4 { attenuation.x, attenuation.y, attenuation.z, average3(attenuation) };
            p = wptrgl::rglPdf<OracleMath>(b, c.sc->rgl_data, wptrgl::V3 { wi.x, wi.y, wi.z }, wptrgl::V3 { wo.x, wo.y, wo.z });
        }
        return srRandom(direction, att, p, ray.refractiveIndex);
    }
    case WPT_MAT_TWOSIDED: /* material.hpp:298-304 */
        return hit.backside ? materialScatterToDirection(c, uint32_t(m.tex[1]), ray, toFrontSide(hit), direction)
                            : materialScatterToDirection(c, uint32_t(m.tex[0]), ray, hit, direction);
    default: /* material.hpp:174-180 */
        return ScatterRecord();
    }
}

V4 materialEmitted(const Ctx& c, uint32_t mat, const Ray& ray, const HitRecord& hit)
{
    const wpt_material& m = c.sc->materials[mat];
    switch (m.type) {
    case WPT_MAT_LIGHT_DIFFUSE: { /* light_diffuse.hpp:50-61 */
        V4 e = v4(0.0f);
        if (!hit.backside) {
            e = v4(m.v[0]);
            if (m.tex[0] >= 0) {
                V3 c3 = rgb(textureValue(c, m.tex[0], hit.texcoords));
                e = e * V4 { c3.x, c3.y, c3.z, average3(c3) };
            }
        }
        return e;
    }
    case WPT_MAT_MODPHONG: { /* material_modphong.hpp:183-190 */
        V4 e = v4(0.0f);
        if (!hit.backside)
            e = mpEmissiveAt(c, m, hit.texcoords);
        return e;
    }
    case WPT_MAT_TWOSIDED: /* material.hpp:306-312 */
        return hit.backside ? materialEmitted(c, uint32_t(m.tex[1]), ray, toFrontSide(hit))
                            : materialEmitted(c, uint32_t(m.tex[0]), ray, hit);
    default: /* material.hpp:183-186 */
        return v4(0.0f);
    }
}

/* ---- sensor_rgb.hpp:63-80 ---- */
inline void accumulateRadiance(const Ctx& c, V4 opticalPathLength, float distanceToLight, V4 radiance, float* acc)
{
    const wpt_params& p = *c.pr;
    for (int i = 0; i < 3; i++) {
        if (distanceToLight >= p.min_dist_to_light && distanceToLight <= p.max_dist_to_light
                && opticalPathLength[i] >= p.min_path_len && opticalPathLength[i] <= p.max_path_len) {
            acc[i] += radiance[i];
        }
    }
}

/* ---- wurblpt.hpp:101-106 ---- */
inline float powerHeuristicWeight(float f, float g)
{
    f *= f;
    g *= g;
    return (f + g > 0.0f ? f / (f + g) : 0.0f);
}

/* ---- wurblpt.hpp:108-275 ---- */
void tracePath(Ctx& c, float* sampleAccumulator, const Ray& startRay, size_t hotSpotsSize, float invHotSpotsSize, Prng& prng)
{
    const wpt_params& params = *c.pr;
    const wpt_scene_desc& sc = *c.sc;
    const bool haveEnv = sc.envmap.type != WPT_ENV_NONE;
    V4 attenuation = v4(1.0f);
    float pathLength = 0.0f;
    V4 opticalPathLength = v4(0.0f);
    Ray ray = startRay;

    for (unsigned int pathComponent = 0;; pathComponent++) {
        V4 radianceToAccumulate;
        HitRecord hr = bvhHit(c, ray, RayHelper(ray), params.min_hit_distance, k_maxval);
        if (!hr.haveHit) {
            if (haveEnv) {
                radianceToAccumulate = attenuation * envL(c, ray.direction);
                accumulateRadiance(c, v4(k_maxval), k_maxval, radianceToAccumulate, sampleAccumulator);
            }
            break;
        }
        pathLength += hr.a;
        opticalPathLength = opticalPathLength + hr.a * ray.refractiveIndex;
        if (!(pathComponent + 1 < params.max_path_components))
            break;

        uint32_t mat = materialOfPrim(c, hr.prim);
        c.cnt.scatters++;
        ScatterRecord sr = materialScatter(c, mat, ray, hr, prng);
        radianceToAccumulate = attenuation * materialEmitted(c, mat, ray, hr);
        accumulateRadiance(c, opticalPathLength, (pathComponent == 0 ? 0.0f : hr.a), radianceToAccumulate, sampleAccumulator);
        if (sr.type == ScatterNone)
            break;

        V4 nextAttenuation = attenuation * sr.attenuation;
        if (sr.type == ScatterRandom) {
            if (sr.pdf > 0.0f)
                nextAttenuation = nextAttenuation / sr.pdf;
            else
                nextAttenuation = v4(0.0f);
        }

        if (sr.type == ScatterRandom && hotSpotsSize > 0) {
            float hotSpotsPdf = 0.0f;
            for (size_t i = 0; i < hotSpotsSize; i++)
                hotSpotsPdf += hotSpotPdfValue(c, i, hr.position, sr.direction);
            hotSpotsPdf *= invHotSpotsSize;
            nextAttenuation = nextAttenuation * powerHeuristicWeight(sr.pdf, hotSpotsPdf);
            size_t hotSpotIndex = prng.in01() * hotSpotsSize;
            hotSpotIndex = hotSpotIndex < hotSpotsSize - 1 ? hotSpotIndex : hotSpotsSize - 1;
            V3 directDir = hotSpotDirection(c, hotSpotIndex, hr.position, prng);
            float directPdf = 0.0f;
            for (size_t i = 0; i < hotSpotsSize; i++)
                directPdf += hotSpotPdfValue(c, i, hr.position, directDir);
            directPdf *= invHotSpotsSize;
            if (directPdf > 0.0f) {
                ScatterRecord directSR = materialScatterToDirection(c, mat, ray, hr, directDir);
                if (directSR.pdf > 0.0f) {
                    Ray directRay { hr.position, directDir, ray.time, directSR.refractiveIndex };
                    HitRecord directHR = bvhHit(c, directRay, RayHelper(directRay), params.min_hit_distance, k_maxval);
                    if (directHR.haveHit && directHR.prim == hotSpotPrim(c, hotSpotIndex)) {
                        float weight = powerHeuristicWeight(directPdf, directSR.pdf);
                        uint32_t lmat = materialOfPrim(c, directHR.prim);
                        radianceToAccumulate = attenuation * directSR.attenuation / directPdf * weight
                            * materialEmitted(c, lmat, directRay, directHR);
                        V4 opticalPathLengthToAccumulate = opticalPathLength + directHR.a * directRay.refractiveIndex;
                        accumulateRadiance(c, opticalPathLengthToAccumulate, directHR.a, radianceToAccumulate, sampleAccumulator);
                    }
                }
            }
        } else if (sr.type == ScatterRandom && haveEnv && sc.envmap.N > 0) {
            float lightsP = envP(c, sr.direction);
            nextAttenuation = nextAttenuation * powerHeuristicWeight(sr.pdf, lightsP);
            V3 lightDir = envD(c, prng);
            float lightDirP = envP(c, lightDir);
            ScatterRecord lightSR = materialScatterToDirection(c, mat, ray, hr, lightDir);
            if (lightSR.pdf > 0.0f) {
                Ray lightRay { hr.position, lightDir, ray.time, lightSR.refractiveIndex };
                HitRecord lightHR = bvhHit(c, lightRay, RayHelper(lightRay), params.min_hit_distance, k_maxval);
                if (!lightHR.haveHit) {
                    float weight = powerHeuristicWeight(lightDirP, lightSR.pdf);
                    radianceToAccumulate = attenuation * lightSR.attenuation / lightDirP * weight * envL(c, lightDir);
                    accumulateRadiance(c, v4(k_maxval), k_maxval, radianceToAccumulate, sampleAccumulator);
                }
            }
        }

        attenuation = nextAttenuation;
        ray = Ray { hr.position, sr.direction, ray.time, sr.refractiveIndex };

        if (max4(attenuation) < params.rr_threshold && pathComponent >= 5) {
            float q = clamp_(1.0f - max4(attenuation), 0.0f, 0.95f);
            if (prng.in01() < q)
                break;
            float rrWeight = 1.0f / (1.0f - q);
            attenuation = attenuation * rrWeight;
        }
    }
}

/* ---- camera.hpp:123-185 (Surround_Off, no stereo, t0 == t1) ---- */
inline Ray cameraGetRay(const wpt_camera& cam, float p, float q, Prng& prng, uint32_t width = 1, uint32_t height = 1, const Ctx* c = nullptr)
{
    float stereoscopicShift = 0.0f;
    if (cam.stereoscopic_distance > 0.0f) { /* camera.hpp:128-138 */
        q *= 2.0f;
        if (q < 1.0f) {
            stereoscopicShift = -0.5f * cam.stereoscopic_distance;
        } else {
            q -= 1.0f;
            stereoscopicShift = +0.5f * cam.stereoscopic_distance;
        }
    }
    V3 O, D;
    if (cam.surround_mode == WPT_SURROUND_OFF) {
        if (cam.distortion_type != WPT_DISTORTION_NONE) /* camera.hpp:143-144 */
            wptlens::undistort(cam, p, q, width, height);
        V3 P = V3 { mix_(cam.l, cam.r, p), mix_(cam.b, cam.t, q), -1.0f };
        O = v3(0.0f);
        if (cam.lens_radius > 0.0f) { /* optics.hpp:326-334 */
            P = P * v3(cam.focus_dist);
            V2 d = cam.lens_radius * inUnitDisk(prng.in01x2());
            O = V3 { d.x, d.y, 0.0f };
        }
        D = P - O;
        O = O + V3 { stereoscopicShift, 0.0f, 0.0f };
    } else { /* camera.hpp:158-170 */
        float lon = (2.0f * p - 1.0f) * k_pi;
        if (cam.surround_mode == WPT_SURROUND_180)
            lon *= 0.5f;
        float lat = (q - 0.5f) * k_pi;
        D = V3 { m_cos(lat) * m_sin(lon), m_sin(lat), -m_cos(lat) * m_cos(lon) };
        O = V3 { -m_cos(lon), 0.0f, -m_sin(lon) } * stereoscopicShift;
    }
    /* camera.hpp:175-184: with an exposure interval the ray draws its time and the camera is taken at that time */
    float t = c ? c->pr->t0 : 0.0f;
    wptanim::Trs T;
    for (int i = 0; i < 3; i++) {
        T.t[i] = cam.translation[i];
        T.s[i] = cam.scaling[i];
    }
    for (int i = 0; i < 4; i++)
        T.q[i] = cam.rotation[i];
    if (c && c->pr->t0 != c->pr->t1) {
        t += prng.in01() * (c->pr->t1 - c->pr->t0);
        if (cam.animation >= 0)
            T = animationAt(*c, cam.animation, t);
    }
    V3 origin = v3(T.t) + quat_rotate(T.q, O * v3(T.s)); /* transformation.hpp:80-83 */
    V3 direction = quat_rotate(T.q, D);
    return Ray { origin, normalize(direction), t, v4(1.0f) };
}

} /* namespace */

extern "C" {

/* wurblpt.hpp:279-436, the pixel loop :335-381; counters are summed over threads */
int wpt_oracle_render(const wpt_scene_desc* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t block_start, uint32_t block_size,
        float* frame, wpt_counters* counters, int num_threads)
{
    if (!scene || !camera || !params || !frame || width == 0 || height == 0 || samples_sqrt == 0)
        return 1;
    if (uint64_t(block_start) + block_size > uint64_t(width) * height)
        return 1;
    if (scene->envmap.type != WPT_ENV_NONE && scene->envmap.N > 0 && (!scene->envmap.M || !scene->envmap.Ms || !scene->envmap.Mcs))
        return 2; /* importance tables requested but not attached: see wpt_oracle_envmap_tables */
    unsigned int samples = samples_sqrt * samples_sqrt;
    float invSamples = 1.0f / samples;
    float invSamplesSqrt = 1.0f / samples_sqrt;
    V2 invSize = V2 { 1.0f / width, 1.0f / height };
    size_t hotSpotsSize = scene->hotspot_count;
    float invHotSpotsSize = 1.0f / scene->hotspot_count;
    wpt_counters total;
    memset(&total, 0, sizeof(total));
    if (num_threads <= 0)
        num_threads = omp_get_max_threads();
#pragma omp parallel num_threads(num_threads)
    {
        Ctx c;
        c.sc = scene;
        c.pr = params;
        memset(&c.cnt, 0, sizeof(c.cnt));
#pragma omp for schedule(dynamic)
        for (unsigned int blockPixel = 0; blockPixel < block_size; blockPixel++) {
            unsigned int pixel = block_start + blockPixel;
            unsigned int y = pixel / width;
            unsigned int x = pixel % width;
            Prng prng(pixel);
            float sampleAccumulator[3] = { 0.0f, 0.0f, 0.0f };
            for (unsigned int sampleIndex = 0; sampleIndex < samples; sampleIndex++) {
                V2 uv = V2 { float(x), float(y) };
                if (params->randomize_ray_over_pixel) {
                    unsigned int j = sampleIndex / samples_sqrt;
                    unsigned int i = sampleIndex % samples_sqrt;
                    /* wurblpt.hpp:355: g++ evaluates `j + in01()` first */
                    float fj = j + prng.in01();
                    float fi = i + prng.in01();
                    uv = uv + V2 { fi, fj } * invSamplesSqrt;
                } else {
                    uv = uv + V2 { 0.5f, 0.5f };
                }
                uv = uv * invSize;
                Ray r = cameraGetRay(*camera, uv.x, uv.y, prng, width, height, &c);
                c.time = r.time; /* perThreadAnimationCaches[threadIndex].init(r.time), wurblpt.hpp:361 */
                c.cnt.samples++;
                tracePath(c, sampleAccumulator, r, hotSpotsSize, invHotSpotsSize, prng);
            }
            frame[size_t(pixel) * 3 + 0] = invSamples * sampleAccumulator[0];
            frame[size_t(pixel) * 3 + 1] = invSamples * sampleAccumulator[1];
            frame[size_t(pixel) * 3 + 2] = invSamples * sampleAccumulator[2];
        }
#pragma omp critical
        {
            total.samples += c.cnt.samples;
            total.rays += c.cnt.rays;
            total.node_visits += c.cnt.node_visits;
            total.leaf_tests += c.cnt.leaf_tests;
            total.pdf_tests += c.cnt.pdf_tests;
            total.scatters += c.cnt.scatters;
        }
    }
    if (counters)
        *counters = total;
    return 0;
}

/* EnvironmentMap::initializeImportanceSampling (envmap.hpp:121-158): M (normalized importance
 * per bin), Ms (bin ids sorted by descending importance), Mcs (cumulative importance) */
int wpt_oracle_envmap_tables(const wpt_scene_desc* scene, int N, float* M, int32_t* Ms, float* Mcs)
{
    if (!scene || scene->envmap.type == WPT_ENV_NONE || N <= 0)
        return 1;
    wpt_params pr;
    memset(&pr, 0, sizeof(pr));
    Ctx c;
    c.sc = scene;
    c.pr = &pr;
    float totalImportance = 0.0f;
    for (int y = 0; y < N; y++) {
        float v = (y + 0.5f) / N;
        for (int x = 0; x < N; x++) {
            float u = (x + 0.5f) / N;
            V3 d = envInvM(V2 { u, v });
            V4 L = envL(c, d);
            float importance = L.x + L.y + L.z + L.w;
            totalImportance += importance;
            M[y * N + x] = importance;
        }
    }
    for (int i = 0; i < N * N; i++)
        M[i] /= totalImportance;
    for (int i = 0; i < N * N; i++)
        Ms[i] = i;
    std::sort(Ms, Ms + N * N, [M](unsigned int i, unsigned int j) { return M[i] > M[j]; });
    float sum = 0.0f;
    for (int i = 0; i < N * N; i++) {
        sum += M[Ms[i]];
        Mcs[i] = sum;
    }
    return 0;
}

/* ---- per-function probes (used by tests/ to compare with the reference's golden vectors) ---- */

void wpt_oracle_prng(uint32_t pixel, int n, float* out)
{
    Prng prng(pixel);
    for (int i = 0; i < n; i++)
        out[i] = prng.in01();
}

void wpt_oracle_prng_x2(uint32_t pixel, int n, float* out_xy)
{
    Prng prng(pixel);
    for (int i = 0; i < n; i++) {
        V2 v = prng.in01x2();
        out_xy[2 * i] = v.x;
        out_xy[2 * i + 1] = v.y;
    }
}

/* which: 0 inUnitDisk (2 out), 1 inTriangle (3 out), 2 cosineDirection (3 out) */
void wpt_oracle_sampler(int which, int n, const float* u_xy, float* out)
{
    for (int i = 0; i < n; i++) {
        V2 u = V2 { u_xy[2 * i], u_xy[2 * i + 1] };
        if (which == 0) {
            V2 r = inUnitDisk(u);
            out[2 * i] = r.x;
            out[2 * i + 1] = r.y;
        } else {
            V3 r = which == 1 ? inTriangle(u) : which == 2 ? cosineDirection(u) : onUnitSphere(u);
            out[3 * i] = r.x;
            out[3 * i + 1] = r.y;
            out[3 * i + 2] = r.z;
        }
    }
}

/* Sampler::toSphere: in4 = direction(3) cosThetaMax, u = 2 uniforms -> 3 floats */
void wpt_oracle_to_sphere(int n, const float* in4, const float* u_xy, float* out)
{
    for (int i = 0; i < n; i++) {
        V3 r = toSphere(v3(in4 + 4 * i), in4[4 * i + 3], V2 { u_xy[2 * i], u_xy[2 * i + 1] });
        out[3 * i] = r.x;
        out[3 * i + 1] = r.y;
        out[3 * i + 2] = r.z;
    }
}

/* HitableSphere: records = centre(3) radius rotation(xyzw) scaling(3) (radius = max(scaling), as the
 * constructor computes it); rays = origin(3) direction(3) amin amax.
 * hits: haveHit a position(3) normal(3) tangent(3) texcoords(2) backside (14 floats);
 * pdf: pdfValue(origin, direction); dirs: direction(origin) with Prng(seed). */
static void sphereProbe(int n, const float* records, const wpt_keyframe* keyframes, float time, const float* rays, const int64_t* seeds,
        float* hits, float* pdf, float* dirs)
{
    for (int i = 0; i < n; i++) {
        const float* rec = records + 11 * i;
        wpt_sphere sp;
        memset(&sp, 0, sizeof(sp));
        for (int k = 0; k < 3; k++)
            sp.center[k] = rec[k];
        sp.radius = fmax_(fmax_(rec[8], rec[9]), rec[10]) == rec[3] ? rec[3] : fmax_(fmax_(rec[8], rec[9]), rec[10]);
        for (int k = 0; k < 4; k++)
            sp.rotation[k] = rec[4 + k];
        sp.animation = keyframes ? 0 : -1;
        wpt_scene_desc sc;
        memset(&sc, 0, sizeof(sc));
        sc.spheres = &sp;
        sc.sphere_count = 1;
        wpt_animation anim = { 0, 2 };
        if (keyframes) { /* two key frames per sphere */
            sc.animations = &anim;
            sc.animation_count = 1;
            sc.keyframes = keyframes + 2 * i;
            sc.keyframe_count = 2;
        }
        wpt_params pr;
        memset(&pr, 0, sizeof(pr));
        Ctx c;
        c.sc = &sc;
        c.pr = &pr;
        c.time = time;
        memset(&c.cnt, 0, sizeof(c.cnt));
        Ray r { v3(rays + 8 * i), v3(rays + 8 * i + 3), time, v4(1.0f) };
        HitRecord hr = sphereHit(c, sp, 0, r, rays[8 * i + 6], rays[8 * i + 7]);
        float* o = hits + 14 * i;
        memset(o, 0, 14 * sizeof(float));
        if (hr.haveHit) {
            o[0] = 1.0f;
            o[1] = hr.a;
            o[2] = hr.position.x; o[3] = hr.position.y; o[4] = hr.position.z;
            o[5] = hr.normal.x; o[6] = hr.normal.y; o[7] = hr.normal.z;
            o[8] = hr.tangent.x; o[9] = hr.tangent.y; o[10] = hr.tangent.z;
            o[11] = hr.texcoords.x; o[12] = hr.texcoords.y;
            o[13] = hr.backside ? 1.0f : 0.0f;
        }
        pdf[i] = spherePdfValue(c, 0, r.origin, r.direction);
        Prng prng(uint32_t(seeds[i]));
        V3 d = sphereDirection(c, 0, r.origin, prng);
        dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
    }
}

void wpt_oracle_sphere(int n, const float* records, const float* rays, const int64_t* seeds, float* hits, float* pdf, float* dirs)
{
    sphereProbe(n, records, nullptr, 0.0f, rays, seeds, hits, pdf, dirs);
}

/* the same for spheres that carry an animation of two key frames each (44 bytes per key frame), seen at `time` */
void wpt_oracle_sphere_animated(int n, const float* records, const wpt_keyframe* keyframes, float time, const float* rays,
        const int64_t* seeds, float* hits, float* pdf, float* dirs)
{
    sphereProbe(n, records, keyframes, time, rays, seeds, hits, pdf, dirs);
}

/* powitacq_rgb::BRDF: in = wi(3) wo(3) u(2); sample_out = weight(3) wo(3) pdf; eval_out = f*cos(3) pdf */
void wpt_oracle_rgl(const wpt_rgl_brdf* brdf, const float* pool, int n, const float* in, float* sample_out, float* eval_out)
{
    for (int i = 0; i < n; i++) {
        const float* q = in + 8 * i;
        wptrgl::V3 wi { q[0], q[1], q[2] }, wo { q[3], q[4], q[5] }, swo;
        float pdf;
        wptrgl::V3 w = wptrgl::rglSample<OracleMath>(*brdf, pool, wptrgl::V2 { q[6], q[7] }, wi, swo, pdf);
        float* so = sample_out + 7 * i;
        so[0] = w.x; so[1] = w.y; so[2] = w.z; so[3] = swo.x; so[4] = swo.y; so[5] = swo.z; so[6] = pdf;
        wptrgl::V3 e = wptrgl::rglEval<OracleMath>(*brdf, pool, wi, wo);
        float* eo = eval_out + 4 * i;
        eo[0] = e.x; eo[1] = e.y; eo[2] = e.z;
        eo[3] = wptrgl::rglPdf<OracleMath>(*brdf, pool, wi, wo);
    }
}

/* inverse(Transformation) (transformation.hpp:157-163) of the camera transformation */
struct InverseCamera {
    float rotation[4];
    V3 translation;
    explicit InverseCamera(const wpt_camera& cam)
    {
        rotation[0] = -cam.rotation[0];
        rotation[1] = -cam.rotation[1];
        rotation[2] = -cam.rotation[2];
        rotation[3] = cam.rotation[3];
        V3 invT = -v3(cam.translation);
        V3 invS = 1.0f / v3(cam.scaling);
        translation = quat_rotate(rotation, invT * invS);
    }
    V3 point(V3 p) const { return quat_rotate(rotation, p) + translation; }
    V3 direction(V3 d) const { return quat_rotate(rotation, d); }
};

/* Camera::cameraSpaceToImageSpace (camera.hpp:194-217) */
inline V2 cameraSpaceToImageSpace(const wpt_camera& cam, V3 p)
{
    float P00 = 2.0f / (cam.r - cam.l);
    float P11 = 2.0f / (cam.t - cam.b);
    float P03 = (cam.r + cam.l) / (cam.r - cam.l);
    float P13 = (cam.t + cam.b) / (cam.t - cam.b);
    V2 projected = V2 { P00 * p.x + P03 * p.z, P11 * p.y + P13 * p.z };
    float w = -p.z;
    V2 ndc = V2 { projected.x / w, projected.y / w };
    V2 imageCoord = 0.5f * ndc + V2 { 0.5f, 0.5f };
    wptlens::distort(cam, imageCoord.x, imageCoord.y);
    return imageCoord;
}

/* AnimationKeyframes::at(t) and what is made of it: per case 10 + 16 + 9 + 3 + 3 + 3 floats (the layout of the
 * anim_out golden vector); in: t, point (3) */
void wpt_oracle_animation(const wpt_keyframe* keyframes, uint32_t count, int n, const float* in, float* out)
{
    for (int i = 0; i < n; i++) {
        const wptanim::Trs T = wptanim::at<OracleMath>(keyframes, count, in[4 * i]);
        float* o = out + 44 * i;
        for (int k = 0; k < 3; k++) o[k] = T.t[k];
        for (int k = 0; k < 4; k++) o[3 + k] = T.q[k];
        for (int k = 0; k < 3; k++) o[7 + k] = T.s[k];
        wptanim::toMat4(T, o + 10);
        wptanim::toMat3(T.q, o + 26);
        wptanim::mulPoint(o + 10, in + 4 * i + 1, o + 35);
        wptanim::mulVec(o + 26, in + 4 * i + 1, o + 38);
        wptanim::applyTrs(T, in + 4 * i + 1, o + 41);
    }
}

/* cameraSpaceToImageSpace of n camera space points: 2 floats per point */
void wpt_oracle_camera_to_image(const wpt_camera* cam, int n, const float* points, float* out)
{
    for (int i = 0; i < n; i++) {
        V2 ic = cameraSpaceToImageSpace(*cam, v3(points + 3 * i));
        out[2 * i] = ic.x;
        out[2 * i + 1] = ic.y;
    }
}

/* world space -> camera space with inverse(camera transformation) (wurblpt.hpp:679): 3 floats per point */
void wpt_oracle_world_to_camera(const wpt_camera* cam, int n, const float* points, float* out)
{
    const InverseCamera inv(*cam);
    for (int i = 0; i < n; i++) {
        V3 p = inv.point(v3(points + 3 * i));
        out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
    }
}

/* getGroundTruth (wurblpt.hpp:626-761) for a static scene; arrays[k] (GroundTruth bit k, wurblpt_hip.h) may be NULL */
int wpt_oracle_ground_truth(const wpt_scene_desc* scene, const wpt_camera* camera, const wpt_camera* camera_prev,
        const wpt_camera* camera_next, const float* times, const wpt_params* params, uint32_t width, uint32_t height, void* const* arrays)
{
    if (!scene || !camera || !params || !arrays || width == 0 || height == 0)
        return 1;
    const float t0 = times ? times[0] : 0.0f, tPrev = times ? times[1] : 0.0f, tNext = times ? times[2] : 0.0f;
    const wpt_camera& camPrev = camera_prev ? *camera_prev : *camera;
    const wpt_camera& camNext = camera_next ? *camera_next : *camera;
    wpt_camera rayCam = *camera;
    rayCam.lens_radius = 0.0f; /* getRay(..., withRandomness = false): no depth of field offset */
    const InverseCamera inv0(*camera), invPrev(camPrev), invNext(camNext);
    const unsigned int pixels = width * height;
    float invWidth = 1.0f / width;
    float invHeight = 1.0f / height;
    auto set3 = [&](int k, unsigned int pixel, V3 v) {
        if (arrays[k]) {
            float* o = static_cast<float*>(arrays[k]) + 3 * size_t(pixel);
            o[0] = v.x; o[1] = v.y; o[2] = v.z;
        }
    };
    auto set2 = [&](int k, unsigned int pixel, V2 v) {
        if (arrays[k]) {
            float* o = static_cast<float*>(arrays[k]) + 2 * size_t(pixel);
            o[0] = v.x; o[1] = v.y;
        }
    };
#pragma omp parallel
    {
        Ctx c;
        c.sc = scene;
        c.pr = params;
        c.time = t0; /* animationCacheT0 */
        memset(&c.cnt, 0, sizeof(c.cnt));
#pragma omp for schedule(dynamic, 64)
        for (unsigned int pixel = 0; pixel < pixels; pixel++) {
            unsigned int y = pixel / width;
            unsigned int x = pixel % width;
            Prng prng { pixel };
            V2 pixelCoord = V2 { (x + 0.5f) * invWidth, (y + 0.5f) * invHeight };
            Ray ray = cameraGetRay(rayCam, pixelCoord.x, pixelCoord.y, prng, width, height);
            ray.time = t0;
            const HitRecord hr = bvhHit(c, ray, RayHelper(ray), params->min_hit_distance, k_maxval);
            V3 wsPos = v3(0.0f), wsGNrm = v3(0.0f), wsGTan = v3(0.0f), wsMNrm = v3(0.0f), wsMTan = v3(0.0f);
            V3 csPos = v3(0.0f), csGNrm = v3(0.0f), csGTan = v3(0.0f), csMNrm = v3(0.0f), csMTan = v3(0.0f);
            float csDepth = 0.0f, csDist = 0.0f;
            V2 txCor = V2 { 0.0f, 0.0f };
            V3 wsOP = v3(0.0f), wsON = v3(0.0f), csOP = v3(0.0f), csON = v3(0.0f);
            V2 psOP = V2 { 0.0f, 0.0f }, psON = V2 { 0.0f, 0.0f };
            int matInd = -1;
            if (hr.haveHit) {
                const uint32_t mat = materialOfPrim(c, hr.prim);
                wsPos = hr.position;
                wsGNrm = hr.normal;
                wsGTan = hr.tangent;
                TangentSpace ts = tangentSpaceAt(c, scene->materials[mat], hr);
                wsMNrm = ts.normal;
                wsMTan = ts.tangent;
                csPos = inv0.point(wsPos);
                csGNrm = inv0.direction(wsGNrm);
                csGTan = inv0.direction(wsGTan);
                csMNrm = inv0.direction(wsMNrm);
                csMTan = inv0.direction(wsMTan);
                csDepth = -csPos.z;
                csDist = length(csPos);
                txCor = hr.texcoords;
                V3 wsPosPrev = wsPos;
                V3 wsPosNext = wsPos;
                int ai = -1;
                if (hr.prim & PRIM_SPHERE)
                    ai = scene->spheres[hr.prim & ~PRIM_SPHERE].animation;
                else if (scene->tri_geom[hr.prim].flags & WPT_TRI_ANIMATE)
                    ai = scene->instances[scene->tri_geom[hr.prim].instance].animation;
                if (ai >= 0) { /* wurblpt.hpp:695-699 */
                    const wptanim::Trs T0 = animationAt(c, ai, t0);
                    wptanim::Trs inv; /* inverse(Transformation), transformation.hpp:157-163 */
                    inv.q[0] = -T0.q[0]; inv.q[1] = -T0.q[1]; inv.q[2] = -T0.q[2]; inv.q[3] = T0.q[3];
                    V3 invS = 1.0f / v3(T0.s);
                    V3 invT = quat_rotate(inv.q, (-v3(T0.t)) * invS);
                    inv.s[0] = invS.x; inv.s[1] = invS.y; inv.s[2] = invS.z;
                    inv.t[0] = invT.x; inv.t[1] = invT.y; inv.t[2] = invT.z;
                    const float pos[3] = { wsPos.x, wsPos.y, wsPos.z };
                    float posOrig[3], moved[3];
                    wptanim::applyTrs(inv, pos, posOrig);
                    wptanim::applyTrs(animationAt(c, ai, tPrev), posOrig, moved);
                    wsPosPrev = v3(moved);
                    wptanim::applyTrs(animationAt(c, ai, tNext), posOrig, moved);
                    wsPosNext = v3(moved);
                }
                wsOP = wsPosPrev - wsPos;
                wsON = wsPosNext - wsPos;
                V3 csPosPrev = invPrev.point(wsPosPrev);
                V3 csPosNext = invNext.point(wsPosNext);
                csOP = csPosPrev - csPos;
                csON = csPosNext - csPos;
                if (arrays[17] || arrays[18]) {
                    V2 size = V2 { float(width), float(height) };
                    V2 psPos = pixelCoord * size;
                    V2 psPosPrev = cameraSpaceToImageSpace(*camera, csPosPrev) * size;
                    V2 psPosNext = cameraSpaceToImageSpace(*camera, csPosNext) * size;
                    psOP = psPosPrev - psPos;
                    psON = psPosNext - psPos;
                }
                matInd = int(mat);
            }
            set3(0, pixel, wsPos); set3(1, pixel, wsGNrm); set3(2, pixel, wsGTan); set3(3, pixel, wsMNrm); set3(4, pixel, wsMTan);
            set3(5, pixel, csPos); set3(6, pixel, csGNrm); set3(7, pixel, csGTan); set3(8, pixel, csMNrm); set3(9, pixel, csMTan);
            if (arrays[10])
                static_cast<float*>(arrays[10])[pixel] = csDepth;
            if (arrays[11])
                static_cast<float*>(arrays[11])[pixel] = csDist;
            set2(12, pixel, txCor);
            set3(13, pixel, wsOP); set3(14, pixel, wsON); set3(15, pixel, csOP); set3(16, pixel, csON);
            set2(17, pixel, psOP); set2(18, pixel, psON);
            if (arrays[19])
                static_cast<int32_t*>(arrays[19])[pixel] = matInd;
        }
    }
    return 0;
}

/* LensDistortion::undistort then ::distort of the result (optics.hpp:214-308): pq -> 4 floats per point */
void wpt_oracle_lens(const wpt_camera* cam, uint32_t width, uint32_t height, int n, const float* pq, float* out)
{
    for (int i = 0; i < n; i++) {
        float p = pq[2 * i], q = pq[2 * i + 1];
        wptlens::undistort(*cam, p, q, width, height);
        out[4 * i] = p;
        out[4 * i + 1] = q;
        wptlens::distort(*cam, p, q);
        out[4 * i + 2] = p;
        out[4 * i + 3] = q;
    }
}

/* Camera::getRay with the frame size (needed by the iterative undistortion): origin, direction */
void wpt_oracle_camera_rays_sized(const wpt_camera* cam, uint32_t width, uint32_t height, int n, const float* pq, float* out)
{
    for (int i = 0; i < n; i++) {
        Prng prng { uint32_t(i) };
        Ray r = cameraGetRay(*cam, pq[2 * i], pq[2 * i + 1], prng, width, height);
        float* o = out + 6 * i;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
    }
}

struct OraclePow {
    static float pow(float x, float y) { return m_pow(x, y); }
};
/* colour conversions: in = rgb(3) newY; out = xyz(3) rgb(3) adjust_y(3) srgb(3); bytes = 3 per input */
void wpt_oracle_color(int n, const float* in, float* out, uint8_t* bytes)
{
    for (int i = 0; i < n; i++) {
        wptpp::V3 rgb { in[4 * i], in[4 * i + 1], in[4 * i + 2] };
        wptpp::V3 xyz = wptpp::rgbToXyz(rgb), back = wptpp::xyzToRgb(xyz), adj = wptpp::adjustY(xyz, in[4 * i + 3]);
        float* o = out + 12 * i;
        o[0] = xyz.x; o[1] = xyz.y; o[2] = xyz.z; o[3] = back.x; o[4] = back.y; o[5] = back.z; o[6] = adj.x; o[7] = adj.y; o[8] = adj.z;
        const float c[3] = { rgb.x, rgb.y, rgb.z };
        for (int k = 0; k < 3; k++) {
            const float v = c[k] < 1.0f ? c[k] : 1.0f;
            o[9 + k] = wptpp::rgbToSrgbHelper<OraclePow>(v);
            bytes[3 * i + k] = wptpp::toSrgbByte<OraclePow>(c[k]);
        }
    }
}
/* postproc.hpp restated per pixel: op 0 = uniformRationalQuantization(a = maxVal, b = brightness), 1 = scaleLuminance(a = factor, b = clamp) */
void wpt_oracle_postproc(int op, int n, const float* rgb, float a, float b, float* out)
{
    for (int i = 0; i < n; i++) {
        wptpp::V3 v { rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] };
        wptpp::V3 r = op == 0 ? wptpp::uniformRationalQuantization(v, a, b) : wptpp::scaleLuminance(v, a, b);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
float wpt_oracle_max_luminance(int n, const float* rgb)
{
    float lum = 0.0f;
    for (int i = 0; i < n; i++) {
        const float y = wptpp::luminance(wptpp::V3 { rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] });
        if (y > lum)
            lum = y;
    }
    return lum;
}

/* TangentSpace(n) (Duff) -> tangent, bitangent, then toWorldSpace(v), toTangentSpace(v): 12 floats */
void wpt_oracle_tangentspace(int n, const float* normals, const float* vecs, float* out)
{
    for (int i = 0; i < n; i++) {
        TangentSpace ts(v3(normals + 3 * i));
        V3 w = ts.toWorldSpace(v3(vecs + 3 * i));
        V3 t = ts.toTangentSpace(v3(vecs + 3 * i));
        float* o = out + 12 * i;
        o[0] = ts.tangent.x; o[1] = ts.tangent.y; o[2] = ts.tangent.z;
        o[3] = ts.bitangent.x; o[4] = ts.bitangent.y; o[5] = ts.bitangent.z;
        o[6] = w.x; o[7] = w.y; o[8] = w.z;
        o[9] = t.x; o[10] = t.y; o[11] = t.z;
    }
}

/* rays: origin(3) direction(3); out: invDir(3) k(3 as float) S(3) */
void wpt_oracle_rayhelper(int n, const float* rays, float* out)
{
    for (int i = 0; i < n; i++) {
        Ray r { v3(rays + 6 * i), v3(rays + 6 * i + 3), 0.0f, v4(1.0f) };
        RayHelper h(r);
        float* o = out + 9 * i;
        o[0] = h.invDirection.x; o[1] = h.invDirection.y; o[2] = h.invDirection.z;
        o[3] = float(h.kx); o[4] = float(h.ky); o[5] = float(h.kz);
        o[6] = h.S.x; o[7] = h.S.y; o[8] = h.S.z;
    }
}

/* boxes: lo(3) hi(3); rays: origin(3) dir(3) amin amax (8 floats) */
void wpt_oracle_aabb(int n, const float* boxes, const float* rays, int32_t* out)
{
    for (int i = 0; i < n; i++) {
        Ray r { v3(rays + 8 * i), v3(rays + 8 * i + 3), 0.0f, v4(1.0f) };
        RayHelper h(r);
        out[i] = aabbMayHit(boxes + 6 * i, boxes + 6 * i + 3, r, rays[8 * i + 6], rays[8 * i + 7], h.invDirection) ? 1 : 0;
    }
}

/* fresnelUnpolarized(cosI, cosT, n1, n2) and fresnelSchlick(vec4 r0 = a, cos) -> 5 floats */
void wpt_oracle_fresnel(int n, const float* in4, float* out)
{
    for (int i = 0; i < n; i++) {
        const float* p = in4 + 4 * i;
        out[5 * i] = fresnelUnpolarized(p[0], p[1], p[2], p[3]);
        V4 s = fresnelSchlick(V4 { p[0], p[1], p[2], p[3] }, p[1]);
        out[5 * i + 1] = s.x; out[5 * i + 2] = s.y; out[5 * i + 3] = s.z; out[5 * i + 4] = s.w;
    }
}

/* reflect(i, n), refract(i, n, eta): in 7 floats, out 6 */
void wpt_oracle_reflect_refract(int n, const float* in7, float* out)
{
    for (int i = 0; i < n; i++) {
        const float* p = in7 + 7 * i;
        V3 a = reflect(v3(p), v3(p + 3));
        V3 b = refract(v3(p), v3(p + 3), p[6]);
        float* o = out + 6 * i;
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = b.x; o[4] = b.y; o[5] = b.z;
    }
}

/* Camera::getRay for n (p, q) pairs; out origin(3) direction(3); Prng(pixel 0) for DOF */
void wpt_oracle_camera_rays(const wpt_camera* cam, int n, const float* pq, float* out)
{
    Prng prng(0);
    for (int i = 0; i < n; i++) {
        Ray r = cameraGetRay(*cam, pq[2 * i], pq[2 * i + 1], prng);
        float* o = out + 6 * i;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
    }
}

/* Material::scatter / scatterToDirection / emitted of material `mat` of the scene on caller-given rays and hit records
 * (tests/test_independent_pins.py holds them against float64 evaluations of the published models).
 * in per record, 18 floats: ray direction(3), hit normal(3), hit tangent(3), texcoords(2), backside, hit distance a,
 * generator seed (as an integer value), direction to evaluate(3), refractive index of the incoming ray.
 * out per record, 22 floats: scatter type, direction(3), attenuation(4), pdf, refractive index(4) | scatterToDirection:
 * attenuation(4), pdf | emitted(4). */
void wpt_oracle_material_probe(const wpt_scene_desc* scene, uint32_t mat, int n, const float* in, float* out)
{
    wpt_params pr;
    memset(&pr, 0, sizeof(pr));
    Ctx c;
    c.sc = scene;
    c.pr = &pr;
    memset(&c.cnt, 0, sizeof(c.cnt));
    for (int i = 0; i < n; i++) {
        const float* r = in + 18 * i;
        Ray ray { v3(0.0f), v3(r), 0.0f, v4(r[17]) };
        HitRecord hr;
        hr.haveHit = true;
        hr.normal = v3(r + 3);
        hr.tangent = v3(r + 6);
        hr.texcoords = V2 { r[9], r[10] };
        hr.backside = r[11] != 0.0f;
        hr.a = r[12];
        hr.position = ray.at(hr.a);
        hr.prim = 0;
        Prng prng((unsigned int)r[13]);
        const ScatterRecord sr = materialScatter(c, mat, ray, hr, prng);
        const ScatterRecord ev = materialScatterToDirection(c, mat, ray, hr, v3(r + 14));
        const V4 em = materialEmitted(c, mat, ray, hr);
        float* o = out + 22 * i;
        o[0] = float(sr.type);
        o[1] = sr.direction.x; o[2] = sr.direction.y; o[3] = sr.direction.z;
        o[4] = sr.attenuation.x; o[5] = sr.attenuation.y; o[6] = sr.attenuation.z; o[7] = sr.attenuation.w;
        o[8] = sr.pdf;
        o[9] = sr.refractiveIndex.x; o[10] = sr.refractiveIndex.y; o[11] = sr.refractiveIndex.z; o[12] = sr.refractiveIndex.w;
        o[13] = ev.attenuation.x; o[14] = ev.attenuation.y; o[15] = ev.attenuation.z; o[16] = ev.attenuation.w;
        o[17] = ev.pdf;
        o[18] = em.x; o[19] = em.y; o[20] = em.z; o[21] = em.w;
    }
}

/* The light sampling of tracePath (wurblpt.hpp:179-199) at caller-given points: in per record 7 floats: origin(3),
 * a direction(3), generator seed; out per record 7 floats: mean pdfValue over all hot spots of that direction
 * (wurblpt.hpp:181-185), index of the hot spot the generator picks (:187-188), the direction drawn towards it(3)
 * (Hitable::direction), the mean pdfValue of that direction (:192-195), pdfValue of the FIRST hot spot alone for the
 * given direction. */
void wpt_oracle_hotspot_probe(const wpt_scene_desc* scene, int n, const float* in, float* out)
{
    wpt_params pr;
    memset(&pr, 0, sizeof(pr));
    Ctx c;
    c.sc = scene;
    c.pr = &pr;
    memset(&c.cnt, 0, sizeof(c.cnt));
    const size_t hotSpotsSize = scene->hotspot_count;
    const float invHotSpotsSize = hotSpotsSize ? 1.0f / float(hotSpotsSize) : 0.0f;
    for (int i = 0; i < n; i++) {
        const float* r = in + 7 * i;
        float* o = out + 7 * i;
        memset(o, 0, 7 * sizeof(float));
        if (hotSpotsSize == 0)
            continue;
        const V3 origin = v3(r), direction = v3(r + 3);
        Prng prng((unsigned int)r[6]);
        float hotSpotsPdf = 0.0f;
        for (size_t k = 0; k < hotSpotsSize; k++)
            hotSpotsPdf += hotSpotPdfValue(c, k, origin, direction);
        hotSpotsPdf *= invHotSpotsSize;
        size_t hotSpotIndex = prng.in01() * hotSpotsSize;
        hotSpotIndex = hotSpotIndex < hotSpotsSize - 1 ? hotSpotIndex : hotSpotsSize - 1;
        const V3 directDir = hotSpotDirection(c, hotSpotIndex, origin, prng);
        float directPdf = 0.0f;
        for (size_t k = 0; k < hotSpotsSize; k++)
            directPdf += hotSpotPdfValue(c, k, origin, directDir);
        directPdf *= invHotSpotsSize;
        o[0] = hotSpotsPdf;
        o[1] = float(hotSpotIndex);
        o[2] = directDir.x; o[3] = directDir.y; o[4] = directDir.z;
        o[5] = directPdf;
        o[6] = hotSpotPdfValue(c, 0, origin, direction);
    }
}

/* EnvironmentMap::L / p / d of the scene's environment map (envmap.hpp:167-210): in per record 4 floats: direction(3),
 * generator seed; out per record 12 floats: L(direction)(4), p(direction), d(prng)(3), p(d), L(d).rgb sum. */
void wpt_oracle_envmap_probe(const wpt_scene_desc* scene, int n, const float* in, float* out)
{
    wpt_params pr;
    memset(&pr, 0, sizeof(pr));
    Ctx c;
    c.sc = scene;
    c.pr = &pr;
    memset(&c.cnt, 0, sizeof(c.cnt));
    for (int i = 0; i < n; i++) {
        const float* r = in + 4 * i;
        float* o = out + 10 * i;
        const V3 direction = v3(r);
        Prng prng((unsigned int)r[3]);
        const V4 L = envL(c, direction);
        o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = L.w;
        o[4] = scene->envmap.N > 0 ? envP(c, direction) : 0.0f;
        V3 d = v3(0.0f);
        float pd = 0.0f;
        if (scene->envmap.N > 0) {
            d = envD(c, prng);
            pd = envP(c, d);
        }
        o[5] = d.x; o[6] = d.y; o[7] = d.z;
        o[8] = pd;
        const V4 Ld = envL(c, d);
        o[9] = Ld.x + Ld.y + Ld.z + Ld.w;
    }
}

} /* extern "C" (the walk below has templates) */

/* ---- A walk that is NOT the reference's, held against it: DESIGN.md section 7.1 proposes a step that tests the boxes of a
 * node's four grandchildren together (the binary tree collapsed by one level) and claims every result stays the same.  This
 * is that walk on the CPU, leaf test for leaf test next to bvhTraverse: test infrastructure for a design decision, not the
 * oracle's path (wpt_oracle_render never comes here).
 *
 * A step at inner node X tests the boxes of X's grandchildren WITHOUT the bound (a child that is a leaf
 * stands for itself), in the order the reference's walk would come to them, and keeps those that pass with their entry
 * distance M = max(amin, near slabs).  The reference tests a later one under whatever bound the earlier subtrees have left;
 * since the far side of the test had already held, that test is `M <= bound`, which is what a kept child is admitted by when
 * its turn comes.  The inner nodes that disappear decide nothing of their own: a child's box lies within its parent's, the
 * slab arithmetic is monotone, so a child that passes implies the parent the reference tested before it.  Both arguments
 * need slab distances that are numbers: with a NaN among them (0 * inf: origin on a slab plane, direction parallel to it)
 * the comparison chains of gvm.hpp depend on operand order, and the step falls back to the reference's own two tests. */
/* what a kernel prototype leaves out, to see what each omission costs: 1 = no fall-back for NaN slabs, 2 = IEEE minimum / maximum
 * instead of the reference's comparison chains, 4 = the root's own box is not tested */
static int g_wideMode = 0;
struct WideWalkStats {
    uint64_t rays, binaryVisits, wideSteps, wideBoxTests, leafTests, nanFallbacks, revalidationsFailed, maxPending;
    uint64_t admissionDisagrees, parentDisagrees; /* self-checks of the two arguments, counted where they fail */
};

inline bool slabsOf(const float* lo, const float* hi, const Ray& ray, V3 invDir, float amin, float amax, float& entry, bool& pass)
{
    const V3 t0 = (v3(lo) - ray.origin) * invDir;
    const V3 t1 = (v3(hi) - ray.origin) * invDir;
    const bool number = t0.x == t0.x && t0.y == t0.y && t0.z == t0.z && t1.x == t1.x && t1.y == t1.y && t1.z == t1.z;
    if (g_wideMode & 2) {
        entry = std::fmax(std::fmax(amin, std::fmin(t0.x, t1.x)), std::fmax(std::fmin(t0.y, t1.y), std::fmin(t0.z, t1.z)));
        const float far = std::fmin(std::fmin(amax, std::fmax(t0.x, t1.x)), std::fmin(std::fmax(t0.y, t1.y), std::fmax(t0.z, t1.z)));
        pass = entry <= far;
        return number || (g_wideMode & 1);
    }
    const V4 tmin = V4 { amin, fmin_(t0.x, t1.x), fmin_(t0.y, t1.y), fmin_(t0.z, t1.z) };
    const V4 tmax = V4 { amax, fmax_(t0.x, t1.x), fmax_(t0.y, t1.y), fmax_(t0.z, t1.z) };
    entry = max4(tmin);
    pass = entry <= min4(tmax);
    return number || (g_wideMode & 1);
}

template<typename LeafHit>
inline HitRecord bvhTraverseWide(const wpt_bvh_node* nodes, const Ray& ray, const RayHelper& rh, float amin, float amax,
        LeafHit&& leafHit, std::vector<uint32_t>& leafOrder, WideWalkStats& st)
{
    struct Pending {
        uint32_t node;
        bool tested; /* kept by a wide step with entry distance `entry`; false: the reference's own test is still to be made */
        float entry;
    };
    HitRecord hr;
    std::vector<Pending> pending;
    pending.push_back(Pending { 0u, (g_wideMode & 4) != 0, 0.0f });
    while (!pending.empty()) {
        if (pending.size() > st.maxPending)
            st.maxPending = pending.size();
        const Pending e = pending.back();
        pending.pop_back();
        const wpt_bvh_node& node = nodes[e.node];
        if (e.tested) {
            /* self-check: the admission by entry distance is the reference's test under the bound of this moment */
            if ((e.entry <= amax) != aabbMayHit(node.lo, node.hi, ray, amin, amax, rh.invDirection))
                st.admissionDisagrees++;
            if (!(e.entry <= amax)) {
                st.revalidationsFailed++;
                continue;
            }
        } else {
            st.wideBoxTests++;
            if (!aabbMayHit(node.lo, node.hi, ray, amin, amax, rh.invDirection))
                continue;
        }
        if (node.kind != WPT_NODE_INNER) {
            if (node.kind != WPT_NODE_EMPTY) {
                st.leafTests++;
                leafOrder.push_back(node.kind == WPT_NODE_SPHERE ? (0x80000000u | node.link) : node.link);
                HitRecord cur = leafHit(node.kind, node.link, amin, amax);
                if (cur.haveHit) {
                    hr = cur;
                    amax = hr.a;
                }
            }
            continue;
        }
        /* one step: the grandchildren in the reference's order (first child = index + 1, second = link) */
        st.wideSteps++;
        const uint32_t child[2] = { e.node + 1u, node.link };
        uint32_t grand[4];
        int count = 0;
        for (int k = 0; k < 2; k++) {
            const wpt_bvh_node& ch = nodes[child[k]];
            if (ch.kind == WPT_NODE_INNER) {
                grand[count++] = child[k] + 1u;
                grand[count++] = ch.link;
            } else {
                grand[count++] = child[k];
            }
        }
        float entry[4];
        bool pass[4];
        bool numbers = true;
        for (int k = 0; k < count; k++) {
            st.wideBoxTests++;
            /* Without the bound: a hit is accepted by one comparison and its distance stored by another (hitable_triangle.hpp:
             * 283-296), so the bound can GROW by an ulp at a hit, and a box that is beyond it now may be within it when its
             * turn comes (found by this check: one ray in 200 000 on the Sponza-class scene).  The bound is applied at the
             * child's turn only, where it is the reference's. */
            numbers = slabsOf(nodes[grand[k]].lo, nodes[grand[k]].hi, ray, rh.invDirection, amin, k_maxval, entry[k], pass[k]) && numbers;
        }
        /* the children that disappear must be numbers too for the implication child => parent to hold */
        for (int k = 0; k < 2 && numbers; k++) {
            float en;
            bool pa;
            numbers = slabsOf(nodes[child[k]].lo, nodes[child[k]].hi, ray, rh.invDirection, amin, amax, en, pa);
        }
        if (!numbers) {
            st.nanFallbacks++;
            pending.push_back(Pending { child[1], false, 0.0f });
            pending.push_back(Pending { child[0], false, 0.0f });
            continue;
        }
        /* self-check: a grandchild that passes implies its parent (the child the reference tests first) */
        {
            int k = 0;
            for (int c2 = 0; c2 < 2; c2++) {
                const wpt_bvh_node& ch = nodes[child[c2]];
                const int members = ch.kind == WPT_NODE_INNER ? 2 : 1;
                const bool parentPasses = aabbMayHit(ch.lo, ch.hi, ray, amin, k_maxval, rh.invDirection);
                for (int m = 0; m < members; m++, k++)
                    if (pass[k] && !parentPasses)
                        st.parentDisagrees++;
            }
        }
        for (int k = count - 1; k >= 0; k--)
            if (pass[k])
                pending.push_back(Pending { grand[k], true, entry[k] });
    }
    return hr;
}

extern "C" {

/* n rays (origin, dir, amin, amax = 8 floats) through both walks: returns the number of rays whose sequence of leaf tests or
 * whose result (hit, primitive, distance bits) differs; stats: 8 x uint64 (WideWalkStats). */
void wpt_oracle_wide_walk_mode(int mode) { g_wideMode = mode; }

int wpt_oracle_wide_walk_check(const wpt_scene_desc* scene, int n, const float* rays, uint64_t* stats)
{
    wpt_params pr;
    memset(&pr, 0, sizeof(pr));
    WideWalkStats st;
    memset(&st, 0, sizeof(st));
    int differ = 0;
    Ctx c;
    c.sc = scene;
    c.pr = &pr;
    memset(&c.cnt, 0, sizeof(c.cnt));
    std::vector<uint32_t> orderBinary, orderWide;
    for (int i = 0; i < n; i++) {
        Ray r { v3(rays + 8 * i), v3(rays + 8 * i + 3), 0.0f, v4(1.0f) };
        const RayHelper rh(r);
        auto leaf = [&](std::vector<uint32_t>* order) {
            return [&c, &r, &rh, order](uint32_t kind, uint32_t index, float lo, float hi) {
                if (order)
                    order->push_back(kind == WPT_NODE_SPHERE ? (0x80000000u | index) : index);
                if (kind == WPT_NODE_SPHERE)
                    return sphereHit(c, c.sc->spheres[index], index, r, lo, hi);
                return triangleHit(c, index, r, rh, lo, hi, true, nullptr, nullptr);
            };
        };
        orderBinary.clear();
        orderWide.clear();
        const uint64_t visitsBefore = c.cnt.node_visits;
        const HitRecord a = bvhTraverse(c.sc->nodes, c.cnt, r, rh, rays[8 * i + 6], rays[8 * i + 7], leaf(&orderBinary));
        st.binaryVisits += c.cnt.node_visits - visitsBefore;
        const HitRecord b = bvhTraverseWide(c.sc->nodes, r, rh, rays[8 * i + 6], rays[8 * i + 7], leaf(nullptr), orderWide, st);
        st.rays++;
        uint32_t abits, bbits;
        memcpy(&abits, &a.a, 4);
        memcpy(&bbits, &b.a, 4);
        if (orderBinary != orderWide || a.haveHit != b.haveHit || (a.haveHit && (a.prim != b.prim || abits != bbits))) {
            differ++;
            if (getenv("WPT_ORACLE_WIDE_DEBUG")) {
                fprintf(stderr, "ray %d: binary hit %d prim %u a %.9g | wide hit %d prim %u a %.9g\n binary leaves:", i, a.haveHit, a.prim, a.a, b.haveHit, b.prim, b.a);
                for (uint32_t x : orderBinary) fprintf(stderr, " %u", x);
                fprintf(stderr, "\n wide leaves:  ");
                for (uint32_t x : orderWide) fprintf(stderr, " %u", x);
                fprintf(stderr, "\n");
            }
        }
    }
    memcpy(stats, &st, sizeof(st));
    return differ;
}

/* BVH::hit against the scene for n rays (origin, dir, amin, amax = 8 floats).
 * out per ray: haveHit, prim, a, position(3), normal(3), tangent(3), texcoords(2), backside = 15 floats */
void wpt_oracle_bvh_hits(const wpt_scene_desc* scene, int n, const float* rays, float* out, wpt_counters* counters)
{
    wpt_params pr;
    memset(&pr, 0, sizeof(pr));
    Ctx c;
    c.sc = scene;
    c.pr = &pr;
    memset(&c.cnt, 0, sizeof(c.cnt));
    for (int i = 0; i < n; i++) {
        Ray r { v3(rays + 8 * i), v3(rays + 8 * i + 3), 0.0f, v4(1.0f) };
        HitRecord hr = bvhHit(c, r, RayHelper(r), rays[8 * i + 6], rays[8 * i + 7]);
        float* o = out + 15 * i;
        memset(o, 0, 15 * sizeof(float));
        o[0] = hr.haveHit ? 1.0f : 0.0f;
        if (hr.haveHit) {
            o[1] = float(hr.prim);
            o[2] = hr.a;
            o[3] = hr.position.x; o[4] = hr.position.y; o[5] = hr.position.z;
            o[6] = hr.normal.x; o[7] = hr.normal.y; o[8] = hr.normal.z;
            o[9] = hr.tangent.x; o[10] = hr.tangent.y; o[11] = hr.tangent.z;
            o[12] = hr.texcoords.x; o[13] = hr.texcoords.y;
            o[14] = hr.backside ? 1.0f : 0.0f;
        }
    }
    if (counters)
        *counters = c.cnt;
}

/* BVH::hit over caller-given nodes whose leaves are "probe hitables": leaf `prim` reports a
 * hit at distance leaf_a[prim] iff that lies in [amin, amax] (oracle/ref_probe.cpp ProbeHitable).
 * Writes the leaf visiting order (up to max_log entries) and the final hit. */
void wpt_oracle_bvh_walk(const wpt_bvh_node* nodes, const float* ray8, const float* leaf_a,
        int64_t* log, int64_t max_log, int64_t* log_len, int64_t* final_prim, float* final_a)
{
    wpt_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    Ray r { v3(ray8), v3(ray8 + 3), 0.0f, v4(1.0f) };
    int64_t n = 0;
    HitRecord hr = bvhTraverse(nodes, cnt, r, RayHelper(r), ray8[6], ray8[7], [&](uint32_t, uint32_t prim, float lo, float hi) {
        if (n < max_log)
            log[n] = prim;
        n++;
        HitRecord h;
        float a = leaf_a[prim];
        if (a >= lo && a <= hi) {
            h.haveHit = true;
            h.a = a;
            h.prim = prim;
        }
        return h;
    });
    *log_len = n;
    *final_prim = hr.haveHit ? int64_t(hr.prim) : -1;
    *final_a = hr.haveHit ? hr.a : 0.0f;
}

/* math back end probe: op 0 sin, 1 cos, 2 exp, 3 pow(a,b), 4 asin, 5 atan2(a,b), 10 acos, 11 atan2(a, 1), 12 float(2 * asin(double)) */
void wpt_oracle_math(int op, int n, const float* a, const float* b, float* out)
{
    for (int i = 0; i < n; i++) {
        switch (op) {
        case 0: out[i] = m_sin(a[i]); break;
        case 1: out[i] = m_cos(a[i]); break;
        case 2: out[i] = m_exp(a[i]); break;
        case 3: out[i] = m_pow(a[i], b[i]); break;
        case 4: out[i] = m_asin(a[i]); break;
        case 10: out[i] = m_acos(a[i]); break;
        case 11: out[i] = m_atan2(a[i], 1.0f); break;
        case 12: out[i] = OracleMath::twiceAsin(a[i]); break;
        default: out[i] = m_atan2(a[i], b[i]); break;
        }
    }
}

const char* wpt_oracle_backend(void)
{
#ifdef WPT_ORACLE_LIBM
    return "libm";
#else
    return "portable";
#endif
}

} /* extern "C" */
