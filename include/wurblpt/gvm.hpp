/*
 * gvm.hpp -- small GLSL-flavoured vector/matrix/quaternion types for the host side.
 *
 * Mirrors the subset of the reference's gvm.hpp API that scene-building code uses
 * (vec2/3/4, ivec3/uvec3, mat3/mat4, quat, radians/normalize/cross/dot/mix/...).
 * Host-side numbers that reach the integrator (vertex data, camera pose, BVH boxes)
 * must be the reference's numbers, so the arithmetic ORDER of these operations follows
 * the reference: dot() accumulates from 0 (reference gvm.hpp:1183-1189), normalize is a
 * true division by the length (:1201-1204), matrix*vector accumulates per column
 * (:1481-1491), quaternion rotation uses the 2*cross form (:1713-1720).
 */
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <type_traits>

namespace WurblPT {

template<typename T> constexpr T pi_v = T(3.1415926535897932384626433832795029L);
inline constexpr float pi = pi_v<float>;
inline constexpr float pi_2 = float(1.5707963267948966192313216916397514L);
inline constexpr float pi_4 = float(0.7853981633974483096156608458198757L);
inline constexpr float inv_pi = float(0.3183098861837906715377675267450287L);
inline constexpr float sqrt2 = float(1.4142135623730950488016887242096981L);
inline constexpr float inv_sqrt2 = float(0.7071067811865475244008443621048490L);
inline constexpr float maxval = std::numeric_limits<float>::max();
inline constexpr float minval = std::numeric_limits<float>::lowest();
inline constexpr float epsilon = std::numeric_limits<float>::epsilon();
template<typename T> constexpr T epsilon_v = std::numeric_limits<T>::epsilon();

/* scalar helpers with the reference's comparison-based semantics (gvm.hpp:88-98) */
template<typename T> constexpr T min(T x, T y) requires(std::is_arithmetic_v<T>) { return x < y ? x : y; }
template<typename T> constexpr T max(T x, T y) requires(std::is_arithmetic_v<T>) { return x > y ? x : y; }
template<typename T> constexpr T clamp(T x, T lo, T hi) requires(std::is_arithmetic_v<T>) { return min(hi, max(lo, x)); }
template<typename T> constexpr T mix(T x, T y, T a) requires(std::is_floating_point_v<T>) { return x + a * (y - x); }
template<typename T> constexpr T sqr(T x) requires(std::is_arithmetic_v<T>) { return x * x; }
template<typename T> constexpr T radians(T x) requires(std::is_floating_point_v<T>) { return x * (pi_v<T> / T(180.0L)); }
template<typename T> constexpr T degrees(T x) requires(std::is_floating_point_v<T>) { return x * (T(180.0L) / pi_v<T>); }
template<typename T> constexpr T fract(T x) requires(std::is_floating_point_v<T>) { return x - std::floor(x); }
template<typename T> constexpr T inversesqrt(T x) requires(std::is_floating_point_v<T>) { return T(1) / std::sqrt(x); }
template<typename T> constexpr T sign(T x) requires(std::is_arithmetic_v<T>) { return x < T(0) ? T(-1) : x > T(0) ? T(1) : T(0); }
using std::abs;
using std::acos;
using std::asin;
using std::atan;
using std::cos;
using std::exp;
using std::floor;
using std::isfinite;
using std::log;
using std::pow;
using std::sin;
using std::sqrt;
using std::tan;
template<typename T> T atan(T y, T x) requires(std::is_floating_point_v<T>) { return std::atan2(y, x); }

template<typename T, int N> struct vector {
    T values[N];

    vector() {}
    vector(T s)
    {
        for (int i = 0; i < N; i++)
            values[i] = s;
    }
    explicit vector(const T* p)
    {
        for (int i = 0; i < N; i++)
            values[i] = p[i];
    }
    explicit vector(T a, T b) requires(N == 2) : values { a, b } {}
    explicit vector(T a, T b, T c) requires(N == 3) : values { a, b, c } {}
    explicit vector(const vector<T, 2>& ab, T c) requires(N == 3) : values { ab[0], ab[1], c } {}
    explicit vector(T a, T b, T c, T d) requires(N == 4) : values { a, b, c, d } {}
    explicit vector(const vector<T, 3>& abc, T d) requires(N == 4) : values { abc[0], abc[1], abc[2], d } {}
    explicit vector(const vector<T, 2>& ab, T c, T d) requires(N == 4) : values { ab[0], ab[1], c, d } {}

    T& operator[](std::ptrdiff_t i) { return values[i]; }
    T operator[](std::ptrdiff_t i) const { return values[i]; }
    const T* data() const { return values; }

    T& x() { return values[0]; }
    T x() const { return values[0]; }
    T& y() requires(N >= 2) { return values[1]; }
    T y() const requires(N >= 2) { return values[1]; }
    T& z() requires(N >= 3) { return values[2]; }
    T z() const requires(N >= 3) { return values[2]; }
    T& w() requires(N >= 4) { return values[3]; }
    T w() const requires(N >= 4) { return values[3]; }
    T& r() { return values[0]; }
    T r() const { return values[0]; }
    T& g() requires(N >= 2) { return values[1]; }
    T g() const requires(N >= 2) { return values[1]; }
    T& b() requires(N >= 3) { return values[2]; }
    T b() const requires(N >= 3) { return values[2]; }
    T& a() requires(N >= 4) { return values[3]; }
    T a() const requires(N >= 4) { return values[3]; }
    T s() const { return values[0]; }
    T t() const requires(N >= 2) { return values[1]; }
    vector<T, 2> xy() const requires(N >= 2) { return vector<T, 2>(values[0], values[1]); }
    vector<T, 2> rg() const requires(N >= 2) { return vector<T, 2>(values[0], values[1]); }
    vector<T, 3> xyz() const requires(N >= 3) { return vector<T, 3>(values[0], values[1], values[2]); }
    vector<T, 3> rgb() const requires(N >= 3) { return vector<T, 3>(values[0], values[1], values[2]); }

    friend vector radians(const vector& p) { vector o; for (int i = 0; i < N; i++) o[i] = radians(p[i]); return o; }
    friend vector degrees(const vector& p) { vector o; for (int i = 0; i < N; i++) o[i] = degrees(p[i]); return o; }
    friend vector operator+(const vector& p, const vector& q) { vector o; for (int i = 0; i < N; i++) o[i] = p[i] + q[i]; return o; }
    friend vector operator-(const vector& p, const vector& q) { vector o; for (int i = 0; i < N; i++) o[i] = p[i] - q[i]; return o; }
    friend vector operator*(const vector& p, const vector& q) { vector o; for (int i = 0; i < N; i++) o[i] = p[i] * q[i]; return o; }
    friend vector operator/(const vector& p, const vector& q) { vector o; for (int i = 0; i < N; i++) o[i] = p[i] / q[i]; return o; }
    friend vector operator-(const vector& p) { vector o; for (int i = 0; i < N; i++) o[i] = -p[i]; return o; }
    friend vector operator*(T k, const vector& p) { vector o; for (int i = 0; i < N; i++) o[i] = k * p[i]; return o; }
    friend vector operator*(const vector& p, T k) { vector o; for (int i = 0; i < N; i++) o[i] = p[i] * k; return o; }
    friend vector operator/(T k, const vector& p) { vector o; for (int i = 0; i < N; i++) o[i] = k / p[i]; return o; }
    friend vector operator/(const vector& p, T k) { vector o; for (int i = 0; i < N; i++) o[i] = p[i] / k; return o; }
    vector& operator+=(const vector& q) { for (int i = 0; i < N; i++) values[i] += q[i]; return *this; }
    vector& operator-=(const vector& q) { for (int i = 0; i < N; i++) values[i] -= q[i]; return *this; }
    vector& operator*=(const vector& q) { for (int i = 0; i < N; i++) values[i] *= q[i]; return *this; }
    vector& operator/=(const vector& q) { for (int i = 0; i < N; i++) values[i] /= q[i]; return *this; }
    friend bool operator==(const vector& p, const vector& q)
    {
        for (int i = 0; i < N; i++)
            if (!(p[i] == q[i]))
                return false;
        return true;
    }
    friend bool operator!=(const vector& p, const vector& q) { return !(p == q); }

    friend T dot(const vector& p, const vector& q)
    {
        T d = T(0);
        for (int i = 0; i < N; i++)
            d += p[i] * q[i];
        return d;
    }
    friend T length(const vector& p) { return sqrt(dot(p, p)); }
    friend T distance(const vector& p, const vector& q) { return length(p - q); }
    friend vector normalize(const vector& p) { return p / length(p); }
    friend vector cross(const vector& v, const vector& u) requires(N == 3)
    {
        return vector(v[1] * u[2] - v[2] * u[1], v[2] * u[0] - v[0] * u[2], v[0] * u[1] - v[1] * u[0]);
    }
    friend vector reflect(const vector& i, const vector& n) requires(N == 3) { return i - T(2) * dot(n, i) * n; }
    friend vector min(const vector& p, const vector& q) { vector o; for (int i = 0; i < N; i++) o[i] = min(p[i], q[i]); return o; }
    friend vector max(const vector& p, const vector& q) { vector o; for (int i = 0; i < N; i++) o[i] = max(p[i], q[i]); return o; }
    friend vector min(const vector& p, const vector& q, const vector& s) { return min(min(p, q), s); }
    friend vector max(const vector& p, const vector& q, const vector& s) { return max(max(p, q), s); }
    friend vector abs(const vector& p) { vector o; for (int i = 0; i < N; i++) o[i] = abs(p[i]); return o; }
    friend vector mix(const vector& p, const vector& q, T k) { vector o; for (int i = 0; i < N; i++) o[i] = mix(p[i], q[i], k); return o; }
    friend vector clamp(const vector& p, T lo, T hi) { vector o; for (int i = 0; i < N; i++) o[i] = clamp(p[i], lo, hi); return o; }
    friend T min(const vector& p) { T o = p[0]; for (int i = 1; i < N; i++) if (p[i] < o) o = p[i]; return o; }
    friend T max(const vector& p) { T o = p[0]; for (int i = 1; i < N; i++) if (p[i] > o) o = p[i]; return o; }
    friend T average(const vector& p)
    {
        constexpr T inv_N = T(1) / T(N);
        T sum = 0;
        for (int i = 0; i < N; i++)
            sum += p[i];
        return inv_N * sum;
    }
    friend vector<bool, N> equal(const vector& p, const vector& q) { vector<bool, N> o; for (int i = 0; i < N; i++) o[i] = p[i] == q[i]; return o; }
    friend vector<bool, N> notEqual(const vector& p, const vector& q) { vector<bool, N> o; for (int i = 0; i < N; i++) o[i] = p[i] != q[i]; return o; }
    friend vector<bool, N> isfinite(const vector& p) { vector<bool, N> o; for (int i = 0; i < N; i++) o[i] = std::isfinite(p[i]); return o; }
};

template<int N> bool all(const vector<bool, N>& p) { for (int i = 0; i < N; i++) if (!p[i]) return false; return true; }
template<int N> bool any(const vector<bool, N>& p) { for (int i = 0; i < N; i++) if (p[i]) return true; return false; }

typedef vector<float, 2> vec2;
typedef vector<float, 3> vec3;
typedef vector<float, 4> vec4;
typedef vector<int, 3> ivec3;
typedef vector<unsigned int, 3> uvec3;

/* column-major matrix, C columns of R rows */
template<typename T, int C, int R> struct matrix {
    T values[C * R];
    matrix() {}
    matrix(T d)
    {
        for (int c = 0; c < C; c++)
            for (int r = 0; r < R; r++)
                values[c * R + r] = (c == r ? d : T(0));
    }
    explicit matrix(const vector<T, R>& c0, const vector<T, R>& c1, const vector<T, R>& c2) requires(C == 3)
    {
        for (int r = 0; r < R; r++) { values[r] = c0[r]; values[R + r] = c1[r]; values[2 * R + r] = c2[r]; }
    }
    explicit matrix(const vector<T, R>& c0, const vector<T, R>& c1, const vector<T, R>& c2, const vector<T, R>& c3) requires(C == 4)
    {
        for (int r = 0; r < R; r++) { values[r] = c0[r]; values[R + r] = c1[r]; values[2 * R + r] = c2[r]; values[3 * R + r] = c3[r]; }
    }
    T* operator[](std::ptrdiff_t c) { return values + c * R; }
    const T* operator[](std::ptrdiff_t c) const { return values + c * R; }
    const T* data() const { return values; }

    friend vector<T, R> operator*(const matrix& m, const vector<T, C>& v)
    {
        vector<T, R> o;
        for (int r = 0; r < R; r++) {
            o[r] = T(0);
            for (int c = 0; c < C; c++)
                o[r] += m[c][r] * v[c];
        }
        return o;
    }
    friend matrix<T, R, R> operator*(const matrix<T, C, R>& m, const matrix<T, R, C>& n)
    {
        matrix<T, R, R> o;
        for (int c = 0; c < R; c++)
            for (int r = 0; r < R; r++) {
                o[c][r] = T(0);
                for (int k = 0; k < C; k++)
                    o[c][r] += m[k][r] * n[c][k];
            }
        return o;
    }
    matrix& operator*=(const matrix& n)
    {
        matrix t = *this * n;
        *this = t;
        return *this;
    }
    void scale(const vector<T, 3>& s) requires(C == 4 && R == 4)
    {
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 4; r++)
                values[c * 4 + r] *= s[c];
    }
    /* post-multiplies a translation: the last column moves along the first three (reference gvm.hpp:1571-1577) */
    void translate(const vector<T, 3>& t) requires(C == 4 && R == 4)
    {
        for (int r = 0; r < 4; r++)
            values[12 + r] = dot(vector<T, 3>(values[r], values[4 + r], values[8 + r]), t) + values[12 + r];
    }
    friend matrix translate(const matrix& m, const vector<T, 3>& v) requires(C == 4 && R == 4) { matrix M = m; M.translate(v); return M; }
    friend matrix scale(const matrix& m, const vector<T, 3>& v) requires(C == 4 && R == 4) { matrix M = m; M.scale(v); return M; }
};
typedef matrix<float, 3, 3> mat3;
typedef matrix<float, 4, 4> mat4;

template<typename T> struct quaternion {
    T x, y, z, w;
    quaternion() {}
    quaternion(T x_, T y_, T z_, T w_) : x(x_), y(y_), z(z_), w(w_) {}
    constexpr static quaternion null() { return quaternion(T(0), T(0), T(0), T(1)); }
    quaternion operator-() const { return quaternion(-x, -y, -z, w); } /* conjugate, as in the reference */
    quaternion operator*(const quaternion& q) const
    {
        quaternion p;
        p.x = w * q.x + x * q.w + y * q.z - z * q.y;
        p.y = w * q.y + y * q.w + z * q.x - x * q.z;
        p.z = w * q.z + z * q.w + x * q.y - y * q.x;
        p.w = w * q.w - x * q.x - y * q.y - z * q.z;
        return p;
    }
    const quaternion& operator*=(const quaternion& q)
    {
        *this = *this * q;
        return *this;
    }
    friend bool operator==(const quaternion& p, const quaternion& q) { return p.x == q.x && p.y == q.y && p.z == q.z && p.w == q.w; }
    friend bool operator!=(const quaternion& p, const quaternion& q) { return !(p == q); }
    friend vector<T, 3> operator*(const quaternion& q, const vector<T, 3>& v)
    {
        vector<T, 3> s(q.x, q.y, q.z);
        vector<T, 3> t = T(2) * cross(s, v);
        return v + q.w * t + cross(s, t);
    }
};
typedef quaternion<float> quat;

/* spherical linear interpolation (reference gvm.hpp:1765-1797) */
template<typename T> quaternion<T> slerp(const quaternion<T>& q, const quaternion<T>& r, T alpha)
{
    quaternion<T> w = r;
    T cosHalfAngle = q.x * r.x + q.y * r.y + q.z * r.z + q.w * r.w;
    if (cosHalfAngle < T(0)) { /* q and -q are the same rotation */
        w = quaternion<T>(-w.x, -w.y, -w.z, -w.w);
        cosHalfAngle = -cosHalfAngle;
    }
    T tmpQ, tmpW;
    if (cosHalfAngle >= T(1)) {
        tmpQ = T(1);
        tmpW = T(0);
    } else {
        T halfAngle = std::acos(cosHalfAngle);
        T sinHalfAngle = std::sqrt(T(1) - cosHalfAngle * cosHalfAngle);
        if (std::abs(sinHalfAngle) < epsilon_v<T>) {
            tmpQ = T(0.5);
            tmpW = T(0.5);
        } else {
            tmpQ = std::sin((T(1) - alpha) * halfAngle) / sinHalfAngle;
            tmpW = std::sin(alpha * halfAngle) / sinHalfAngle;
        }
    }
    return quaternion<T>(q.x * tmpQ + w.x * tmpW, q.y * tmpQ + w.y * tmpW, q.z * tmpQ + w.z * tmpW, q.w * tmpQ + w.w * tmpW);
}

/* Duff et al. orthonormal basis helper (reference gvm.hpp:1828-1835) */
template<typename T> vector<T, 3> someTangentTo(const vector<T, 3>& v)
{
    T sg = std::copysign(T(1), v.z());
    T a = T(-1) / (sg + v.z());
    T b = v.x() * v.y() * a;
    return vector<T, 3>(T(1) + sg * v.x() * v.x() * a, sg * b, -sg * v.x());
}

template<typename T> quaternion<T> toQuat(T angle, const vector<T, 3>& axis)
{
    vector<T, 3> n = normalize(axis);
    T sin_a = sin(T(0.5l) * angle);
    T cos_a = cos(T(0.5l) * angle);
    return quaternion<T>(n.x() * sin_a, n.y() * sin_a, n.z() * sin_a, cos_a);
}

template<typename T> quaternion<T> toQuat(const vector<T, 3>& dir1, const vector<T, 3>& dir2)
{
    T cosAngle = dot(dir1, dir2);
    if (cosAngle >= T(1) - epsilon_v<T>)
        return quaternion<T>::null();
    if (cosAngle <= T(-1) + epsilon_v<T>) {
        vector<T, 3> ax = someTangentTo(dir1);
        return quaternion<T>(ax.x(), ax.y(), ax.z(), T(0));
    }
    return toQuat(acos(cosAngle), cross(dir1, dir2));
}

/* rotation about x, then y, then z by the three angles of `euler` (reference gvm.hpp:1873-1890): the product
 * qz * qy * qx of the three axis rotations written out, half angles, terms in the reference's order */
template<typename T> quaternion<T> toQuat(const vector<T, 3>& euler)
{
    const T half = T(0.5l);
    const T cx = cos(half * euler.x()), sx = sin(half * euler.x());
    const T cy = cos(half * euler.y()), sy = sin(half * euler.y());
    const T cz = cos(half * euler.z()), sz = sin(half * euler.z());
    return quaternion<T>(sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz, cx * cy * cz + sx * sy * sz);
}

template<typename T> matrix<T, 3, 3> toMat3(const quaternion<T>& q)
{
    matrix<T, 3, 3> m;
    T xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
    T yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w;
    T zz = q.z * q.z, zw = q.z * q.w;
    m[0][0] = T(1) - T(2) * (yy + zz);
    m[0][1] = T(2) * (xy + zw);
    m[0][2] = T(2) * (xz - yw);
    m[1][0] = T(2) * (xy - zw);
    m[1][1] = T(1) - T(2) * (xx + zz);
    m[1][2] = T(2) * (yz + xw);
    m[2][0] = T(2) * (xz + yw);
    m[2][1] = T(2) * (yz - xw);
    m[2][2] = T(1) - T(2) * (xx + yy);
    return m;
}

template<typename T> matrix<T, 4, 4> toMat4(const quaternion<T>& q)
{
    matrix<T, 3, 3> r3 = toMat3(q);
    matrix<T, 4, 4> m(T(1));
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            m[c][r] = r3[c][r];
    return m;
}

template<typename T> matrix<T, 4, 4> rotate(const matrix<T, 4, 4>& m, const quaternion<T>& q) { return m * toMat4(q); }

}
