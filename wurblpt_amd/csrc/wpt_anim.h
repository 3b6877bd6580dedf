/*
 * wpt_anim.h -- animated transformations at a ray's time: AnimationKeyframes::at
 * (animation_keyframes.hpp:71-96,186-214), mix() of two transformations (transformation.hpp:199-205),
 * quaternion slerp (gvm.hpp:1765-1797), Transformation::toMat4 / toNormalMatrix
 * (transformation.hpp:105-122, gvm.hpp:1979-2075) and the matrix products they are applied with.
 * The operation order is the reference's, down to the sums of the 4x4 product that builds toMat4().
 * Written once; compiled for the device (kernels with FEAT_ANIM) and by the test oracle, which is
 * pinned to the reference's own headers (oracle/ref_probe.cpp, `anim_*` golden vectors).
 * M supplies acos and sin: the device's double-evaluated ones, or libm in the oracle's libm build.
 */
#ifndef WPT_ANIM_H
#define WPT_ANIM_H

#include "../../include/wurblpt_hip.h"

#if defined(__HIPCC__)
#define WPT_ANIM_HD __host__ __device__ __forceinline__
#else
#define WPT_ANIM_HD inline
#endif

namespace wptanim {

struct Trs {
    float t[3];
    float q[4]; /* x, y, z, w */
    float s[3];
};

WPT_ANIM_HD float mixf(float x, float y, float a) { return x + a * (y - x); } /* gvm.hpp:166 */

template<class M> WPT_ANIM_HD void slerp(const float* q, const float* r, float alpha, float* out)
{
    float w[4] = { r[0], r[1], r[2], r[3] };
    float cosHalfAngle = q[0] * r[0] + q[1] * r[1] + q[2] * r[2] + q[3] * r[3];
    if (cosHalfAngle < 0.0f) {
        w[0] = -w[0];
        w[1] = -w[1];
        w[2] = -w[2];
        w[3] = -w[3];
        cosHalfAngle = -cosHalfAngle;
    }
    float tmpQ, tmpW;
    if (cosHalfAngle >= 1.0f) {
        tmpQ = 1.0f;
        tmpW = 0.0f;
    } else {
        const float halfAngle = M::acos(cosHalfAngle);
        const float sinHalfAngle = M::sqrt(1.0f - cosHalfAngle * cosHalfAngle);
        if ((sinHalfAngle < 0.0f ? -sinHalfAngle : sinHalfAngle) < 1.1920928955078125e-7f) {
            tmpQ = 0.5f;
            tmpW = 0.5f;
        } else {
            tmpQ = M::sin((1.0f - alpha) * halfAngle) / sinHalfAngle;
            tmpW = M::sin(alpha * halfAngle) / sinHalfAngle;
        }
    }
    for (int i = 0; i < 4; i++)
        out[i] = q[i] * tmpQ + w[i] * tmpW;
}

WPT_ANIM_HD Trs keyframeTrs(const wpt_keyframe& k)
{
    Trs r;
    for (int i = 0; i < 3; i++) {
        r.t[i] = k.translation[i];
        r.s[i] = k.scaling[i];
    }
    for (int i = 0; i < 4; i++)
        r.q[i] = k.rotation[i];
    return r;
}

/* AnimationKeyframes::at(t) for `count` key frames sorted by time */
template<class M> WPT_ANIM_HD Trs at(const wpt_keyframe* kf, uint32_t count, float t)
{
    Trs r;
    if (count == 0) {
        r.t[0] = r.t[1] = r.t[2] = 0.0f;
        r.q[0] = r.q[1] = r.q[2] = 0.0f;
        r.q[3] = 1.0f;
        r.s[0] = r.s[1] = r.s[2] = 1.0f;
        return r;
    }
    if (t <= kf[0].t)
        return keyframeTrs(kf[0]);
    if (t >= kf[count - 1].t)
        return keyframeTrs(kf[count - 1]);
    /* binary search for the neighbours (animation_keyframes.hpp:71-96) */
    int a = 0, b = int(count) - 1;
    int lower = -1, higher = -1;
    while (b >= a) {
        const int c = (a + b) / 2;
        if (kf[c].t < t) {
            a = c + 1;
        } else if (kf[c].t > t) {
            b = c - 1;
        } else {
            lower = higher = c;
            break;
        }
    }
    if (lower < 0) {
        lower = b;
        higher = a;
    }
    if (lower == higher)
        return keyframeTrs(kf[lower]);
    const float alpha = 1.0f - (kf[higher].t - t) / (kf[higher].t - kf[lower].t);
    for (int i = 0; i < 3; i++) {
        r.t[i] = mixf(kf[lower].translation[i], kf[higher].translation[i], alpha);
        r.s[i] = mixf(kf[lower].scaling[i], kf[higher].scaling[i], alpha);
    }
    slerp<M>(kf[lower].rotation, kf[higher].rotation, alpha, r.q);
    return r;
}

/* toMat3(quaternion), column major (gvm.hpp:1979-2001) */
WPT_ANIM_HD void toMat3(const float* q, float* m)
{
    const float xx = q[0] * q[0], xy = q[0] * q[1], xz = q[0] * q[2], xw = q[0] * q[3];
    const float yy = q[1] * q[1], yz = q[1] * q[2], yw = q[1] * q[3];
    const float zz = q[2] * q[2], zw = q[2] * q[3];
    m[0] = 1.0f - 2.0f * (yy + zz);
    m[1] = 2.0f * (xy + zw);
    m[2] = 2.0f * (xz - yw);
    m[3] = 2.0f * (xy - zw);
    m[4] = 1.0f - 2.0f * (xx + zz);
    m[5] = 2.0f * (yz + xw);
    m[6] = 2.0f * (xz + yw);
    m[7] = 2.0f * (yz - xw);
    m[8] = 1.0f - 2.0f * (xx + yy);
}

/* Transformation::toMat4(): translation matrix times rotation matrix (the full 4x4 product with its
 * sums, so that signed zeros come out as they do there), columns scaled; column major */
WPT_ANIM_HD void toMat4(const Trs& T, float* out)
{
    float A[16] = { 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, T.t[0], T.t[1], T.t[2], 1.0f };
    float r3[9];
    toMat3(T.q, r3);
    float B[16] = { r3[0], r3[1], r3[2], 0.0f, r3[3], r3[4], r3[5], 0.0f, r3[6], r3[7], r3[8], 0.0f, 0.0f, 0.0f, 0.0f, 1.0f };
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float o = 0.0f;
            for (int k = 0; k < 4; k++)
                o += A[k * 4 + r] * B[c * 4 + k];
            out[c * 4 + r] = o;
        }
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 4; r++)
            out[c * 4 + r] *= T.s[c];
}

/* (M * vec4(p, 1)).xyz() */
WPT_ANIM_HD void mulPoint(const float* m, const float* p, float* out)
{
    for (int r = 0; r < 3; r++) {
        float o = 0.0f;
        o += m[r] * p[0];
        o += m[4 + r] * p[1];
        o += m[8 + r] * p[2];
        o += m[12 + r] * 1.0f;
        out[r] = o;
    }
}

/* mat3 * vec3 */
WPT_ANIM_HD void mulVec(const float* m, const float* v, float* out)
{
    for (int r = 0; r < 3; r++) {
        float o = 0.0f;
        o += m[r] * v[0];
        o += m[3 + r] * v[1];
        o += m[6 + r] * v[2];
        out[r] = o;
    }
}

/* Transformation * vec3: translation + rotation * (v * scaling) (transformation.hpp:80-83) */
WPT_ANIM_HD void applyTrs(const Trs& T, const float* v, float* out)
{
    const float sv[3] = { v[0] * T.s[0], v[1] * T.s[1], v[2] * T.s[2] };
    const float s[3] = { T.q[0], T.q[1], T.q[2] };
    const float c1[3] = { s[1] * sv[2] - s[2] * sv[1], s[2] * sv[0] - s[0] * sv[2], s[0] * sv[1] - s[1] * sv[0] };
    const float t[3] = { 2.0f * c1[0], 2.0f * c1[1], 2.0f * c1[2] };
    const float c2[3] = { s[1] * t[2] - s[2] * t[1], s[2] * t[0] - s[0] * t[2], s[0] * t[1] - s[1] * t[0] };
    for (int i = 0; i < 3; i++)
        out[i] = T.t[i] + ((sv[i] + T.q[3] * t[i]) + c2[i]);
}

}

#endif
