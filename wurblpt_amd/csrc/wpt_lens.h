/*
 * wpt_lens.h -- LensDistortion::distort / undistort (optics.hpp:214-309) on the wpt_camera record:
 * RadialAndPlanar (closed form, Lambers et al.), RadialOnly (exact inverse series, Drap and Lefevre),
 * OpenCV (fixed-point iteration until the reprojection error is below 0.001 pixel, at most 256 steps).
 * Written once; compiled for the device (blockNew) and by the test oracle, which is pinned to the
 * reference's own optics.hpp (oracle/ref_probe.cpp).
 */
#ifndef WPT_LENS_H
#define WPT_LENS_H

#include "../../include/wurblpt_hip.h"

#if defined(__HIPCC__)
#define WPT_LENS_HD __host__ __device__ __forceinline__
#define WPT_LENS_ENTRY static inline __host__ __device__ __attribute__((noinline))
#else
#define WPT_LENS_HD inline
#define WPT_LENS_ENTRY inline
#endif

namespace wptlens {

/* the distortion part of a wpt_camera record: small enough to travel by value into the out-of-line device entry, so
 * that no caller has to keep its camera record in memory for it */
struct Coeffs {
    uint32_t type;
    float cx, cy;     /* dist_center */
    float ifx, ify;   /* dist_inverse_focal_length */
    float fx, fy;     /* dist_focal_length */
    float k1, k2, k3, p1, p2;
    float b1, b2, b3, b4;
};
struct Point {
    float p, q;
};
WPT_LENS_HD Coeffs coeffs(const wpt_camera& c)
{
    Coeffs k;
    k.type = c.distortion_type;
    k.cx = c.dist_center[0];
    k.cy = c.dist_center[1];
    k.ifx = c.dist_inverse_focal_length[0];
    k.ify = c.dist_inverse_focal_length[1];
    k.fx = c.dist_focal_length[0];
    k.fy = c.dist_focal_length[1];
    k.k1 = c.k1; k.k2 = c.k2; k.k3 = c.k3; k.p1 = c.p1; k.p2 = c.p2;
    k.b1 = c.b1; k.b2 = c.b2; k.b3 = c.b3; k.b4 = c.b4;
    return k;
}

/* optics.hpp:214-238 */
WPT_LENS_HD void distort(const Coeffs& c, float& p, float& q)
{
    if (c.type == WPT_DISTORTION_NONE)
        return;
    const float s = (p - c.cx) * c.ifx;
    const float t = (q - c.cy) * c.ify;
    const float r2 = s * s + t * t;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float rd = 1.0f + c.k1 * r2 + c.k2 * r4 + c.k3 * r6;
    const float term1 = 2.0f * s * t;
    const float term2 = r2 + 2.0f * s * s;
    const float term3 = r2 + 2.0f * t * t;
    const float newS = s * rd + c.p1 * term1 + c.p2 * term2;
    const float newT = t * rd + c.p1 * term3 + c.p2 * term1;
    p = newS * c.fx + c.cx;
    q = newT * c.fy + c.cy;
}
WPT_LENS_HD void distort(const wpt_camera& cam, float& p, float& q)
{
    distort(coeffs(cam), p, q);
}

/* optics.hpp:241-308; width and height are the frame's (the iteration's error is measured in pixels) */
WPT_LENS_HD Point undistortPoint(const Coeffs& c, float p, float q, uint32_t width, uint32_t height)
{
    if (c.type == WPT_DISTORTION_RADIAL_AND_PLANAR) {
        const float s = (p - c.cx) * c.ifx;
        const float t = (q - c.cy) * c.ify;
        const float r2 = s * s + t * t;
        const float r4 = r2 * r2;
        const float d1 = c.k1 * r2 + c.k2 * r4;
        const float d2 = 1.0f / (4.0f * c.k1 * r2 + 6.0f * c.k2 * r4 + 8.0f * c.p1 * t + 8.0f * c.p2 * s + 1.0f);
        p = (s - d2 * (d1 * s + 2.0f * c.p1 * s * t + c.p2 * (r2 + 2.0f * s * s))) * c.fx + c.cx;
        q = (t - d2 * (d1 * t + c.p1 * (r2 + 2.0f * t * t) + 2.0f * c.p2 * s * t)) * c.fy + c.cy;
    } else if (c.type == WPT_DISTORTION_RADIAL_ONLY) {
        const float s = (p - c.cx) * c.ifx;
        const float t = (q - c.cy) * c.ify;
        const float r2 = s * s + t * t;
        const float r4 = r2 * r2;
        const float r6 = r4 * r2;
        const float r8 = r4 * r4;
        const float d = 1.0f + c.b1 * r2 + c.b2 * r4 + c.b3 * r6 + c.b4 * r8;
        p = s * d * c.fx + c.cx;
        q = t * d * c.fy + c.cy;
    } else if (c.type == WPT_DISTORTION_OPENCV) {
        float s = (p - c.cx) * c.ifx;
        float t = (q - c.cy) * c.ify;
        const float s0 = s, t0 = t;
        const int maxIterations = 256;
        const float epsilon = 0.001f;
        float squaredError = 3.402823466e+38f;
        for (int i = 0; i < maxIterations && squaredError >= epsilon * epsilon; i++) {
            const float r2 = s * s + t * t;
            const float r4 = r2 * r2;
            const float r6 = r4 * r2;
            const float rd = 1.0f + c.k1 * r2 + c.k2 * r4 + c.k3 * r6;
            const float invrd = 1.0f / rd;
            const float ds = 2.0f * c.p1 * s * t + c.p2 * (r2 + 2.0f * s * s);
            const float dt = 2.0f * c.p2 * s * t + c.p1 * (r2 + 2.0f * t * t);
            s = (s0 - ds) * invrd;
            t = (t0 - dt) * invrd;
            float ep = s * c.fx + c.cx;
            float eq = t * c.fy + c.cy;
            distort(c, ep, eq);
            const float ox = (ep - p) * (float)width, oy = (eq - q) * (float)height;
            float e = 0.0f; /* dot(): accumulates from zero */
            e += ox * ox;
            e += oy * oy;
            squaredError = e;
        }
        p = s * c.fx + c.cx;
        q = t * c.fy + c.cy;
    }
    Point r;
    r.p = p;
    r.q = q;
    return r;
}
/* out of line on the device (few scenes run it, and inlined it takes registers from every camera ray); everything
 * goes in and out by value */
WPT_LENS_ENTRY Point undistortCall(Coeffs c, float p, float q, uint32_t width, uint32_t height)
{
    return undistortPoint(c, p, q, width, height);
}
WPT_LENS_HD void undistort(const wpt_camera& cam, float& p, float& q, uint32_t width, uint32_t height)
{
    const Point r = undistortCall(coeffs(cam), p, q, width, height);
    p = r.p;
    q = r.q;
}

} /* namespace wptlens */

#endif
