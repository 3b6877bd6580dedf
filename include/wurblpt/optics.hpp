/*
 * optics.hpp -- projection and thin-lens parameters (reference optics.hpp:37-110,311-352).
 * The lens distortion models (optics.hpp:112-309) are described here and evaluated by the kernel
 * (wurblpt_amd/csrc/wpt_lens.h).
 */
#pragma once

#include "gvm.hpp"

namespace WurblPT {

class Projection
{
public:
    float t, b, r, l; /* frustum at near = 1 */

    Projection(float l_, float r_, float b_, float t_) : t(t_), b(b_), r(r_), l(l_) {}
    /* from the vertical opening angle and the aspect ratio */
    Projection(float vfov, float aspectRatio) : t(tan(vfov * 0.5f)), b(-t), r(t * aspectRatio), l(-r) {}
    /* from OpenCV-style intrinsics */
    Projection(unsigned int width, unsigned int height, const vec2& centerPixel, const vec2& focalLength) :
        t((height - centerPixel.y()) / focalLength.y()), b((0.0f - centerPixel.y()) / focalLength.y()),
        r((width - centerPixel.x()) / focalLength.x()), l((0.0f - centerPixel.x()) / focalLength.x())
    {
    }
    Projection() : Projection(radians(60.0f), 4.0f / 3.0f) {}

    float vFov() const { return atan(t) - atan(b); }
    float hFov() const { return atan(r) - atan(l); }
    float aspectRatio() const { return (r - l) / (t - b); }
    vec2 center() const { return vec2(l / (l - r), b / (b - t)); }
    vec2 focalLength() const { return vec2(1.0f / (r - l), 1.0f / (t - b)); }
    vec2 inverseFocalLength() const { return vec2(r - l, t - b); }
};

/* optics.hpp:112-212: three models, chosen by the constructor */
class LensDistortion
{
public:
    typedef enum { None, RadialAndPlanar, RadialOnly, OpenCV } Type;
    Type type;
    float k1, k2, k3, p1, p2;
    float b1, b2, b3, b4;

    LensDistortion() : type(None), k1(0.0f), k2(0.0f), k3(0.0f), p1(0.0f), p2(0.0f), b1(0.0f), b2(0.0f), b3(0.0f), b4(0.0f) {}
    LensDistortion(float k1_, float k2_, float p1_, float p2_) :
        type((k1_ == 0.0f && k2_ == 0.0f && p1_ == 0.0f && p2_ == 0.0f) ? None : RadialAndPlanar),
        k1(k1_), k2(k2_), k3(0.0f), p1(p1_), p2(p2_), b1(0.0f), b2(0.0f), b3(0.0f), b4(0.0f)
    {
    }
    LensDistortion(float k1_, float k2_, float k3_) :
        type((k1_ == 0.0f && k2_ == 0.0f && k3_ == 0.0f) ? None : RadialOnly),
        k1(k1_), k2(k2_), k3(k3_), p1(0.0f), p2(0.0f),
        b1(-k1), b2(3.0f * k1 * k1 - k2), b3(-12.0f * k1 * k1 * k1 + 8.0f * k1 * k2 - k3),
        b4(55.0f * k1 * k1 * k1 * k1 - 55.0f * k1 * k1 * k2 + 5.0f * k2 * k2 + 10.0f * k1 * k3)
    {
    }
    LensDistortion(float k1_, float k2_, float k3_, float p1_, float p2_) :
        type((k1_ == 0.0f && k2_ == 0.0f && k3_ == 0.0f && p1_ == 0.0f && p2_ == 0.0f) ? None : OpenCV),
        k1(k1_), k2(k2_), k3(k3_), p1(p1_), p2(p2_), b1(0.0f), b2(0.0f), b3(0.0f), b4(0.0f)
    {
    }
};

class LensDepthOfField
{
public:
    float lensRadius;
    float focusDist;
    LensDepthOfField(float aperture = 0.0f, float focusDist_ = 1.0f) : lensRadius(aperture * 0.5f), focusDist(focusDist_) {}
};

class Optics
{
public:
    Projection projection;
    LensDistortion distortion;
    LensDepthOfField depthOfField;
    Optics(const Projection& P = Projection(), const LensDistortion& LD = LensDistortion(),
            const LensDepthOfField& LDOF = LensDepthOfField()) :
        projection(P), distortion(LD), depthOfField(LDOF)
    {
    }
};

}
