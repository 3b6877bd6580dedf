/*
 * optics.hpp -- projection and thin-lens parameters (reference optics.hpp:37-110,311-352).
 * Lens distortion models (optics.hpp:112-309) are outside the device path ("next" row).
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
};

class LensDistortion
{
public:
    bool active;
    LensDistortion() : active(false) {}
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
