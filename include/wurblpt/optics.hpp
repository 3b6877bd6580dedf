/*
 * optics.hpp -- projection and thin-lens parameters (reference optics.hpp:37-110,311-352).
 * The lens distortion models (optics.hpp:112-309) are described here and evaluated by the kernel
 * (wurblpt_amd/csrc/wpt_lens.h).
 *
 * Interface (class, member and function names, argument order) and the arithmetic that the bit-parity contract fixes
 * follow marlam/wurblpt, which is distributed under the MIT licence: Copyright (c) 2023 Martin Lambers
 * <marlam@marlam.de>; the permission notice is reproduced in the LICENSE file of this repository.  The implementation
 * below is this repository's own.
 */
#pragma once

#include "gvm.hpp"

namespace WurblPT {

/* The view frustum as its four side planes at distance 1 (optics.hpp:37-110). */
class Projection
{
private:
    struct Extent { float low, high; };
    /* symmetric about the axis: half the opening, to both sides */
    static Extent symmetric(float halfOpening) { return Extent { -halfOpening, halfOpening }; }
    /* a sensor axis of `pixels` pixels whose principal point and focal length are given in pixels */
    static Extent fromIntrinsics(unsigned int pixels, float center, float focal) { return Extent { (0.0f - center) / focal, (pixels - center) / focal }; }
    Projection(const Extent& horizontal, const Extent& vertical) : t(vertical.high), b(vertical.low), r(horizontal.high), l(horizontal.low) {}

public:
    float t, b, r, l;

    Projection(float l_, float r_, float b_, float t_) : t(t_), b(b_), r(r_), l(l_) {}
    /* vertical opening angle and aspect ratio; the horizontal extent is the vertical one times the ratio */
    Projection(float vfov, float aspectRatio) : Projection(symmetric(tan(vfov * 0.5f) * aspectRatio), symmetric(tan(vfov * 0.5f))) {}
    /* OpenCV-style intrinsics of a width x height image */
    Projection(unsigned int width, unsigned int height, const vec2& centerPixel, const vec2& focalLength) :
        Projection(fromIntrinsics(width, centerPixel.x(), focalLength.x()), fromIntrinsics(height, centerPixel.y(), focalLength.y()))
    {
    }
    Projection() : Projection(radians(60.0f), 4.0f / 3.0f) {}

    float vFov() const { return atan(t) - atan(b); }
    float hFov() const { return atan(r) - atan(l); }
    float aspectRatio() const { return (r - l) / (t - b); }
    vec2 center() const { return vec2(l / (l - r), b / (b - t)); }
    vec2 inverseFocalLength() const { return vec2(r - l, t - b); }
    vec2 focalLength() const { return vec2(1.0f / (r - l), 1.0f / (t - b)); }
};

/* Lens distortion: which of the three models (optics.hpp:112-212) applies follows from the coefficients given;
 * all-zero coefficients mean no distortion.  The models are evaluated by the kernels (wurblpt_amd/csrc/wpt_lens.h). */
class LensDistortion
{
public:
    typedef enum { None, RadialAndPlanar, RadialOnly, OpenCV } Type;
    Type type;
    float k1, k2, k3, p1, p2;
    float b1, b2, b3, b4; /* RadialOnly: coefficients of the exact inverse series (Drap and Lefevre) */

private:
    void set(Type model, float k1_, float k2_, float k3_, float p1_, float p2_)
    {
        k1 = k1_; k2 = k2_; k3 = k3_; p1 = p1_; p2 = p2_;
        b1 = b2 = b3 = b4 = 0.0f;
        const bool nothing = k1 == 0.0f && k2 == 0.0f && k3 == 0.0f && p1 == 0.0f && p2 == 0.0f;
        type = nothing ? None : model;
    }

public:
    LensDistortion() { set(None, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f); }
    LensDistortion(float k1_, float k2_, float p1_, float p2_) { set(RadialAndPlanar, k1_, k2_, 0.0f, p1_, p2_); }
    LensDistortion(float k1_, float k2_, float k3_)
    {
        set(RadialOnly, k1_, k2_, k3_, 0.0f, 0.0f);
        /* products from left to right, as the reference writes them (optics.hpp:170-173): the bits depend on it */
        b1 = -k1;
        b2 = 3.0f * k1 * k1 - k2;
        b3 = -12.0f * k1 * k1 * k1 + 8.0f * k1 * k2 - k3;
        b4 = 55.0f * k1 * k1 * k1 * k1 - 55.0f * k1 * k1 * k2 + 5.0f * k2 * k2 + 10.0f * k1 * k3;
    }
    LensDistortion(float k1_, float k2_, float k3_, float p1_, float p2_) { set(OpenCV, k1_, k2_, k3_, p1_, p2_); }
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
