/*
 * camera.hpp -- camera description (reference camera.hpp:42-120).  Ray generation itself
 * (camera.hpp:123-185) runs in the HIP kernel from the wpt_camera record made here.
 * Surround and stereoscopic modes, lens distortion and depth of field are part of the record; an
 * animated camera is described at t0 and its key frames go to the kernel with the scene's (mcpt()).
 */
#pragma once

#include "../wurblpt_hip.h"
#include <memory>

#include "animation.hpp"
#include "optics.hpp"
#include "transformation.hpp"

namespace WurblPT {

class Camera
{
public:
    enum SurroundMode { Surround_Off, Surround_180, Surround_360 };

    SurroundMode surroundMode;
    float stereoscopicDistance;
    Optics optics;
    Transformation transformation;
    std::shared_ptr<const Animation> animation; /* takes the place of `transformation` when set; owned by the camera */

    Camera(SurroundMode surroundMode, float stereoscopicDistance, const Optics& optics,
            const Transformation& transformation = Transformation()) :
        surroundMode(surroundMode), stereoscopicDistance(stereoscopicDistance), optics(optics), transformation(transformation)
    {
    }
    Camera(const Optics& optics, const Transformation& transformation = Transformation()) :
        surroundMode(Surround_Off), stereoscopicDistance(0.0f), optics(optics), transformation(transformation)
    {
    }

    Camera(SurroundMode surroundMode, float stereoscopicDistance, const Optics& optics, const Animation* animation) :
        surroundMode(surroundMode), stereoscopicDistance(stereoscopicDistance), optics(optics), transformation(), animation(animation)
    {
    }
    Camera(const Optics& optics, const Animation* animation) :
        surroundMode(Surround_Off), stereoscopicDistance(0.0f), optics(optics), transformation(), animation(animation)
    {
    }

    Transformation at(float t = 0.0f) const { return animation ? animation->at(t) : transformation; }

    /* The camera at time t0 (Camera::getRayHelper, camera.hpp:114-120); false if it needs a feature the kernel
     * does not have.  out.animation is -1: the caller enters the index of the key frames in the scene's pool. */
    bool describe(wpt_camera& out, float t0 = 0.0f) const
    {
        const Transformation transformation = at(t0);
        out.animation = -1;
        out.surround_mode = surroundMode == Surround_180 ? WPT_SURROUND_180 : surroundMode == Surround_360 ? WPT_SURROUND_360 : WPT_SURROUND_OFF;
        out.stereoscopic_distance = stereoscopicDistance;
        out.l = optics.projection.l;
        out.r = optics.projection.r;
        out.b = optics.projection.b;
        out.t = optics.projection.t;
        for (int k = 0; k < 3; k++) {
            out.translation[k] = transformation.translation[k];
            out.scaling[k] = transformation.scaling[k];
        }
        out.rotation[0] = transformation.rotation.x;
        out.rotation[1] = transformation.rotation.y;
        out.rotation[2] = transformation.rotation.z;
        out.rotation[3] = transformation.rotation.w;
        out.lens_radius = optics.depthOfField.lensRadius;
        out.focus_dist = optics.depthOfField.focusDist;
        /* LensDistortion and its helper (optics.hpp:203-212) */
        const LensDistortion& ld = optics.distortion;
        out.distortion_type = ld.type == LensDistortion::RadialAndPlanar ? WPT_DISTORTION_RADIAL_AND_PLANAR
            : ld.type == LensDistortion::RadialOnly ? WPT_DISTORTION_RADIAL_ONLY
            : ld.type == LensDistortion::OpenCV ? WPT_DISTORTION_OPENCV : WPT_DISTORTION_NONE;
        out.k1 = ld.k1; out.k2 = ld.k2; out.k3 = ld.k3; out.p1 = ld.p1; out.p2 = ld.p2;
        out.b1 = ld.b1; out.b2 = ld.b2; out.b3 = ld.b3; out.b4 = ld.b4;
        const vec2 c = optics.projection.center(), f = optics.projection.focalLength(), fi = optics.projection.inverseFocalLength();
        out.dist_center[0] = c.x(); out.dist_center[1] = c.y();
        out.dist_focal_length[0] = f.x(); out.dist_focal_length[1] = f.y();
        out.dist_inverse_focal_length[0] = fi.x(); out.dist_inverse_focal_length[1] = fi.y();
        return true;
    }
};

}
