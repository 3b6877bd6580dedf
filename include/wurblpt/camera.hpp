/*
 * camera.hpp -- camera description (reference camera.hpp:42-120).  Ray generation itself
 * (camera.hpp:123-185) runs in the HIP kernel from the wpt_camera record made here.
 * Surround / stereoscopic modes and camera animation are outside the device path.
 */
#pragma once

#include "../wurblpt_hip.h"
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

    Camera(SurroundMode surroundMode, float stereoscopicDistance, const Optics& optics,
            const Transformation& transformation = Transformation()) :
        surroundMode(surroundMode), stereoscopicDistance(stereoscopicDistance), optics(optics), transformation(transformation)
    {
    }
    Camera(const Optics& optics, const Transformation& transformation = Transformation()) :
        surroundMode(Surround_Off), stereoscopicDistance(0.0f), optics(optics), transformation(transformation)
    {
    }

    Transformation at(float /* t */ = 0.0f) const { return transformation; }

    /* false if this camera needs a feature the kernel does not have */
    bool describe(wpt_camera& out) const
    {
        if (surroundMode != Surround_Off || stereoscopicDistance > 0.0f || optics.distortion.active)
            return false;
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
        return true;
    }
};

}
