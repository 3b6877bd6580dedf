/*
 * transformation.hpp -- translation / rotation / scaling pose, the host-side mirror of the
 * reference's Transformation (transformation.hpp:45-207).  toMat4()/toNormalMatrix() feed
 * vertex baking (mesh.hpp) and fromLookAt() feeds the camera, so their arithmetic follows
 * the reference (:105-137).
 *
 * Interface (class, member and function names, argument order) and the arithmetic that the bit-parity contract fixes
 * follow marlam/wurblpt, which is distributed under the MIT licence: Copyright (c) 2023 Martin Lambers
 * <marlam@marlam.de>; the permission notice is reproduced in the LICENSE file of this repository.  The implementation
 * below is this repository's own.
 */
#pragma once

#include "gvm.hpp"

namespace WurblPT {

/* A pose: scale, then rotate, then translate.  Composition appends on the right, so that T.translate(v) moves
 * along T's own (rotated and scaled) axes. */
class Transformation
{
public:
    vec3 translation;
    quat rotation;
    vec3 scaling;

    Transformation(const vec3& t = vec3(0.0f), const quat& r = quat::null(), const vec3& s = vec3(1.0f)) :
        translation(t), rotation(r), scaling(s)
    {
    }

    /* applied to a point (40 % dearer than a 4x4 product on the reference author's machine, transformation.hpp:40-42;
     * meshes therefore bake toMat4() instead) */
    vec3 operator*(const vec3& point) const { return translation + (rotation * (point * scaling)); }

    void translate(const vec3& v) { translation += rotation * (v * scaling); }
    void rotate(const quat& q) { rotation *= q; }
    void scale(const vec3& s) { scaling *= s; }

    bool isIdentity() const
    {
        const bool unmoved = all(equal(translation, vec3(0.0f)));
        const bool unturned = rotation.w >= 1.0f; /* a unit quaternion with w = 1 has no vector part */
        const bool unscaled = all(equal(scaling, vec3(1.0f)));
        return unmoved && unturned && unscaled;
    }
    bool operator==(const Transformation& other) const
    {
        return all(equal(translation, other.translation)) && rotation == other.rotation && all(equal(scaling, other.scaling));
    }

    /* translation matrix times rotation matrix, columns then scaled: the order of the products is part of the bits */
    mat4 toMat4() const
    {
        mat4 M(vec4(1.0f, 0.0f, 0.0f, 0.0f), vec4(0.0f, 1.0f, 0.0f, 0.0f), vec4(0.0f, 0.0f, 1.0f, 0.0f), vec4(translation, 1.0f));
        M *= WurblPT::toMat4(rotation);
        M.scale(scaling);
        return M;
    }
    /* for directions that must stay perpendicular to surfaces: the rotation alone (normals are renormalised later) */
    mat3 toNormalMatrix() const { return toMat3(rotation); }

    /* the pose of a viewer at `eye` looking at `center`: first the rotation that takes -z to the view direction, then
     * the one about that direction that takes the turned y axis to the viewer's up vector */
    static Transformation fromLookAt(const vec3& eye, const vec3& center, const vec3& up = vec3(0.0f, 1.0f, 0.0f))
    {
        const vec3 forward = normalize(center - eye);
        const vec3 side = normalize(cross(forward, up));
        const vec3 viewerUp = cross(side, forward);
        const quat aim = toQuat(vec3(0.0f, 0.0f, -1.0f), forward);
        const quat roll = toQuat(aim * vec3(0.0f, 1.0f, 0.0f), viewerUp);
        return Transformation(eye, roll * aim);
    }
    vec3 lookFrom() const { return translation; }
    vec3 lookAt() const { return translation + rotation * vec3(0.0f, 0.0f, -1.0f); }
    vec3 up() const { return rotation * vec3(0.0f, 1.0f, 0.0f); }
};

inline Transformation translate(const Transformation& T, const vec3& v) { Transformation R = T; R.translate(v); return R; }
inline Transformation rotate(const Transformation& T, const quat& q) { Transformation R = T; R.rotate(q); return R; }
inline Transformation scale(const Transformation& T, const vec3& s) { Transformation R = T; R.scale(s); return R; }
/* positions and scalings linearly, rotations by slerp (transformation.hpp:199-205) */
inline Transformation mix(const Transformation& T0, const Transformation& T1, float alpha)
{
    return Transformation(mix(T0.translation, T1.translation, alpha), slerp(T0.rotation, T1.rotation, alpha), mix(T0.scaling, T1.scaling, alpha));
}
inline Transformation operator*(const Transformation& S, const Transformation& T)
{
    return scale(rotate(translate(S, T.translation), T.rotation), T.scaling);
}

}
