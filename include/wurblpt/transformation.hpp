/*
 * transformation.hpp -- translation / rotation / scaling pose, the host-side mirror of the
 * reference's Transformation (transformation.hpp:45-207).  toMat4()/toNormalMatrix() feed
 * vertex baking (mesh.hpp) and fromLookAt() feeds the camera, so their arithmetic follows
 * the reference (:105-137).
 */
#pragma once

#include "gvm.hpp"

namespace WurblPT {

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

    bool isIdentity() const
    {
        return all(equal(translation, vec3(0.0f))) && rotation.w >= 1.0f && all(equal(scaling, vec3(1.0f)));
    }

    bool operator==(const Transformation& o) const
    {
        return all(equal(translation, o.translation)) && rotation == o.rotation && all(equal(scaling, o.scaling));
    }

    vec3 operator*(const vec3& v) const { return translation + (rotation * (v * scaling)); }

    void translate(const vec3& v) { translation += rotation * (v * scaling); }
    void rotate(const quat& q) { rotation *= q; }
    void scale(const vec3& s) { scaling *= s; }

    mat4 toMat4() const
    {
        mat4 M(vec4(1.0f, 0.0f, 0.0f, 0.0f), vec4(0.0f, 1.0f, 0.0f, 0.0f), vec4(0.0f, 0.0f, 1.0f, 0.0f),
                vec4(translation, 1.0f));
        M *= WurblPT::toMat4(rotation);
        M.scale(scaling);
        return M;
    }

    mat3 toNormalMatrix() const { return toMat3(rotation); }

    static Transformation fromLookAt(const vec3& eye, const vec3& center, const vec3& up = vec3(0.0f, 1.0f, 0.0f))
    {
        vec3 f = normalize(center - eye);
        vec3 s = normalize(cross(f, up));
        vec3 u = cross(s, f);
        quat rot0 = toQuat(vec3(0.0f, 0.0f, -1.0f), f);
        quat rot1 = toQuat(rot0 * vec3(0.0f, 1.0f, 0.0f), u);
        return Transformation(eye, rot1 * rot0);
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
