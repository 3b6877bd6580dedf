/*
 * scene.hpp -- scene container with the reference's ownership rules and take() overloads
 * (scene.hpp:52-191), plus the flattener that turns it into the POD buffers of
 * include/wurblpt_hip.h.
 *
 * Hitable order = order of take() calls, triangles of an instance in index order, a sphere
 * where it was taken; this is the order the BVH builder sees (scene.hpp:151-162), so it is part
 * of the results contract.
 */
#pragma once

#include <cassert>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../wurblpt_hip.h"
#include "animation.hpp"
#include "bvh.hpp"
#include "envmap.hpp"
#include "material.hpp"
#include "mesh.hpp"
#include "scene_component.hpp"
#include "texture.hpp"

namespace WurblPT {

typedef enum { ColdSpot, HotSpot } HotSpotType;

/* Opaque handle for one hitable of the scene (the reference hands out `const Hitable*`):
 * its position in the scene's hitable list. */
struct Hitable {
    unsigned int index;
};

/* sphere.hpp:35-58: a sphere as a scene component; centre, radius and texture rotation come from
 * the transformation as in HitableSphere's constructor (hitable_sphere.hpp:71-75) */
class Sphere : public SceneComponent
{
public:
    const Material* material;
    const Transformation transformation;
    const int animationIndex;

    Sphere(const Material* m, const Transformation& T) : material(m), transformation(T), animationIndex(-1) {}
    Sphere(const Material* m, int ai = -1) : material(m), transformation(), animationIndex(ai) {}
    Sphere(const vec3& center, float radius, const Material* m) : Sphere(m, Transformation(center, quat::null(), vec3(radius))) {}

    vec3 center() const { return transformation.translation; }
    float radius() const { return max(transformation.scaling); }
};

/* The flattened scene: owns the arrays that a wpt_scene_desc points into. */
class FlatScene
{
public:
    std::vector<wpt_bvh_node> nodes;
    std::vector<wpt_tri_geom> triGeom;
    std::vector<wpt_tri_attr> triAttr;
    std::vector<wpt_instance> instances;
    std::vector<wpt_material> materials;
    std::vector<wpt_texture> textures;
    std::vector<unsigned char> texels;
    std::vector<wpt_hotspot> hotspots;
    std::vector<wpt_sphere> spheres;
    std::vector<wpt_rgl_brdf> rglBrdfs;
    std::vector<float> rglData;
    wpt_envmap envmap;
    size_t bvhLevels = 0;
    std::vector<wpt_animation> animations; /* the scene's animations, then whatever addAnimation() appended (the camera's) */
    std::vector<wpt_keyframe> keyframes;
    std::vector<int> materialSceneIndex; /* per flattened material: Scene::materialIndex() of it, -1 if the scene does not own it */

    /* appends an animation to the pool; -1 if it is not a key frame animation */
    int addAnimation(const Animation* anim)
    {
        const AnimationKeyframes* kf = dynamic_cast<const AnimationKeyframes*>(anim);
        if (!kf)
            return -1;
        wpt_animation a;
        a.first_keyframe = keyframes.size();
        a.keyframe_count = kf->keyframes().size();
        for (const AnimationKeyframes::Keyframe& k : kf->keyframes()) {
            wpt_keyframe r;
            r.t = k.t;
            for (int i = 0; i < 3; i++) {
                r.translation[i] = k.transformation.translation[i];
                r.scaling[i] = k.transformation.scaling[i];
            }
            r.rotation[0] = k.transformation.rotation.x;
            r.rotation[1] = k.transformation.rotation.y;
            r.rotation[2] = k.transformation.rotation.z;
            r.rotation[3] = k.transformation.rotation.w;
            keyframes.push_back(r);
        }
        animations.push_back(a);
        return int(animations.size()) - 1;
    }

    wpt_scene_desc desc() const
    {
        wpt_scene_desc d;
        memset(&d, 0, sizeof(d));
        d.abi_version = WPT_ABI_VERSION;
        d.node_count = nodes.size();
        d.tri_count = triGeom.size();
        d.instance_count = instances.size();
        d.material_count = materials.size();
        d.texture_count = textures.size();
        d.hotspot_count = hotspots.size();
        d.sphere_count = spheres.size();
        d.texel_bytes = texels.size();
        d.nodes = nodes.data();
        d.tri_geom = triGeom.data();
        d.tri_attr = triAttr.data();
        d.instances = instances.data();
        d.materials = materials.data();
        d.textures = textures.data();
        d.texels = texels.data();
        d.hotspots = hotspots.data();
        d.envmap = envmap;
        d.spheres = spheres.data();
        d.rgl_count = rglBrdfs.size();
        d.rgl_data_count = rglData.size();
        d.rgl_brdfs = rglBrdfs.data();
        d.rgl_data = rglData.data();
        d.animation_count = animations.size();
        d.keyframe_count = keyframes.size();
        d.animations = animations.data();
        d.keyframes = keyframes.data();
        return d;
    }
};

class Scene
{
private:
    struct Triangle {
        const MeshInstance* instance;
        unsigned int instanceIndex;
        unsigned int triangle;
    };
    std::vector<std::unique_ptr<Mesh>> _meshes;
    std::vector<std::unique_ptr<SceneComponent>> _components;
    std::vector<const MeshInstance*> _instances;
    std::vector<Triangle> _triangles;             /* the triangle hitables, in take() order */
    std::vector<const Sphere*> _spheres;          /* the sphere hitables, in take() order */
    struct HitableRef {
        unsigned int kind;  /* WPT_NODE_TRIANGLE or WPT_NODE_SPHERE */
        unsigned int index; /* into _triangles or _spheres */
    };
    std::vector<HitableRef> _hitables;            /* all hitables, in take() order */
    std::vector<std::unique_ptr<Hitable>> _handles;
    std::vector<const Hitable*> _hotSpots;
    std::unique_ptr<EnvironmentMap> _envmap;
    std::vector<std::unique_ptr<Animation>> _animations;
    std::vector<const Animation*> _animationsForCaching;
    std::vector<wpt_bvh_node> _bvh;
    size_t _bvhLevels;
    bool _bvhNeedsRebuild;
    float _bvhT0, _bvhT1;
    std::map<const Material*, int> _materialMap;
    std::vector<std::string> _materialNames;
    std::string _error;

    static vec3 vtx(const Mesh* mesh, unsigned int i, size_t offset) { return vec3(mesh->vertices.data() + i * mesh->vertexSize() + offset); }

    /* World-space corners of one hitable, with the arithmetic of hitable_triangle.hpp:193-206 */
    void corners(const Triangle& t, vec3 v[3]) const
    {
        const Mesh* mesh = t.instance->mesh;
        for (int k = 0; k < 3; k++) {
            v[k] = mesh->position(mesh->indices[3 * t.triangle + k]);
            if (!t.instance->transformation.isIdentity())
                v[k] = (t.instance->transformationM * vec4(v[k], 1.0f)).xyz();
        }
    }

public:
    Scene() : _bvhLevels(0), _bvhNeedsRebuild(true), _bvhT0(0.0f), _bvhT1(0.0f) {}

    /* Returns the index mesh instances refer to (scene.hpp:86-91).  Camera animations stay with the camera. */
    int take(Animation* anim)
    {
        _animations.push_back(std::unique_ptr<Animation>(anim));
        _animationsForCaching.push_back(anim);
        return int(_animations.size()) - 1;
    }
    const std::vector<const Animation*>& animations() const { return _animationsForCaching; }

private:
    /* HitableTriangle::aabb (hitable_triangle.hpp:336-393) for an animated instance: the box at t0, widened by
     * intermediate shapes when the rotation changes over the interval.  As there, the second set of corners is
     * taken from the t0 matrix again, so a pure translation between t0 and t1 does not widen the box. */
    AABB animatedBox(const vec3 v[3], int ai, AnimationCache& animationCacheT0, AnimationCache& animationCacheT1) const
    {
        const Transformation& T0 = animationCacheT0.get(ai);
        const Transformation& T1 = animationCacheT1.get(ai);
        const mat4& M0 = animationCacheT0.getM(ai);
        vec3 v00 = (M0 * vec4(v[0], 1.0f)).xyz();
        vec3 v01 = (M0 * vec4(v[1], 1.0f)).xyz();
        vec3 v02 = (M0 * vec4(v[2], 1.0f)).xyz();
        if (T0 == T1)
            return AABB(min(v00, v01, v02), max(v00, v01, v02));
        const mat4& M1 = animationCacheT0.getM(ai);
        vec3 v10 = (M1 * vec4(v[0], 1.0f)).xyz();
        vec3 v11 = (M1 * vec4(v[1], 1.0f)).xyz();
        vec3 v12 = (M1 * vec4(v[2], 1.0f)).xyz();
        AABB box = merge(AABB(min(v00, v01, v02), max(v00, v01, v02)), AABB(min(v10, v11, v12), max(v10, v11, v12)));
        if (T0.rotation != T1.rotation) {
            unsigned int samples = 4;
            float cosHalfAngle = dot(vec4(T0.rotation.x, T0.rotation.y, T0.rotation.z, T0.rotation.w),
                    vec4(T1.rotation.x, T1.rotation.y, T1.rotation.z, T1.rotation.w));
            if (std::abs(cosHalfAngle) < 1.0f)
                samples += degrees(std::acos(cosHalfAngle)) * 2.0f;
            for (unsigned int i = 1; i < samples - 1; i++) {
                float alpha = i / (samples - 1.0f);
                Transformation T = mix(T0, T1, alpha);
                vec3 vt0 = T * v[0];
                vec3 vt1 = T * v[1];
                vec3 vt2 = T * v[2];
                box = merge(box, AABB(min(vt0, vt1, vt2), max(vt0, vt1, vt2)));
            }
        }
        return box;
    }

public:

    Mesh* take(Mesh* mesh)
    {
        _meshes.push_back(std::unique_ptr<Mesh>(mesh));
        return mesh;
    }

    Texture* take(Texture* tex)
    {
        _components.push_back(std::unique_ptr<SceneComponent>(tex));
        return tex;
    }

    Material* take(Material* mat, const std::string& name = std::string())
    {
        _components.push_back(std::unique_ptr<SceneComponent>(mat));
        _materialMap.insert(std::pair<const Material*, int>(mat, _materialNames.size()));
        _materialNames.push_back(name);
        return mat;
    }

    std::vector<const Hitable*> take(MeshInstance* instance, HotSpotType hotSpotType = ColdSpot)
    {
        std::vector<const Hitable*> handles;
        if (instance->animationIndex >= int(_animations.size()))
            _error = "a mesh instance refers to an animation the scene does not have";
        unsigned int instanceIndex = _instances.size();
        _instances.push_back(instance);
        size_t n = instance->mesh->triangleCount();
        _triangles.reserve(_triangles.size() + n);
        _hitables.reserve(_hitables.size() + n);
        for (size_t i = 0; i < n; i++) {
            Hitable* h = new Hitable { (unsigned int)(_hitables.size()) };
            _handles.push_back(std::unique_ptr<Hitable>(h));
            _hitables.push_back(HitableRef { WPT_NODE_TRIANGLE, (unsigned int)(_triangles.size()) });
            _triangles.push_back(Triangle { instance, instanceIndex, (unsigned int)(i) });
            handles.push_back(h);
        }
        if (hotSpotType == HotSpot)
            _hotSpots.insert(_hotSpots.end(), handles.begin(), handles.end());
        _components.push_back(std::unique_ptr<SceneComponent>(instance));
        _bvhNeedsRebuild = true;
        return handles;
    }

    /* scene.hpp:127-149 for a component that creates one HitableSphere (sphere.hpp:60-63) */
    std::vector<const Hitable*> take(Sphere* sphere, HotSpotType hotSpotType = ColdSpot)
    {
        if (sphere->animationIndex >= int(_animations.size()))
            _error = "a sphere refers to an animation the scene does not have";
        Hitable* h = new Hitable { (unsigned int)(_hitables.size()) };
        _handles.push_back(std::unique_ptr<Hitable>(h));
        _hitables.push_back(HitableRef { WPT_NODE_SPHERE, (unsigned int)(_spheres.size()) });
        _spheres.push_back(sphere);
        std::vector<const Hitable*> handles(1, h);
        if (hotSpotType == HotSpot)
            _hotSpots.push_back(h);
        _components.push_back(std::unique_ptr<SceneComponent>(sphere));
        _bvhNeedsRebuild = true;
        return handles;
    }

    /* scene.hpp:113-125: a component handed over through its base class (wurblpt-furnace-test.cpp:52-67 keeps its
     * spheres as SceneComponent*).  The kernels know the components that create triangles and spheres; any other
     * kind is owned as usual and reported by updateBVH() / mcpt(), which refuse the scene: there is no CPU fallback
     * that could ask a user-defined component for its hitables. */
    std::vector<const Hitable*> take(SceneComponent* component, HotSpotType hotSpotType = ColdSpot)
    {
        if (MeshInstance* instance = dynamic_cast<MeshInstance*>(component))
            return take(instance, hotSpotType);
        if (Sphere* sphere = dynamic_cast<Sphere*>(component))
            return take(sphere, hotSpotType);
        _error = "a scene component of a kind the HIP kernels do not know was taken as hitable";
        _components.push_back(std::unique_ptr<SceneComponent>(component));
        return std::vector<const Hitable*>();
    }

    EnvironmentMap* take(EnvironmentMap* envmap)
    {
        _envmap = std::unique_ptr<EnvironmentMap>(envmap);
        return envmap;
    }

    bool bvhNeedsUpdate(float t0 = 0.0f, float t1 = 0.0f) const
    {
        return _bvhNeedsRebuild || (_animations.size() > 0 && (_bvhT0 > t0 || _bvhT1 < t1));
    }

    void updateBVH(float t0 = 0.0f, float t1 = 0.0f)
    {
        if (!bvhNeedsUpdate(t0, t1)) {
            fprintf(stderr, "Bounding volume hierarchy does not need updating.\n");
            return;
        }
        fprintf(stderr, "Building bounding volume hierarchy for %zu hitables\n", _hitables.size());
        std::vector<AABB> boxes(_hitables.size());
        AnimationCache animationCacheT0(animations(), t0);
        AnimationCache animationCacheT1(animations(), t1);
        for (size_t i = 0; i < _hitables.size(); i++) {
            if (_hitables[i].kind == WPT_NODE_TRIANGLE) {
                vec3 v[3];
                const Triangle& tri = _triangles[_hitables[i].index];
                corners(tri, v);
                if (tri.instance->animationIndex >= 0)
                    boxes[i] = animatedBox(v, tri.instance->animationIndex, animationCacheT0, animationCacheT1);
                else
                    boxes[i] = AABB(min(v[0], v[1], v[2]), max(v[0], v[1], v[2]));
            } else {
                /* HitableSphere::aabb (hitable_sphere.hpp:88-91) */
                const Sphere* sp = _spheres[_hitables[i].index];
                if (sp->animationIndex < 0) {
                    boxes[i] = AABB(sp->center() - vec3(sp->radius()), sp->center() + vec3(sp->radius()));
                } else { /* :98-106 */
                    const Transformation& T0 = animationCacheT0.get(sp->animationIndex);
                    const Transformation& T1 = animationCacheT1.get(sp->animationIndex);
                    vec3 c0 = T0 * sp->center();
                    vec3 c1 = T1 * sp->center();
                    float r0 = max(T0.scaling) * sp->radius();
                    float r1 = max(T1.scaling) * sp->radius();
                    boxes[i] = merge(AABB(c0 - vec3(r0), c0 + vec3(r0)), AABB(c1 - vec3(r1), c1 + vec3(r1)));
                }
            }
        }
        BVHBuilder builder(boxes);
        _bvh = builder.build(&_bvhLevels);
        /* the builder's leaves refer to the hitable list: make them refer to their own array */
        for (wpt_bvh_node& nd : _bvh) {
            if (nd.kind == WPT_NODE_TRIANGLE) {
                const HitableRef& r = _hitables[nd.link];
                nd.kind = r.kind;
                nd.link = r.index;
            }
        }
        fprintf(stderr, "Linearized bounding volume hierarchy with %zu nodes on %zu levels\n", _bvh.size(), _bvhLevels);
        _bvhNeedsRebuild = false;
        _bvhT0 = t0;
        _bvhT1 = t1;
    }

    const std::vector<const Hitable*>& hotSpots() const { return _hotSpots; }
    const EnvironmentMap* environmentMap() const { return _envmap.get(); }
    const std::vector<wpt_bvh_node>& bvhNodes() const { return _bvh; }
    size_t hitableCount() const { return _hitables.size(); }

    int materialIndex(const Material* mat) const
    {
        auto it = _materialMap.find(mat);
        return it == _materialMap.end() ? -1 : it->second;
    }
    const std::vector<std::string>& materialNames() const { return _materialNames; }

    /* Flatten into POD buffers.  Returns false (with a message) when the scene uses something
     * that the device path does not know. */
    bool flatten(FlatScene& out, std::string* error = nullptr) const
    {
        auto fail = [&](const std::string& msg) {
            if (error)
                *error = msg;
            return false;
        };
        if (!_error.empty())
            return fail(_error);
        if (_bvhNeedsRebuild)
            return fail("Scene::updateBVH() must run before rendering");
        FlattenContext ctx;
        out = FlatScene();
        out.nodes = _bvh;
        out.bvhLevels = _bvhLevels;
        for (const Animation* anim : _animationsForCaching)
            if (out.addAnimation(anim) < 0)
                return fail("only key frame animations (AnimationKeyframes) can go to the device");
        out.instances.resize(_instances.size());
        for (size_t i = 0; i < _instances.size(); i++) {
            const MeshInstance* inst = _instances[i];
            wpt_instance& r = out.instances[i];
            memset(&r, 0, sizeof(r));
            for (int k = 0; k < 9; k++)
                r.N[k] = inst->transformationN.values[k];
            int m = ctx.indexOf(inst->material);
            if (m < 0)
                return fail(ctx.error);
            r.material = m;
            r.flags = (inst->mesh->haveTexCoords ? WPT_TRI_HAVE_TEXCOORDS : 0) | (inst->mesh->haveTangents ? WPT_TRI_HAVE_TANGENTS : 0)
                | (inst->transformation.isIdentity() ? 0 : WPT_TRI_TRANSFORM) | (inst->animationIndex >= 0 ? WPT_TRI_ANIMATE : 0);
            r.animation = inst->animationIndex >= 0 ? inst->animationIndex : -1;
        }
        out.triGeom.resize(_triangles.size());
        out.triAttr.resize(_triangles.size());
        for (size_t i = 0; i < _triangles.size(); i++) {
            const Triangle& t = _triangles[i];
            const Mesh* mesh = t.instance->mesh;
            wpt_tri_geom& g = out.triGeom[i];
            wpt_tri_attr& a = out.triAttr[i];
            memset(&a, 0, sizeof(a));
            vec3 v[3];
            corners(t, v);
            for (int k = 0; k < 3; k++) {
                g.v0[k] = v[0][k];
                g.v1[k] = v[1][k];
                g.v2[k] = v[2][k];
            }
            g.instance = t.instanceIndex;
            g.material = out.instances[t.instanceIndex].material;
            g.flags = out.instances[t.instanceIndex].flags;
            float* nrm[3] = { a.n0, a.n1, a.n2 };
            float* tc[3] = { a.tc0, a.tc1, a.tc2 };
            float* tan[3] = { a.t0, a.t1, a.t2 };
            for (int c = 0; c < 3; c++) {
                unsigned int vi = mesh->indices[3 * t.triangle + c];
                const float* src = mesh->vertices.data() + vi * mesh->vertexSize();
                for (int k = 0; k < 3; k++)
                    nrm[c][k] = src[Mesh::normalOffset + k];
                if (mesh->haveTexCoords)
                    for (int k = 0; k < 2; k++)
                        tc[c][k] = src[Mesh::texcoordOffset + k];
                if (mesh->haveTangents)
                    for (int k = 0; k < 3; k++)
                        tan[c][k] = src[Mesh::tangentOffset + k];
            }
        }
        out.spheres.resize(_spheres.size());
        for (size_t i = 0; i < _spheres.size(); i++) {
            const Sphere* sp = _spheres[i];
            wpt_sphere& r = out.spheres[i];
            memset(&r, 0, sizeof(r));
            for (int k = 0; k < 3; k++)
                r.center[k] = sp->center()[k];
            r.radius = sp->radius();
            r.rotation[0] = sp->transformation.rotation.x;
            r.rotation[1] = sp->transformation.rotation.y;
            r.rotation[2] = sp->transformation.rotation.z;
            r.rotation[3] = sp->transformation.rotation.w;
            int m = ctx.indexOf(sp->material);
            if (m < 0)
                return fail(ctx.error);
            r.material = m;
            r.animation = sp->animationIndex >= 0 ? sp->animationIndex : -1;
        }
        out.hotspots.resize(_hotSpots.size());
        for (size_t i = 0; i < _hotSpots.size(); i++) {
            const HitableRef& ref = _hitables[_hotSpots[i]->index];
            wpt_hotspot& h = out.hotspots[i];
            memset(&h, 0, sizeof(h));
            h.animation = -1;
            h.prim = ref.index;
            if (ref.kind == WPT_NODE_SPHERE) {
                h.kind = WPT_HOTSPOT_SPHERE;
                continue;
            }
            h.kind = WPT_HOTSPOT_TRIANGLE;
            const Triangle& t = _triangles[ref.index];
            h.animation = t.instance->animationIndex >= 0 ? t.instance->animationIndex : -1;
            const Mesh* mesh = t.instance->mesh;
            h.transform = t.instance->transformation.isIdentity() ? 0 : 1;
            float* p[3] = { h.p0, h.p1, h.p2 };
            for (int c = 0; c < 3; c++) {
                vec3 q = mesh->position(mesh->indices[3 * t.triangle + c]);
                for (int k = 0; k < 3; k++)
                    p[c][k] = q[k];
            }
            for (int k = 0; k < 16; k++)
                h.M[k] = t.instance->transformationM.values[k];
        }
        memset(&out.envmap, 0, sizeof(out.envmap));
        out.envmap.tex = -1;
        if (_envmap) {
            if (!_envmap->describe(out.envmap, ctx))
                return fail(ctx.error.empty() ? "an EnvironmentMap subclass that the device path does not know is used" : ctx.error);
        }
        out.materials = ctx.materials;
        out.materialSceneIndex.assign(out.materials.size(), -1);
        for (const auto& entry : ctx.materialIndex)
            if (entry.second >= 0 && size_t(entry.second) < out.materialSceneIndex.size())
                out.materialSceneIndex[entry.second] = materialIndex(entry.first);
        out.rglBrdfs = ctx.rglBrdfs;
        out.rglData = ctx.rglData;
        out.textures = ctx.textures;
        out.texels = ctx.texels;
        return true;
    }
};

}
