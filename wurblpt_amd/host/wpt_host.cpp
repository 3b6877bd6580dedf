/*
 * wpt_host.cpp -- C entry points over the host-side C++ API (the headers under include/wurblpt) so that
 * Python (tests/, bench.py) can build the benchmark scenes, flatten them and hand the
 * resulting wpt_scene_desc to the device library.  Pure CPU code; no HIP here.
 *
 * The scenes are the BASELINE.json configs restated from the reference applications:
 *   cornell   wurblpt-cornellbox/wurblpt-cornellbox.cpp:44-227 (scene), :251-260 (camera)
 */
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/wurblpt/camera.hpp"
#include "../../include/wurblpt/generator.hpp"
#include "../../include/wurblpt/scene.hpp"
#include "../../include/wurblpt/sensor.hpp"

using namespace WurblPT;

struct wpt_host_scene {
    std::unique_ptr<Scene> scenePtr { new Scene };
    Scene& scene = *scenePtr;
    FlatScene flat;
    wpt_scene_desc desc;
    wpt_camera camera;
    std::string error;
    float vfov = 0.0f; /* the look-at camera the scene was finished with */
    vec3 from, at;
    std::shared_ptr<Camera> cameraObject; /* the Camera the wpt_camera record was made from */
};

namespace {

/* One quad of the Cornell box: 4 positions, 4 normals, texcoords (0,0) (1,0) (1,1) (0,1)
 * unless given, two triangles 0 1 2 / 0 2 3 */
void addQuad(Scene& scene, const Material* material, const float (&p)[4][3], const float (&n)[4][3],
        HotSpotType hotSpot = ColdSpot, const float (*tc)[2] = nullptr)
{
    static const float defaultTc[4][2] = { { 0.0f, 0.0f }, { 1.0f, 0.0f }, { 1.0f, 1.0f }, { 0.0f, 1.0f } };
    if (!tc)
        tc = defaultTc;
    std::vector<vec3> pos, nrm;
    std::vector<vec2> uv;
    for (int i = 0; i < 4; i++) {
        pos.push_back(vec3(p[i][0], p[i][1], p[i][2]));
        nrm.push_back(vec3(n[i][0], n[i][1], n[i][2]));
        uv.push_back(vec2(tc[i][0], tc[i][1]));
    }
    scene.take(new MeshInstance(scene.take(new Mesh(pos, nrm, uv, { 0, 1, 2, 0, 2, 3 })), material), hotSpot);
}

void addQuadN(Scene& scene, const Material* material, const float (&p)[4][3], float nx, float ny, float nz,
        HotSpotType hotSpot = ColdSpot, const float (*tc)[2] = nullptr)
{
    const float n[4][3] = { { nx, ny, nz }, { nx, ny, nz }, { nx, ny, nz }, { nx, ny, nz } };
    addQuad(scene, material, p, n, hotSpot, tc);
}

/* wurblpt-cornellbox.cpp:44-227; shortObjectType 0 (box) only, materials 0 = white,
 * tall box 1 = GGX metal, short box 2 = glass */
bool buildCornell(Scene& scene, int tallBoxMaterialType, int shortObjectType, int shortObjectMaterialType, std::string& error)
{
    if (shortObjectType != 0 || shortObjectMaterialType == 1) {
        error = "cornell: only the box short object with white or glass material is built here";
        return false;
    }
    Material* white = scene.take(new MaterialLambertian(vec3(0.725f, 0.71f, 0.68f)));
    Material* red = scene.take(new MaterialLambertian(vec3(0.63f, 0.065f, 0.05f)));
    Material* green = scene.take(new MaterialLambertian(vec3(0.14f, 0.45f, 0.091f)));
    Material* light = scene.take(new LightDiffuse(vec3(4.0f)));
    Material* metal = scene.take(new MaterialGGX(vec3(1.0f), vec2(0.04f)));
    Material* glass = scene.take(new MaterialGlass(vec3(0.2f), 1.5f));
    Material* tall = (tallBoxMaterialType == 0 ? white : metal);
    Material* shortM = (shortObjectMaterialType == 0 ? white : glass);

    { /* left wall */
        const float p[4][3] = { { -1.01f, 0.0f, 0.99f }, { -0.99f, 0.0f, -1.04f }, { -1.02f, 1.99f, -1.04f }, { -1.02f, 1.99f, 0.99f } };
        const float n[4][3] = { { 0.9999874f, 0.005025057f, 0.0f }, { 0.9998379f, 0.01507292f, 0.009850611f },
            { 0.9999874f, 0.005025057f, 0.0f }, { 0.9999874f, 0.005025057f, 0.0f } };
        addQuad(scene, red, p, n);
    }
    { /* right wall */
        const float p[4][3] = { { 1.0f, 0.0f, -1.04f }, { 1.0f, 0.0f, 0.99f }, { 1.0f, 1.99f, 0.99f }, { 1.0f, 1.99f, -1.04f } };
        addQuadN(scene, green, p, -1.0f, 0.0f, 0.0f);
    }
    { /* floor */
        const float p[4][3] = { { -1.01f, 0.0f, 0.99f }, { 1.0f, 0.0f, 0.99f }, { 1.0f, 0.0f, -1.04f }, { -0.99f, 0.0f, -1.04f } };
        addQuadN(scene, white, p, 0.0f, 1.0f, 0.0f);
    }
    { /* ceiling */
        const float p[4][3] = { { -1.02f, 1.99f, 0.99f }, { -1.02f, 1.99f, -1.04f }, { 1.0f, 1.99f, -1.04f }, { 1.0f, 1.99f, 0.99f } };
        addQuadN(scene, white, p, 0.0f, -1.0f, 0.0f);
    }
    { /* back wall */
        const float p[4][3] = { { -0.99f, 0.0f, -1.04f }, { 1.0f, 0.0f, -1.04f }, { 1.0f, 1.99f, -1.04f }, { -1.02f, 1.99f, -1.04f } };
        addQuadN(scene, white, p, 0.0f, 0.0f, 1.0f);
    }
    /* short box: left, right, floor, ceiling, back, front */
    {
        const float p[4][3] = { { -0.05f, 0.0f, 0.57f }, { -0.05f, 0.6f, 0.57f }, { 0.13f, 0.6f, 0.0f }, { 0.13f, 0.0f, 0.0f } };
        addQuadN(scene, shortM, p, -0.9535826f, 0.0f, -0.3011314f);
    }
    {
        const float p[4][3] = { { 0.7f, 0.0f, 0.17f }, { 0.7f, 0.6f, 0.17f }, { 0.53f, 0.6f, 0.75f }, { 0.53f, 0.0f, 0.75f } };
        addQuadN(scene, shortM, p, 0.9596285f, 0.0f, 0.2812705f);
    }
    {
        const float p[4][3] = { { 0.53f, 0.0f, 0.75f }, { 0.7f, 0.0f, 0.17f }, { 0.13f, 0.0f, 0.0f }, { -0.05f, 0.0f, 0.57f } };
        addQuadN(scene, shortM, p, 0.0f, -1.0f, 0.0f);
    }
    {
        const float p[4][3] = { { 0.53f, 0.6f, 0.75f }, { 0.7f, 0.6f, 0.17f }, { 0.13f, 0.6f, 0.0f }, { -0.05f, 0.6f, 0.57f } };
        addQuadN(scene, shortM, p, 0.0f, 1.0f, 0.0f);
    }
    {
        const float p[4][3] = { { 0.13f, 0.0f, 0.0f }, { 0.13f, 0.6f, 0.0f }, { 0.7f, 0.6f, 0.17f }, { 0.7f, 0.0f, 0.17f } };
        addQuadN(scene, shortM, p, 0.2858051f, 0.0f, -0.9582878f);
    }
    {
        const float p[4][3] = { { 0.53f, 0.0f, 0.75f }, { 0.53f, 0.6f, 0.75f }, { -0.05f, 0.6f, 0.57f }, { -0.05f, 0.0f, 0.57f } };
        addQuadN(scene, shortM, p, -0.2963993f, 0.0f, 0.9550642f);
    }
    /* tall box: left, right, floor, ceiling, back, front */
    {
        const float p[4][3] = { { -0.53f, 0.0f, 0.09f }, { -0.53f, 1.2f, 0.09f }, { -0.71f, 1.2f, -0.49f }, { -0.71f, 0.0f, -0.49f } };
        addQuadN(scene, tall, p, -0.9550642f, 0.0f, 0.2963992f);
    }
    {
        const float p[4][3] = { { -0.14f, 0.0f, -0.67f }, { -0.14f, 1.2f, -0.67f }, { 0.04f, 1.2f, -0.09f }, { 0.04f, 0.0f, -0.09f } };
        addQuadN(scene, tall, p, 0.9550642f, 0.0f, -0.2963992f);
    }
    {
        const float p[4][3] = { { -0.53f, 0.0f, 0.09f }, { 0.04f, 0.0f, -0.09f }, { -0.14f, 0.0f, -0.67f }, { -0.71f, 0.0f, -0.49f } };
        addQuadN(scene, tall, p, 0.0f, -1.0f, 0.0f);
    }
    {
        const float p[4][3] = { { -0.53f, 1.2f, 0.09f }, { 0.04f, 1.2f, -0.09f }, { -0.14f, 1.2f, -0.67f }, { -0.71f, 1.2f, -0.49f } };
        addQuadN(scene, tall, p, 0.0f, 1.0f, 0.0f);
    }
    {
        const float p[4][3] = { { -0.71f, 0.0f, -0.49f }, { -0.71f, 1.2f, -0.49f }, { -0.14f, 1.2f, -0.67f }, { -0.14f, 0.0f, -0.67f } };
        addQuadN(scene, tall, p, -0.3011314f, 0.0f, -0.9535826f);
    }
    {
        const float p[4][3] = { { 0.04f, 0.0f, -0.09f }, { 0.04f, 1.2f, -0.09f }, { -0.53f, 1.2f, 0.09f }, { -0.53f, 0.0f, 0.09f } };
        addQuadN(scene, tall, p, 0.3011314f, 0.0f, 0.9535826f);
    }
    { /* light source, the only hot spot */
        const float light_y = 1.98f;
        const float p[4][3] = { { -0.24f, light_y, 0.16f }, { -0.24f, light_y, -0.22f }, { 0.23f, light_y, -0.22f }, { 0.23f, light_y, 0.16f } };
        const float tc[4][2] = { { 0.0f, 1.0f }, { 0.0f, 0.0f }, { 1.0f, 0.0f }, { 1.0f, 1.0f } };
        addQuadN(scene, light, p, 0.0f, -1.0f, 0.0f, HotSpot, tc);
    }
    return true;
}

/* n random triangles in the unit cube around the origin, one Lambertian material, a light
 * quad on top as hot spot: BVH build / traversal stress scene for the tests */
void buildRandomTriangles(Scene& scene, unsigned int n, unsigned int seed, bool withTexcoords)
{
    std::mt19937 rng(seed);
    auto u01 = [&rng]() { return float(rng() >> 8) * (1.0f / 16777216.0f); };
    Material* grey = scene.take(new MaterialLambertian(vec3(0.6f, 0.6f, 0.6f)));
    Material* light = scene.take(new LightDiffuse(vec3(6.0f)));
    std::vector<vec3> pos, nrm;
    std::vector<vec2> uv;
    std::vector<unsigned int> ind;
    for (unsigned int i = 0; i < n; i++) {
        vec3 c(u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f);
        float s = 0.02f + 0.2f * u01();
        vec3 v[3];
        for (int k = 0; k < 3; k++)
            v[k] = c + s * vec3(u01() - 0.5f, u01() - 0.5f, u01() - 0.5f);
        vec3 fn = cross(v[1] - v[0], v[2] - v[0]);
        if (!(dot(fn, fn) > 0.0f))
            fn = vec3(0.0f, 0.0f, 1.0f);
        fn = normalize(fn);
        for (int k = 0; k < 3; k++) {
            pos.push_back(v[k]);
            nrm.push_back(fn);
            if (withTexcoords)
                uv.push_back(vec2(u01(), u01()));
            ind.push_back(3 * i + k);
        }
    }
    scene.take(new MeshInstance(scene.take(new Mesh(pos, nrm, uv, ind)), grey));
    Transformation T(vec3(0.0f, 1.5f, 0.0f), toQuat(radians(90.0f), vec3(1.0f, 0.0f, 0.0f)), vec3(0.7f));
    scene.take(new MeshInstance(scene.take(generateQuad()), light, T), HotSpot);
}

wpt_host_scene* finishSceneOf(wpt_host_scene* hs, Scene& scene, unsigned int width, unsigned int height, float vfovRadians,
        const vec3& from, const vec3& at, float aperture, float focusDist)
{
    scene.updateBVH();
    if (!scene.flatten(hs->flat, &hs->error)) {
        fprintf(stderr, "wpt_host: %s\n", hs->error.c_str());
        delete hs;
        return nullptr;
    }
    hs->desc = hs->flat.desc();
    Optics optics(Projection(vfovRadians, float(width) / height), LensDistortion(), LensDepthOfField(aperture, focusDist));
    Camera camera(optics, Transformation::fromLookAt(from, at, vec3(0.0f, 1.0f, 0.0f)));
    camera.describe(hs->camera);
    hs->cameraObject.reset(new Camera(camera));
    hs->vfov = vfovRadians;
    hs->from = from;
    hs->at = at;
    return hs;
}

wpt_host_scene* finishScene(wpt_host_scene* hs, unsigned int width, unsigned int height, float vfovRadians,
        const vec3& from, const vec3& at, float aperture = 0.0f, float focusDist = 1.0f)
{
    return finishSceneOf(hs, hs->scene, width, height, vfovRadians, from, at, aperture, focusDist);
}

}

/* used by the other files of this library */
Scene& wptHostSceneOf(wpt_host_scene* hs) { return *hs->scenePtr; } /* not hs->scene: wptHostFinish replaces the scene */
const Camera& wptHostCameraObjectOf(const wpt_host_scene* hs) { return *hs->cameraObject; }
void wptHostCameraOf(const wpt_host_scene* hs, float& vfovRadians, vec3& from, vec3& at)
{
    vfovRadians = hs->vfov;
    from = hs->from;
    at = hs->at;
}

/* takes ownership of `scene` */
wpt_host_scene* wptHostFinish(Scene* scene, unsigned int width, unsigned int height, float vfovRadians, const vec3& from,
        const vec3& at, float aperture, float focusDist)
{
    wpt_host_scene* hs = new wpt_host_scene;
    hs->scenePtr.reset(scene);
    return finishSceneOf(hs, *scene, width, height, vfovRadians, from, at, aperture, focusDist);
}

/* the same for a scene with animations: BVH over [t0, t1], camera from an animation (owned by the camera) described
 * at t0, its key frames appended to the scene's pool */
wpt_host_scene* wptHostFinishAnimated(Scene* scene, unsigned int width, unsigned int height, float vfovRadians,
        const Animation* cameraAnimation, float t0, float t1, float aperture, float focusDist)
{
    wpt_host_scene* hs = new wpt_host_scene;
    hs->scenePtr.reset(scene);
    scene->updateBVH(t0, t1);
    if (!scene->flatten(hs->flat, &hs->error)) {
        fprintf(stderr, "wpt_host: %s\n", hs->error.c_str());
        delete hs;
        return nullptr;
    }
    Optics optics(Projection(vfovRadians, float(width) / height), LensDistortion(), LensDepthOfField(aperture, focusDist));
    Camera camera(optics, cameraAnimation);
    camera.describe(hs->camera, t0);
    hs->camera.animation = hs->flat.addAnimation(cameraAnimation);
    hs->cameraObject.reset(new Camera(camera));
    hs->desc = hs->flat.desc();
    hs->vfov = vfovRadians;
    hs->from = camera.at(t0).lookFrom();
    hs->at = camera.at(t0).lookAt();
    return hs;
}

extern "C" {

/* Cornell box of wurblpt-cornellbox.cpp with its camera (:251-260) */
wpt_host_scene* wpt_host_cornell(int tallBoxMaterial, int shortObjectType, int shortObjectMaterial, unsigned int width, unsigned int height)
{
    wpt_host_scene* hs = new wpt_host_scene;
    if (!buildCornell(hs->scene, tallBoxMaterial, shortObjectType, shortObjectMaterial, hs->error)) {
        fprintf(stderr, "wpt_host: %s\n", hs->error.c_str());
        delete hs;
        return nullptr;
    }
    return finishScene(hs, width, height, radians(50.0f), vec3(0.0f, 1.0f, 3.2f), vec3(0.0f, 1.0f, -1.0f));
}

wpt_host_scene* wpt_host_random_triangles(unsigned int n, unsigned int seed, int withTexcoords, unsigned int width,
        unsigned int height, float aperture)
{
    wpt_host_scene* hs = new wpt_host_scene;
    buildRandomTriangles(hs->scene, n, seed, withTexcoords != 0);
    return finishScene(hs, width, height, radians(60.0f), vec3(0.3f, 0.4f, 3.5f), vec3(0.0f, 0.0f, 0.0f), aperture, 3.5f);
}

/* Gives the scene's camera a lens distortion: model 1 = RadialAndPlanar(k1, k2, p1, p2), 2 = RadialOnly(k1, k2, k3),
 * 3 = OpenCV(k1, k2, k3, p1, p2), 0 = none; constructors and helper of optics.hpp:155-212 */
void wpt_host_scene_set_distortion(wpt_host_scene* hs, int model, float k1, float k2, float k3, float p1, float p2)
{
    wpt_camera& c = hs->camera;
    const Projection proj(c.l, c.r, c.b, c.t);
    const LensDistortion ld = model == 1 ? LensDistortion(k1, k2, p1, p2) : model == 2 ? LensDistortion(k1, k2, k3)
        : model == 3 ? LensDistortion(k1, k2, k3, p1, p2) : LensDistortion();
    Camera cam(Optics(proj, ld, LensDepthOfField(2.0f * c.lens_radius, c.focus_dist)),
            Transformation(vec3(c.translation), quat(c.rotation[0], c.rotation[1], c.rotation[2], c.rotation[3]), vec3(c.scaling)));
    wpt_camera out;
    memset(&out, 0, sizeof(out));
    if (cam.describe(out)) {
        /* keep the depth-of-field values bit for bit */
        out.lens_radius = c.lens_radius;
        out.focus_dist = c.focus_dist;
        c = out;
    }
}

/* Camera::surroundMode (0 off, 1 = 180 degrees, 2 = 360 degrees) and ::stereoscopicDistance of the scene's camera */
void wpt_host_scene_set_camera_mode(wpt_host_scene* hs, int surroundMode, float stereoscopicDistance)
{
    hs->camera.surround_mode = surroundMode == 1 ? WPT_SURROUND_180 : surroundMode == 2 ? WPT_SURROUND_360 : WPT_SURROUND_OFF;
    hs->camera.stereoscopic_distance = stereoscopicDistance;
}

const wpt_scene_desc* wpt_host_scene_desc(const wpt_host_scene* hs) { return &hs->desc; }
const wpt_camera* wpt_host_scene_camera(const wpt_host_scene* hs) { return &hs->camera; }
unsigned int wpt_host_scene_bvh_levels(const wpt_host_scene* hs) { return hs->flat.bvhLevels; }
/* Scene::materialIndex() of every flattened material (material_count ints) */
void wpt_host_scene_material_scene_index(const wpt_host_scene* hs, int* out)
{
    for (size_t i = 0; i < hs->flat.materialSceneIndex.size(); i++)
        out[i] = hs->flat.materialSceneIndex[i];
}
void wpt_host_scene_free(wpt_host_scene* hs) { delete hs; }

/* Default Parameters (wurblpt.hpp:89-95) and SensorRGB gates (sensor_rgb.hpp:41-44) */
void wpt_host_default_params(wpt_params* p)
{
    p->max_path_components = 128;
    p->rr_threshold = 1.0f;
    p->randomize_ray_over_pixel = 1;
    p->min_hit_distance = 0.00001f;
    p->min_dist_to_light = 0.0f;
    p->max_dist_to_light = std::numeric_limits<float>::max();
    p->min_path_len = 0.0f;
    p->max_path_len = std::numeric_limits<float>::max();
    p->t0 = 0.0f;
    p->t1 = 0.0f;
}

/* ---- building blocks, exposed for parity tests against the reference's golden vectors ---- */

/* A generated mesh (include/wurblpt/generator.hpp) as arrays: kind 0 quad, 1 cube, 2 cube side (a = side), 3 disk (f = inner
 * radius), 4 sphere, 5 cylinder, 6 closed cylinder, 7 cone, 8 closed cone, 9 torus (f = inner radius), 10 tetrahedron,
 * 11 octahedron, 12 icosahedron; a, b = slices / stacks (or sides / rings) where the shape has them.  Writes up to
 * `capacity` vertices (11 floats: position, normal, texcoord, tangent) and 3 * capacity indices; returns the vertex
 * count and stores the index count. */
unsigned int wpt_host_generate_mesh(int kind, int a, int b, float f, float* vertices, unsigned int* indices, unsigned int capacity,
        unsigned int* indexCount)
{
    const Transformation T;
    std::unique_ptr<Mesh> mesh;
    switch (kind) {
    case 0: mesh.reset(generateQuad(T, a)); break;
    case 1: mesh.reset(generateCube(T, a)); break;
    case 2: mesh.reset(generateCubeSide(a, T, b)); break;
    case 3: mesh.reset(generateDisk(T, f, a)); break;
    case 4: mesh.reset(generateSphere(T, a, b)); break;
    case 5: mesh.reset(generateCylinder(T, a)); break;
    case 6: mesh.reset(generateClosedCylinder(T, a)); break;
    case 7: mesh.reset(generateCone(T, a, b)); break;
    case 8: mesh.reset(generateClosedCone(T, a, b)); break;
    case 9: mesh.reset(generateTorus(T, f, a, b)); break;
    case 10: mesh.reset(generateTetrahedron(T)); break;
    case 11: mesh.reset(generateOctahedron(T)); break;
    default: mesh.reset(generateIcosahedron(T)); break;
    }
    const unsigned int n = mesh->vertexCount();
    *indexCount = mesh->indices.size();
    for (unsigned int i = 0; i < n && i < capacity; i++) {
        float* o = vertices + 11 * i;
        const vec3 p = mesh->position(i), nr = mesh->normal(i);
        const vec2 tc = mesh->haveTexCoords ? mesh->texcoord(i) : vec2(0.0f);
        const vec3 tg = mesh->haveTangents ? mesh->tangent(i) : vec3(0.0f);
        for (int k = 0; k < 3; k++) {
            o[k] = p[k];
            o[3 + k] = nr[k];
            o[8 + k] = tg[k];
        }
        o[6] = tc[0];
        o[7] = tc[1];
    }
    for (size_t i = 0; i < mesh->indices.size() && i < size_t(3) * capacity; i++)
        indices[i] = mesh->indices[i];
    return n;
}

/* The transformations of the reference's tests/test-transformation.cpp made with include/wurblpt/transformation.hpp
 * and gvm.hpp: per chain the Transformation (10 floats), its toMat4() (16) and the same chain in mat4 operations (16) */
void wpt_host_transformation_chains(float* out)
{
    auto chain = [](int order, Transformation& T, mat4& M, const vec3& tr, const quat& q, const vec3& sc) {
        for (int step = 0; step < 3; step++) {
            const int what = (order >> (2 * step)) & 3; /* 0 translate, 1 rotate, 2 scale */
            if (what == 0) { T = translate(T, tr); M = translate(M, tr); }
            else if (what == 1) { T = rotate(T, q); M = rotate(M, q); }
            else { T = scale(T, sc); M = scale(M, sc); }
        }
    };
    Transformation T[8];
    mat4 M[8];
    for (int i = 0; i < 8; i++)
        M[i] = mat4(1.0f);
    const quat q27 = toQuat(radians(27.0f), vec3(1.0f, 0.0f, 0.0f));
    chain(0 | (1 << 2) | (2 << 4), T[0], M[0], vec3(1, 2, 3), toQuat(radians(15.0f), vec3(1.0f, 1.0f, 0.0f)), vec3(0.5f));
    chain(2 | (1 << 2) | (0 << 4), T[1], M[1], vec3(3, 2, 1), q27, vec3(0.4f));
    chain(0 | (2 << 2) | (1 << 4), T[2], M[2], vec3(3, 2, 1), q27, vec3(0.4f));
    chain(1 | (0 << 2) | (2 << 4), T[3], M[3], vec3(3, 2, 1), q27, vec3(0.4f));
    chain(1 | (2 << 2) | (0 << 4), T[4], M[4], vec3(3, 2, 1), q27, vec3(0.4f));
    chain(2 | (0 << 2) | (1 << 4), T[5], M[5], vec3(3, 2, 1), q27, vec3(0.4f));
    T[6] = T[0] * T[1] * T[2];
    M[6] = M[0] * M[1] * M[2];
    T[7] = T[2] * T[1] * T[0];
    M[7] = M[2] * M[1] * M[0];
    for (int i = 0; i < 8; i++) {
        float* o = out + 42 * i;
        for (int k = 0; k < 3; k++) {
            o[k] = T[i].translation[k];
            o[7 + k] = T[i].scaling[k];
        }
        o[3] = T[i].rotation.x; o[4] = T[i].rotation.y; o[5] = T[i].rotation.z; o[6] = T[i].rotation.w;
        const mat4 TM = T[i].toMat4();
        for (int k = 0; k < 16; k++) {
            o[10 + k] = TM.values[k];
            o[26 + k] = M[i].values[k];
        }
    }
}

/* AnimationKeyframes of include/wurblpt/animation.hpp from `count` key frames (11 floats each: t, translation,
 * rotation xyzw, scaling), evaluated for n cases (t, point): per case the transformation (10 floats), toMat4 (16),
 * toNormalMatrix (9), M * p, N * p, T * p -- the layout of the anim_out golden vector */
void wpt_host_animation_at(const float* keyframes, unsigned int count, int n, const float* in, float* out)
{
    AnimationKeyframes anim;
    /* added in reverse order: addKeyframe() has to sort them */
    for (unsigned int k = count; k-- > 0;) {
        const float* f = keyframes + 11 * k;
        anim.addKeyframe(f[0], Transformation(vec3(f + 1), quat(f[4], f[5], f[6], f[7]), vec3(f + 8)));
    }
    for (int i = 0; i < n; i++) {
        const Transformation T = anim.at(in[4 * i]);
        const mat4 M = T.toMat4();
        const mat3 N = T.toNormalMatrix();
        const vec3 p(in + 4 * i + 1);
        const vec3 viaM = (M * vec4(p, 1.0f)).xyz(), viaN = N * p, viaT = T * p;
        float* o = out + 44 * i;
        for (int k = 0; k < 3; k++) {
            o[k] = T.translation[k];
            o[7 + k] = T.scaling[k];
            o[35 + k] = viaM[k];
            o[38 + k] = viaN[k];
            o[41 + k] = viaT[k];
        }
        o[3] = T.rotation.x; o[4] = T.rotation.y; o[5] = T.rotation.z; o[6] = T.rotation.w;
        for (int k = 0; k < 16; k++)
            o[10 + k] = M.values[k];
        for (int k = 0; k < 9; k++)
            o[26 + k] = N.values[k];
    }
}

/* boxes: n x (lo[3], hi[3]); nodes_out must hold 2n-1 nodes; returns the node count */
unsigned int wpt_host_bvh_build(unsigned int n, const float* boxes, wpt_bvh_node* nodes_out, unsigned int* levels)
{
    std::vector<AABB> b(n);
    for (unsigned int i = 0; i < n; i++)
        b[i] = AABB(vec3(boxes + 6 * i), vec3(boxes + 6 * i + 3));
    BVHBuilder builder(b);
    size_t lv = 0;
    std::vector<wpt_bvh_node> nodes = builder.build(&lv);
    memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(wpt_bvh_node));
    if (levels)
        *levels = lv;
    return nodes.size();
}

void wpt_host_compute_tangents(unsigned int nv, const float* pos, const float* nrm, const float* tc, unsigned int ni,
        const unsigned int* ind, float* out)
{
    std::vector<vec3> p(nv), n(nv);
    std::vector<vec2> t(nv);
    for (unsigned int i = 0; i < nv; i++) {
        p[i] = vec3(pos + 3 * i);
        n[i] = vec3(nrm + 3 * i);
        t[i] = vec2(tc + 2 * i);
    }
    std::vector<unsigned int> idx(ind, ind + ni);
    std::vector<vec3> r = computeTangents(p, n, t, idx);
    for (unsigned int i = 0; i < nv; i++)
        for (int k = 0; k < 3; k++)
            out[3 * i + k] = r[i][k];
}

void wpt_host_compute_normals(unsigned int nv, const float* pos, unsigned int ni, const unsigned int* ind, int source, float* out)
{
    std::vector<vec3> p(nv);
    for (unsigned int i = 0; i < nv; i++)
        p[i] = vec3(pos + 3 * i);
    std::vector<unsigned int> idx(ind, ind + ni);
    std::vector<vec3> r = computeNormals(p, idx, NormalSource(source));
    for (unsigned int i = 0; i < nv; i++)
        for (int k = 0; k < 3; k++)
            out[3 * i + k] = r[i][k];
}

/* Transformation::fromLookAt -> translation(3) rotation(4 xyzw) scaling(3), then toMat4 (16), toNormalMatrix (9) */
void wpt_host_lookat(const float* eye, const float* center, const float* up, float* out35)
{
    Transformation T = Transformation::fromLookAt(vec3(eye), vec3(center), vec3(up));
    float* o = out35;
    for (int k = 0; k < 3; k++)
        *o++ = T.translation[k];
    *o++ = T.rotation.x;
    *o++ = T.rotation.y;
    *o++ = T.rotation.z;
    *o++ = T.rotation.w;
    for (int k = 0; k < 3; k++)
        *o++ = T.scaling[k];
    mat4 M = T.toMat4();
    for (int k = 0; k < 16; k++)
        *o++ = M.values[k];
    mat3 N = T.toNormalMatrix();
    for (int k = 0; k < 9; k++)
        *o++ = N.values[k];
}

/* a general Transformation(t, toQuat(angle, axis), s) -> toMat4 (16) + toNormalMatrix (9) + T*v (3) */
void wpt_host_transformation(const float* t, float angle, const float* axis, const float* s, const float* v, float* out28)
{
    Transformation T(vec3(t), toQuat(angle, vec3(axis)), vec3(s));
    float* o = out28;
    mat4 M = T.toMat4();
    for (int k = 0; k < 16; k++)
        *o++ = M.values[k];
    mat3 N = T.toNormalMatrix();
    for (int k = 0; k < 9; k++)
        *o++ = N.values[k];
    vec3 r = T * vec3(v);
    for (int k = 0; k < 3; k++)
        *o++ = r[k];
}

/* Projection(vfov, aspect) -> l r b t */
void wpt_host_projection(float vfov, float aspect, float* out4)
{
    Projection P(vfov, aspect);
    out4[0] = P.l;
    out4[1] = P.r;
    out4[2] = P.b;
    out4[3] = P.t;
}

}
