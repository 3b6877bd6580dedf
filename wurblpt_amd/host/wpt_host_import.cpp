/*
 * wpt_host_import.cpp -- test and tool entry points around the importer (include/wurblpt/import.hpp,
 * objreader.hpp, imageio.hpp): what the parity tests need to look inside.
 */
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wurblpt/camera.hpp"
#include "../../include/wurblpt/imageio.hpp"
#include "../../include/wurblpt/import.hpp"
#include "../../include/wurblpt/sensor.hpp"
#include "../../include/wurblpt/objreader.hpp"
#include "../../include/wurblpt/postproc.hpp"
#include "../../include/wurblpt/wurblpt.hpp"

using namespace WurblPT;

struct wpt_host_scene;
wpt_host_scene* wptHostFinish(Scene* scene, unsigned int width, unsigned int height, float vfovRadians, const vec3& from,
        const vec3& at, float aperture, float focusDist);

Scene& wptHostSceneOf(wpt_host_scene* hs);
void wptHostCameraOf(const wpt_host_scene* hs, float& vfovRadians, vec3& from, vec3& at);
const Camera& wptHostCameraObjectOf(const wpt_host_scene* hs);

namespace {

void hexFloats(FILE* f, const char* key, const std::vector<float>& v, bool comma = true)
{
    fprintf(f, "\"%s\": [", key);
    for (size_t i = 0; i < v.size(); i++) {
        uint32_t u;
        memcpy(&u, &v[i], 4);
        fprintf(f, "%s\"%08x\"", i ? "," : "", u);
    }
    fprintf(f, "]%s\n", comma ? "," : "");
}

void ints(FILE* f, const char* key, const std::vector<long long>& v, bool comma = true)
{
    fprintf(f, "\"%s\": [", key);
    for (size_t i = 0; i < v.size(); i++)
        fprintf(f, "%s%lld", i ? "," : "", v[i]);
    fprintf(f, "]%s\n", comma ? "," : "");
}

void chars(std::vector<long long>& dst, const std::string& s)
{
    dst.push_back((long long)s.size());
    for (char c : s)
        dst.push_back((unsigned char)c);
}

}

/* Parses an OBJ file with include/wurblpt/objreader.hpp and writes what it produced as JSON with
 * the keys and layouts of the obj_* entries of tests/golden/ref_golden.json (which hold what the
 * reference's vendored tinyobjloader produces for the same file).  Returns 0 on success. */
extern "C" int wpt_host_obj_dump(const char* objFile, const char* jsonFile)
{
    ObjData d;
    bool ok = loadObj(objFile, d);
    FILE* f = fopen(jsonFile, "w");
    if (!f)
        return 2;
    fprintf(f, "{\n");
    ints(f, "obj_valid", std::vector<long long>(1, ok ? 1 : 0));
    hexFloats(f, "obj_vertices", d.vertices);
    hexFloats(f, "obj_normals", d.normals);
    hexFloats(f, "obj_texcoords", d.texcoords);
    std::vector<long long> shapeNames, shapeSizes, indices, materialIds, matStrings;
    std::vector<float> matFloats;
    for (const ObjShape& sh : d.shapes) {
        chars(shapeNames, sh.name);
        shapeSizes.push_back((long long)sh.indices.size());
        for (const ObjIndex& ix : sh.indices) {
            indices.push_back(ix.vertex);
            indices.push_back(ix.normal);
            indices.push_back(ix.texcoord);
        }
        for (int id : sh.materialIds)
            materialIds.push_back(id);
    }
    for (const ObjMaterial& M : d.materials) {
        chars(matStrings, M.name);
        const std::string* names[7] = { &M.diffuseTex, &M.specularTex, &M.shininessTex, &M.bumpTex, &M.alphaTex, &M.emissiveTex, &M.normalTex };
        const ObjTexOpt* opts[7] = { &M.diffuseOpt, &M.specularOpt, &M.shininessOpt, &M.bumpOpt, &M.alphaOpt, &M.emissiveOpt, &M.normalOpt };
        for (int k = 0; k < 3; k++) matFloats.push_back(M.diffuse[k]);
        for (int k = 0; k < 3; k++) matFloats.push_back(M.specular[k]);
        for (int k = 0; k < 3; k++) matFloats.push_back(M.emission[k]);
        for (int k = 0; k < 3; k++) matFloats.push_back(M.transmittance[k]);
        matFloats.push_back(M.shininess);
        matFloats.push_back(M.dissolve);
        matFloats.push_back(M.ior);
        for (int t = 0; t < 7; t++) {
            chars(matStrings, *names[t]);
            for (int k = 0; k < 3; k++) matFloats.push_back(opts[t]->scale[k]);
            for (int k = 0; k < 3; k++) matFloats.push_back(opts[t]->originOffset[k]);
            matFloats.push_back(opts[t]->bumpMultiplier);
        }
    }
    ints(f, "obj_shape_names", shapeNames);
    ints(f, "obj_shape_index_counts", shapeSizes);
    ints(f, "obj_indices", indices);
    ints(f, "obj_material_ids", materialIds);
    ints(f, "obj_material_strings", matStrings);
    hexFloats(f, "obj_material_floats", matFloats, false);
    fprintf(f, "}\n");
    fclose(f);
    if (!ok)
        fprintf(stderr, "wpt_host: %s", d.error.c_str());
    return ok ? 0 : 1;
}

/* Decodes an image file with include/wurblpt/imageio.hpp.  info = width, height, components,
 * component type (0 uint8, 1 uint16, 2 float32); copies at most `capacity` bytes of the array
 * (row 0 = bottom).  Returns the array size in bytes, 0 on failure (message on stderr). */
extern "C" unsigned long long wpt_host_image_load(const char* filename, unsigned int* info, void* data, unsigned long long capacity)
{
    std::string error;
    ArrayContainer img = loadImage(filename, &error);
    if (img.elementCount() == 0) {
        fprintf(stderr, "wpt_host: %s\n", error.c_str());
        return 0;
    }
    if (info) {
        info[0] = img.dimension(0);
        info[1] = img.dimension(1);
        info[2] = img.componentCount();
        info[3] = img.componentType();
    }
    if (data)
        memcpy(data, img.data(), img.dataSize() < capacity ? img.dataSize() : capacity);
    return img.dataSize();
}

/* Writes an array (row 0 = bottom) with saveImage(); 1 on success. */
extern "C" int wpt_host_image_save(const char* filename, unsigned int width, unsigned int height, unsigned int comps,
        unsigned int type, const void* data)
{
    ArrayContainer img(width, height, comps, ComponentType(type));
    memcpy(img.data(), data, img.dataSize());
    std::string error;
    if (!saveImage(img, filename, &error)) {
        fprintf(stderr, "wpt_host: %s\n", error.c_str());
        return 0;
    }
    return 1;
}

/* The output side through include/wurblpt/postproc.hpp: op 0 toSRGB (out uint8 x 3), 1 uniformRationalQuantization(a, b),
 * 2 scaleLuminance(a, b) (out float x comps), 3 maxLuminance (out one float).  1 on success. */
extern "C" int wpt_host_postproc(int op, unsigned int width, unsigned int height, unsigned int comps, const float* in, void* out,
        float a, float b)
{
    try {
        Array<float> img(width, height, comps);
        memcpy(img.data(), in, img.dataSize());
        if (op == 0) {
            Array<uint8_t> r = toSRGB(img);
            memcpy(out, r.data(), r.dataSize());
        } else if (op == 3) {
            *static_cast<float*>(out) = maxLuminance(img);
        } else {
            Array<float> r = op == 1 ? uniformRationalQuantization(img, a, b) : scaleLuminance(img, a, b);
            memcpy(out, r.data(), r.dataSize());
        }
        return 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "wpt_host: %s\n", e.what());
        return 0;
    }
}

/* mcpt() of include/wurblpt/wurblpt.hpp -- the call an application makes -- for a scene of this library and the
 * camera it was finished with: renders width x height x samplesSqrt^2 over the exposure interval [t0, t1] on the
 * device and copies the sensor's frame (row 0 = bottom) to `frame`.  workers > 1: through an MPICoordinator with that
 * many worker threads.  1 on success; failures inside mcpt() abort like the reference's asserts. */
extern "C" int wpt_host_mcpt(wpt_host_scene* hs, unsigned int width, unsigned int height, unsigned int samplesSqrt, float t0, float t1,
        float* frame, unsigned int workers)
{
    Scene& scene = wptHostSceneOf(hs);
    if (scene.bvhNeedsUpdate(t0, t1))
        scene.updateBVH(t0, t1);
    SensorRGB sensor(width, height);
    if (workers > 1) {
        /* the several-devices path of mcpt() (one worker thread per entry, blocks from the shared counter), with the
         * current device named `workers` times where there is only one GPU */
        int device = 0;
        MPICoordinator coordinator(4096, std::vector<int>(workers, device));
        mcpt(coordinator, sensor, wptHostCameraObjectOf(hs), scene, samplesSqrt, t0, t1);
    } else
    mcpt(sensor, wptHostCameraObjectOf(hs), scene, samplesSqrt, t0, t1);
    memcpy(frame, sensor.result().data(), size_t(width) * height * 3 * sizeof(float));
    return 1;
}

/* getGroundTruth() of include/wurblpt/wurblpt.hpp for a scene of this library.  The camera is the one the scene was
 * finished with (its animation gives the camera at tPrev / tNext); prevFromAt / nextFromAt, if given, replace it at
 * those times by a look-at camera (eye and target, 6 floats).  times = t0, tPrev, tNext.  arrays[k]: host array of
 * GroundTruth bit k (width * height * components), or NULL.  1 on success. */
extern "C" int wpt_host_get_ground_truth(wpt_host_scene* hs, unsigned int width, unsigned int height, const float* prevFromAt,
        const float* nextFromAt, const float* times, void* const* arrays)
{
    const Camera& camera = wptHostCameraObjectOf(hs);
    const vec3 up(0.0f, 1.0f, 0.0f);
    Camera cameraPrev(camera.optics, prevFromAt ? Transformation::fromLookAt(vec3(prevFromAt), vec3(prevFromAt + 3), up) : Transformation());
    Camera cameraNext(camera.optics, nextFromAt ? Transformation::fromLookAt(vec3(nextFromAt), vec3(nextFromAt + 3), up) : Transformation());
    unsigned int bits = 0;
    for (int k = 0; k < WPT_GT_ARRAY_COUNT; k++)
        if (arrays[k])
            bits |= 1u << k;
    SensorRGB sensor(width, height);
    Scene& scene = wptHostSceneOf(hs);
    const float t0 = times ? times[0] : 0.0f, tPrev = times ? times[1] : 0.0f, tNext = times ? times[2] : 0.0f;
    if (scene.bvhNeedsUpdate(t0, t0))
        scene.updateBVH(t0, t0);
    GroundTruth gt = getGroundTruth(sensor, camera, prevFromAt ? &cameraPrev : nullptr, nextFromAt ? &cameraNext : nullptr, scene, t0, tPrev,
            tNext, bits);
    void* src[WPT_GT_ARRAY_COUNT];
    gt.arrayPointers(src);
    for (int k = 0; k < WPT_GT_ARRAY_COUNT; k++)
        if (arrays[k])
            memcpy(arrays[k], src[k], size_t(width) * height * wpt_gt_components[k] * 4);
    return 1;
}

/* importIntoScene (include/wurblpt/import.hpp) + a constant environment of the given radiance (0 = none)
 * + a look-at camera.  rotateYDegrees / scale form the import transformation the way
 * wurblpt-sponza.cpp:49-53 does. */
extern "C" wpt_host_scene* wpt_host_import_obj(const char* objFile, unsigned int importBits, float scale, float rotateYDegrees,
        float envRadiance, const float* eye, const float* at, float vfovDegrees, unsigned int width, unsigned int height)
{
    Scene* scene = new Scene;
    const Transformation T(vec3(0.0f), toQuat(radians(rotateYDegrees), vec3(0.0f, 1.0f, 0.0f)), vec3(scale));
    if (!importIntoScene(*scene, objFile, T, importBits)) {
        delete scene;
        return nullptr;
    }
    if (envRadiance > 0.0f) {
        Texture* tex = scene->take(new TextureConstant(vec4(envRadiance)));
        scene->take(new EnvironmentMapEquiRect(tex));
    }
    return wptHostFinish(scene, width, height, radians(vfovDegrees), vec3(eye), vec3(at), 0.0f, 1.0f);
}

/* The same with an environment map from an image file (Radiance HDR, OpenEXR, PFM, ...) as wurblpt-sponza.cpp:46-59 sets
 * one up, and importance sampling of it where importanceN > 0 (wurblpt-envmap.cpp:76). */
extern "C" wpt_host_scene* wpt_host_import_obj_env(const char* objFile, const char* envFile, int importanceN, unsigned int importBits, float scale,
        float rotateYDegrees, const float* eye, const float* at, float vfovDegrees, unsigned int width, unsigned int height)
{
    Scene* scene = new Scene;
    const Transformation T(vec3(0.0f), toQuat(radians(rotateYDegrees), vec3(0.0f, 1.0f, 0.0f)), vec3(scale));
    if (!importIntoScene(*scene, objFile, T, importBits)) {
        delete scene;
        return nullptr;
    }
    Texture* tex = createTextureImage(std::string(envFile));
    if (!tex) {
        delete scene;
        return nullptr;
    }
    EnvironmentMap* env = scene->take(new EnvironmentMapEquiRect(scene->take(tex)));
    if (importanceN > 0)
        env->initializeImportanceSampling(importanceN);
    return wptHostFinish(scene, width, height, radians(vfovDegrees), vec3(eye), vec3(at), 0.0f, 1.0f);
}
