/*
 * wurblpt.hpp -- the public entry point: Parameters and mcpt() with the reference's
 * signatures (wurblpt.hpp:79-96,279-286,439-449).
 *
 * mcpt() does what the reference's does -- renders samplesSqrt^2 samples per pixel into the
 * sensor's frame -- by flattening the scene and handing pixel blocks to the MI355X kernels
 * behind the C ABI of include/wurblpt_hip.h.  There is no CPU fallback: without a device, or
 * with scene content the kernels do not know, it prints the reason and aborts (the reference's
 * own fatal-error behaviour, bvh.hpp:260-263).
 */
#pragma once

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../wurblpt_hip.h"
#include "animation.hpp"
#include "camera.hpp"
#include "constants.hpp"
#include "envmap.hpp"
#include "generator.hpp"
#include "gvm.hpp"
#include "imageio.hpp"
#include "import.hpp"
#include "material.hpp"
#include "mesh.hpp"
#include "mpi.hpp"
#include "postproc.hpp"
#include "prng.hpp"
#include "scene.hpp"
#include "sensor.hpp"
#include "texture.hpp"
#include "transformation.hpp"

namespace WurblPT {

class Parameters
{
public:
    unsigned int maxPathComponents;
    float rrThreshold;
    bool randomizeRayOverPixel;
    float minHitDistance;

    Parameters() : maxPathComponents(128), rrThreshold(1.0f), randomizeRayOverPixel(true), minHitDistance(0.00001f) {}
};

inline wpt_params makeParams(const Parameters& params, const SensorRGB& sensor)
{
    wpt_params p;
    p.max_path_components = params.maxPathComponents;
    p.rr_threshold = params.rrThreshold;
    p.randomize_ray_over_pixel = params.randomizeRayOverPixel ? 1 : 0;
    p.min_hit_distance = params.minHitDistance;
    p.min_dist_to_light = sensor.minDistToLight;
    p.max_dist_to_light = sensor.maxDistToLight;
    p.min_path_len = sensor.minPathLen;
    p.max_path_len = sensor.maxPathLen;
    p.t0 = 0.0f;
    p.t1 = 0.0f;
    return p;
}

[[noreturn]] inline void mcptFatal(const std::string& what)
{
    fprintf(stderr, "mcpt: %s\n", what.c_str());
    abort();
}

inline void mcpt(MPICoordinator& mpiCoordinator, Sensor& sensor, const Camera& camera, const Scene& scene,
        unsigned int samplesSqrt, float t0 = 0.0f, float t1 = 0.0f, const Parameters& params = Parameters())
{
    if (scene.bvhNeedsUpdate(t0, t1))
        mcptFatal("Scene::updateBVH(t0, t1) must run before mcpt()");
    SensorRGB* rgb = dynamic_cast<SensorRGB*>(&sensor);
    if (!rgb)
        mcptFatal("only SensorRGB runs on the device path");
    wpt_camera cam;
    if (!camera.describe(cam, t0))
        mcptFatal("this camera cannot be described to the device path");
    FlatScene flat;
    std::string error;
    if (!scene.flatten(flat, &error))
        mcptFatal(error);
    if (camera.animation) {
        /* the camera's key frames join the scene's pool; rays take them at their own time when t0 != t1 */
        cam.animation = flat.addAnimation(camera.animation.get());
        if (cam.animation < 0)
            mcptFatal("only key frame animations (AnimationKeyframes) can go to the device");
    }
    const wpt_scene_desc desc = flat.desc();
    wpt_params p = makeParams(params, *rgb);
    p.t0 = t0;
    p.t1 = t1;
    const unsigned int width = sensor.width();
    const unsigned int height = sensor.height();
    ArrayContainer* pixelArray = sensor.pixelArray();

    fprintf(stderr, "Number of hitables that are hot spots: %zu\n", scene.hotSpots().size());
    fprintf(stderr, "Rendering %ux%u pixels with %u samples.\n", width, height, samplesSqrt * samplesSqrt);
    if (wpt_device_count() <= 0)
        mcptFatal(std::string("no HIP device: ") + wpt_last_error());

    mpiCoordinator.init(width, height, static_cast<float*>(pixelArray->data()), pixelArray->componentCount());
    std::vector<std::string> workerErrors(mpiCoordinator.devices().size());
    const auto renderStart = std::chrono::steady_clock::now();
    auto worker = [&](size_t w) {
        int device = mpiCoordinator.devices()[w];
        if (device >= 0 && wpt_select_device(device) != WPT_OK) {
            workerErrors[w] = wpt_last_error();
            return;
        }
        wpt_scene* dscene = nullptr;
        if (wpt_scene_upload(&desc, &dscene) != WPT_OK) {
            workerErrors[w] = wpt_last_error();
            return;
        }
        if (mpiCoordinator.devices().size() > 1) {
            /* several devices: a lane owns a pixel for all its samples, so a device is only as busy as it has pixels in
             * flight.  Each device takes its share in ONE launch -- the bands of 16 rows whose index is w modulo the number
             * of devices -- and writes them into the shared frame (wurblpt_hip.h: wpt_render_bands). */
            const unsigned int count = mpiCoordinator.devices().size();
            fprintf(stderr, "%s: device %d renders bands %zu, %zu, ... of 16 rows\n", mpiCoordinator.processId(), device, w, w + count);
            if (wpt_render_bands(dscene, &cam, &p, width, height, samplesSqrt, 16, w, count, mpiCoordinator.blockData(0)) != WPT_OK)
                workerErrors[w] = wpt_last_error();
            wpt_scene_free(dscene);
            return;
        }
        for (;;) {
            unsigned int blockStart, blockSize;
            mpiCoordinator.getBlock(&blockStart, &blockSize);
            if (blockSize == 0)
                break;
            fprintf(stderr, "%s: device %d renders block of size %u starting at %u\n", mpiCoordinator.processId(), device, blockSize, blockStart);
            if (wpt_render_block(dscene, &cam, &p, width, height, samplesSqrt, blockStart, blockSize,
                        mpiCoordinator.blockData(blockStart)) != WPT_OK) {
                workerErrors[w] = wpt_last_error();
                break;
            }
            mpiCoordinator.submitBlock(blockStart, blockSize);
        }
        wpt_scene_free(dscene);
    };
    if (mpiCoordinator.devices().size() == 1) {
        worker(0);
    } else {
        std::vector<std::thread> threads;
        for (size_t w = 0; w < mpiCoordinator.devices().size(); w++)
            threads.emplace_back(worker, w);
        for (auto& t : threads)
            t.join();
    }
    mpiCoordinator.finish();
    for (const std::string& e : workerErrors)
        if (!e.empty())
            mcptFatal(e);

    pixelArray->globalTagList().set("WURBLPT/SAMPLES_PER_PIXEL", std::to_string(samplesSqrt * samplesSqrt));
    pixelArray->globalTagList().set("WURBLPT/MAX_PATH_COMPONENTS", std::to_string(params.maxPathComponents));
    pixelArray->globalTagList().set("WURBLPT/RUSSIAN_ROULETTE_THRESHOLD", std::to_string(params.rrThreshold));
    pixelArray->globalTagList().set("WURBLPT/DEVICE_KERNEL", wpt_kernel_name());
    /* the run's record (wurblpt.hpp:393-400,425-435 notes compiler, CPU model, threads and CPU seconds): here the
     * compiler of the kernels, the device(s) and the wall-clock seconds from scene upload to the finished frame */
    std::string deviceModel;
    for (int device : mpiCoordinator.devices()) {
        int current = device;
        if (current < 0 && wpt_current_device(&current) != WPT_OK)
            current = 0;
        deviceModel += (deviceModel.empty() ? "" : "; ") + std::string(wpt_device_name(current));
    }
    pixelArray->globalTagList().set("WURBLPT/COMPILER", wpt_build_info());
    pixelArray->globalTagList().set("WURBLPT/DEVICE_MODEL", deviceModel);
    pixelArray->globalTagList().set("WURBLPT/DEVICE_COUNT", std::to_string(mpiCoordinator.devices().size()));
    pixelArray->globalTagList().set("WURBLPT/DEVICE_SECONDS",
            std::to_string(std::chrono::duration<double>(std::chrono::steady_clock::now() - renderStart).count()));
}

inline void mcpt(Sensor& sensor, const Camera& camera, const Scene& scene, unsigned int samplesSqrt, float t0 = 0.0f,
        float t1 = 0.0f, const Parameters& params = Parameters())
{
    MPICoordinator mpiCoordinator;
    mcpt(mpiCoordinator, sensor, camera, scene, samplesSqrt, t0, t1, params);
}

/* Ground Truth generation (reference: wurblpt.hpp:453-769).  One ray through the centre of every pixel
 * on the device (wpt_ground_truth, wurblpt_hip.h); the arrays a caller did not ask for stay empty. */

class GroundTruth
{
public:
    static constexpr unsigned int WorldSpacePositions         = (1 << 0);
    static constexpr unsigned int WorldSpaceGeometryNormals   = (1 << 1);
    static constexpr unsigned int WorldSpaceGeometryTangents  = (1 << 2);
    static constexpr unsigned int WorldSpaceMaterialNormals   = (1 << 3);
    static constexpr unsigned int WorldSpaceMaterialTangents  = (1 << 4);
    static constexpr unsigned int CameraSpacePositions        = (1 << 5);
    static constexpr unsigned int CameraSpaceGeometryNormals  = (1 << 6);
    static constexpr unsigned int CameraSpaceGeometryTangents = (1 << 7);
    static constexpr unsigned int CameraSpaceMaterialNormals  = (1 << 8);
    static constexpr unsigned int CameraSpaceMaterialTangents = (1 << 9);
    static constexpr unsigned int CameraSpaceDepths           = (1 << 10);
    static constexpr unsigned int CameraSpaceDistances        = (1 << 11);
    static constexpr unsigned int TexCoords                   = (1 << 12);
    static constexpr unsigned int WorldSpaceOffsetToPrev      = (1 << 13);
    static constexpr unsigned int WorldSpaceOffsetToNext      = (1 << 14);
    static constexpr unsigned int CameraSpaceOffsetToPrev     = (1 << 15);
    static constexpr unsigned int CameraSpaceOffsetToNext     = (1 << 16);
    static constexpr unsigned int PixelSpaceOffsetToPrev      = (1 << 17);
    static constexpr unsigned int PixelSpaceOffsetToNext      = (1 << 18);
    static constexpr unsigned int Materials                   = (1 << 19);
    static constexpr unsigned int All                         = (1 << 20) - 1;

    unsigned int bits;
    Array<float> worldSpacePositions;
    Array<float> worldSpaceGeometryNormals;
    Array<float> worldSpaceGeometryTangents;
    Array<float> worldSpaceMaterialNormals;
    Array<float> worldSpaceMaterialTangents;
    Array<float> cameraSpacePositions;
    Array<float> cameraSpaceGeometryNormals;
    Array<float> cameraSpaceGeometryTangents;
    Array<float> cameraSpaceMaterialNormals;
    Array<float> cameraSpaceMaterialTangents;
    Array<float> cameraSpaceDepths;    // == -cameraSpacePosition.z
    Array<float> cameraSpaceDistances; // == length(cameraSpacePosition)
    Array<float> texCoords;
    Array<float> worldSpaceOffsetToPrev;
    Array<float> worldSpaceOffsetToNext;
    Array<float> cameraSpaceOffsetToPrev;
    Array<float> cameraSpaceOffsetToNext;
    Array<float> pixelSpaceOffsetToPrev;
    Array<float> pixelSpaceOffsetToNext;
    Array<int32_t> materials;          // Scene::materialIndex() of the hitable's material, -1 where nothing is hit

    GroundTruth() : bits(0) {}

    GroundTruth(unsigned int width, unsigned int height, unsigned int bits = All) : bits(bits)
    {
        Array<float>* f[19] = { &worldSpacePositions, &worldSpaceGeometryNormals, &worldSpaceGeometryTangents,
            &worldSpaceMaterialNormals, &worldSpaceMaterialTangents, &cameraSpacePositions, &cameraSpaceGeometryNormals,
            &cameraSpaceGeometryTangents, &cameraSpaceMaterialNormals, &cameraSpaceMaterialTangents, &cameraSpaceDepths,
            &cameraSpaceDistances, &texCoords, &worldSpaceOffsetToPrev, &worldSpaceOffsetToNext, &cameraSpaceOffsetToPrev,
            &cameraSpaceOffsetToNext, &pixelSpaceOffsetToPrev, &pixelSpaceOffsetToNext };
        for (int k = 0; k < 19; k++)
            if (bits & (1u << k))
                *f[k] = Array<float>(width, height, wpt_gt_components[k]);
        if (bits & Materials)
            materials = Array<int32_t>(width, height, 1);
    }

    /* the arrays in the order of the bits, as wpt_ground_truth takes them (NULL where not asked for) */
    void arrayPointers(void* out[WPT_GT_ARRAY_COUNT])
    {
        ArrayContainer* a[WPT_GT_ARRAY_COUNT] = { &worldSpacePositions, &worldSpaceGeometryNormals, &worldSpaceGeometryTangents,
            &worldSpaceMaterialNormals, &worldSpaceMaterialTangents, &cameraSpacePositions, &cameraSpaceGeometryNormals,
            &cameraSpaceGeometryTangents, &cameraSpaceMaterialNormals, &cameraSpaceMaterialTangents, &cameraSpaceDepths,
            &cameraSpaceDistances, &texCoords, &worldSpaceOffsetToPrev, &worldSpaceOffsetToNext, &cameraSpaceOffsetToPrev,
            &cameraSpaceOffsetToNext, &pixelSpaceOffsetToPrev, &pixelSpaceOffsetToNext, &materials };
        for (int k = 0; k < WPT_GT_ARRAY_COUNT; k++)
            out[k] = (bits & (1u << k)) ? a[k]->data() : nullptr;
    }
};

/* The picture is taken at t0; the flow arrays compare with tPrev and tNext: the camera at those times (its animation,
 * or cameraPrev / cameraNext where an application keeps separate camera objects per frame) and the places the hit
 * points of animated instances have then (wurblpt.hpp:626-761). */
inline GroundTruth getGroundTruth(const Sensor& sensor, const Camera& camera, const Camera* cameraPrev, const Camera* cameraNext,
        const Scene& scene, float t0, float tPrev, float tNext, unsigned int groundTruthBits = GroundTruth::All,
        const Parameters& params = Parameters())
{
    if (scene.bvhNeedsUpdate(t0, t0))
        mcptFatal("Scene::updateBVH() must run before getGroundTruth()");
    wpt_camera cam, camPrev, camNext;
    if (!camera.describe(cam, t0) || !(cameraPrev ? cameraPrev->describe(camPrev, tPrev) : camera.describe(camPrev, tPrev))
            || !(cameraNext ? cameraNext->describe(camNext, tNext) : camera.describe(camNext, tNext)))
        mcptFatal("this camera cannot be described to the device path");
    FlatScene flat;
    std::string error;
    if (!scene.flatten(flat, &error))
        mcptFatal(error);
    const wpt_scene_desc desc = flat.desc();
    wpt_params p;
    p.max_path_components = params.maxPathComponents;
    p.rr_threshold = params.rrThreshold;
    p.randomize_ray_over_pixel = 0;
    p.min_hit_distance = params.minHitDistance;
    p.min_dist_to_light = 0.0f;
    p.max_dist_to_light = std::numeric_limits<float>::max();
    p.min_path_len = 0.0f;
    p.max_path_len = std::numeric_limits<float>::max();
    p.t0 = t0;
    p.t1 = t0;
    const unsigned int width = sensor.width();
    const unsigned int height = sensor.height();
    GroundTruth gt(width, height, groundTruthBits);
    fprintf(stderr, "Getting ground truth for %ux%u pixels at %.3fs... ", width, height, t0);
    if (wpt_device_count() <= 0)
        mcptFatal(std::string("no HIP device: ") + wpt_last_error());
    wpt_scene* dscene = nullptr;
    if (wpt_scene_upload(&desc, &dscene) != WPT_OK)
        mcptFatal(wpt_last_error());
    void* arrays[WPT_GT_ARRAY_COUNT];
    gt.arrayPointers(arrays);
    const float times[3] = { t0, tPrev, tNext };
    const wpt_status st = wpt_ground_truth(dscene, &cam, &camPrev, &camNext, times, &p, width, height, arrays);
    wpt_scene_free(dscene);
    if (st != WPT_OK)
        mcptFatal(wpt_last_error());
    /* the kernel reports positions in the flattened material list; the reference reports Scene::materialIndex() */
    if (groundTruthBits & GroundTruth::Materials) {
        for (size_t i = 0; i < gt.materials.elementCount(); i++) {
            int32_t& m = gt.materials[i][0];
            m = (m >= 0 && size_t(m) < flat.materialSceneIndex.size()) ? flat.materialSceneIndex[m] : -1;
        }
    }
    fprintf(stderr, "done\n");
    return gt;
}

inline GroundTruth getGroundTruth(const Sensor& sensor, const Camera& camera, const Scene& scene, float t0, float tPrev,
        float tNext, unsigned int groundTruthBits = GroundTruth::All, const Parameters& params = Parameters())
{
    return getGroundTruth(sensor, camera, nullptr, nullptr, scene, t0, tPrev, tNext, groundTruthBits, params);
}

inline GroundTruth getGroundTruth(const Sensor& sensor, const Camera& camera, const Scene& scene, float t0 = 0.0f,
        unsigned int groundTruthBits = GroundTruth::All, const Parameters& params = Parameters())
{
    return getGroundTruth(sensor, camera, scene, t0, t0, t0, groundTruthBits, params);
}

}
