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
#include <string>
#include <thread>
#include <vector>

#include "../wurblpt_hip.h"
#include "camera.hpp"
#include "constants.hpp"
#include "envmap.hpp"
#include "generator.hpp"
#include "gvm.hpp"
#include "material.hpp"
#include "mesh.hpp"
#include "mpi.hpp"
#include "postproc.hpp"
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
        mcptFatal("Scene::updateBVH() must run before mcpt()");
    if (t0 != t1)
        mcptFatal("motion blur (t0 != t1) is outside the device path");
    SensorRGB* rgb = dynamic_cast<SensorRGB*>(&sensor);
    if (!rgb)
        mcptFatal("only SensorRGB runs on the device path");
    wpt_camera cam;
    if (!camera.describe(cam))
        mcptFatal("this camera cannot be described to the device path");
    FlatScene flat;
    std::string error;
    if (!scene.flatten(flat, &error))
        mcptFatal(error);
    const wpt_scene_desc desc = flat.desc();
    const wpt_params p = makeParams(params, *rgb);
    const unsigned int width = sensor.width();
    const unsigned int height = sensor.height();
    ArrayContainer* pixelArray = sensor.pixelArray();

    fprintf(stderr, "Number of hitables that are hot spots: %zu\n", scene.hotSpots().size());
    fprintf(stderr, "Rendering %ux%u pixels with %u samples.\n", width, height, samplesSqrt * samplesSqrt);
    if (wpt_device_count() <= 0)
        mcptFatal(std::string("no HIP device: ") + wpt_last_error());

    mpiCoordinator.init(width, height, static_cast<float*>(pixelArray->data()), pixelArray->componentCount());
    std::vector<std::string> workerErrors(mpiCoordinator.devices().size());
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
}

inline void mcpt(Sensor& sensor, const Camera& camera, const Scene& scene, unsigned int samplesSqrt, float t0 = 0.0f,
        float t1 = 0.0f, const Parameters& params = Parameters())
{
    MPICoordinator mpiCoordinator;
    mcpt(mpiCoordinator, sensor, camera, scene, samplesSqrt, t0, t1, params);
}

}
