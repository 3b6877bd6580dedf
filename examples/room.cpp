/*
 * room.cpp -- a small application written against include/wurblpt the way the reference's example
 * applications are written against libwurblpt (compare wurblpt-cornellbox.cpp:230-280): build a scene,
 * update the BVH, render with mcpt(), tone-map, save.  It links libwurblpt_hip.so and nothing else;
 * everything from mcpt() on runs on the GPU.
 *
 *   g++ -std=c++20 -O2 -fopenmp -Iinclude examples/room.cpp -Lwurblpt_amd/lib -lwurblpt_hip -Wl,-rpath,$PWD/wurblpt_amd/lib -o room
 *   ./room [width height samplesSqrt outdir]
 *
 * Writes room.png (sRGB), room.pfm (the linear frame), room-depth.pfm (camera space depth from
 * getGroundTruth) and room-blur.png (the swinging panel over an exposure interval).
 */
#include <cstdio>
#include <cstdlib>
#include <string>

#include <wurblpt/wurblpt.hpp>

using namespace WurblPT;

int main(int argc, char* argv[])
{
    const unsigned int width = argc > 1 ? atoi(argv[1]) : 640;
    const unsigned int height = argc > 2 ? atoi(argv[2]) : 480;
    const unsigned int samplesSqrt = argc > 3 ? atoi(argv[3]) : 8;
    const std::string outdir = argc > 4 ? argv[4] : ".";

    Scene scene;
    Material* white = scene.take(new MaterialLambertian(vec3(0.73f)), "white");
    Material* red = scene.take(new MaterialLambertian(vec3(0.65f, 0.05f, 0.05f)), "red");
    Material* green = scene.take(new MaterialLambertian(vec3(0.12f, 0.45f, 0.15f)), "green");
    Texture* checker = scene.take(new TextureChecker(vec3(0.2f, 0.3f, 0.7f), vec3(0.9f), 8, 8));
    Material* tiles = scene.take(new MaterialLambertian(vec3(0.8f), checker), "tiles");
    Material* metal = scene.take(new MaterialGGX(vec3(0.95f, 0.85f, 0.6f), vec2(0.1f)), "metal");
    Material* glass = scene.take(new MaterialGlass(vec3(0.1f), 1.5f), "glass");
    Material* lamp = scene.take(new LightDiffuse(vec3(12.0f)), "lamp");

    /* a 2 x 2 x 2 room, open towards the camera; generateQuad() makes a quad in the xy plane, [-1,1]^2 */
    const quat toFloor = toQuat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f));
    const quat toCeiling = toQuat(radians(90.0f), vec3(1.0f, 0.0f, 0.0f));
    const quat toLeft = toQuat(radians(90.0f), vec3(0.0f, 1.0f, 0.0f));
    const quat toRight = toQuat(radians(-90.0f), vec3(0.0f, 1.0f, 0.0f));
    scene.take(new MeshInstance(scene.take(generateQuad()), tiles, Transformation(vec3(0.0f, 0.0f, 0.0f), toFloor)));
    scene.take(new MeshInstance(scene.take(generateQuad()), white, Transformation(vec3(0.0f, 2.0f, 0.0f), toCeiling)));
    scene.take(new MeshInstance(scene.take(generateQuad()), white, Transformation(vec3(0.0f, 1.0f, -1.0f))));
    scene.take(new MeshInstance(scene.take(generateQuad()), red, Transformation(vec3(-1.0f, 1.0f, 0.0f), toLeft)));
    scene.take(new MeshInstance(scene.take(generateQuad()), green, Transformation(vec3(1.0f, 1.0f, 0.0f), toRight)));
    scene.take(new MeshInstance(scene.take(generateQuad()), lamp, Transformation(vec3(0.0f, 1.99f, 0.0f), toCeiling, vec3(0.3f))), HotSpot);
    scene.take(new MeshInstance(scene.take(generateCube()), metal,
                Transformation(vec3(-0.4f, 0.5f, -0.3f), toQuat(radians(25.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.25f, 0.5f, 0.25f))));
    scene.take(new Sphere(vec3(0.45f, 0.3f, 0.2f), 0.3f, glass));
    /* a panel that swings about the y axis between t = 0 and t = 1 */
    const int swing = scene.take(new AnimationKeyframes(0.0f, Transformation(vec3(0.3f, 1.2f, -0.6f), toQuat(radians(-30.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.3f)),
                1.0f, Transformation(vec3(0.3f, 1.2f, -0.6f), toQuat(radians(40.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.3f))));
    scene.take(new MeshInstance(scene.take(generateQuad()), scene.take(new MaterialTwoSided(red, green)), swing));

    Optics optics(Projection(radians(45.0f), float(width) / height));
    Camera camera(optics, Transformation::fromLookAt(vec3(0.0f, 1.0f, 3.4f), vec3(0.0f, 1.0f, 0.0f)));
    std::string error;

    /* a still at t = 0.5 */
    scene.updateBVH(0.5f, 0.5f);
    SensorRGB sensor(width, height);
    mcpt(sensor, camera, scene, samplesSqrt, 0.5f, 0.5f);
    const Array<float>& hdr = sensor.result();
    if (!saveImage(hdr, outdir + "/room.pfm", &error) || !saveImage(toSRGB(uniformRationalQuantization(hdr, maxLuminance(hdr) / 100.0f, 8.0f)), outdir + "/room.png", &error)) {
        fprintf(stderr, "%s\n", error.c_str());
        return 1;
    }
    GroundTruth gt = getGroundTruth(sensor, camera, scene, 0.5f, GroundTruth::CameraSpaceDepths | GroundTruth::Materials);
    if (!saveImage(gt.cameraSpaceDepths, outdir + "/room-depth.pfm", &error)) {
        fprintf(stderr, "%s\n", error.c_str());
        return 1;
    }
    fprintf(stderr, "centre pixel: depth %.4f, material \"%s\"\n", gt.cameraSpaceDepths.at(width / 2, height / 2)[0],
            scene.materialNames()[gt.materials.at(width / 2, height / 2)[0]].c_str());
    /* the run's record in the frame's tags (wurblpt.hpp:425-435 for the CPU; here for the device) */
    for (const char* tag : { "WURBLPT/SAMPLES_PER_PIXEL", "WURBLPT/COMPILER", "WURBLPT/DEVICE_MODEL", "WURBLPT/DEVICE_COUNT", "WURBLPT/DEVICE_SECONDS", "WURBLPT/DEVICE_KERNEL" })
        fprintf(stderr, "%s = %s\n", tag, hdr.globalTagList().value(tag).c_str());

    /* the same view over the exposure interval [0, 1]: the panel blurs */
    scene.updateBVH(0.0f, 1.0f);
    SensorRGB blurred(width, height);
    mcpt(blurred, camera, scene, samplesSqrt, 0.0f, 1.0f);
    if (!saveImage(toSRGB(uniformRationalQuantization(blurred.result(), maxLuminance(blurred.result()) / 100.0f, 8.0f)), outdir + "/room-blur.png", &error)) {
        fprintf(stderr, "%s\n", error.c_str());
        return 1;
    }
    return 0;
}
