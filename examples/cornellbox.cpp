/*
 * cornellbox.cpp -- the Cornell box rendered through the C++ API of include/wurblpt/ on the GPU.
 *
 * Same scene description style as the reference's wurblpt-cornellbox application (Scene::take,
 * Mesh, MeshInstance, materials, Camera, SensorRGB, mcpt); writes the linear frame as PFM.
 * Build:  g++ -std=c++20 -fopenmp -I include examples/cornellbox.cpp -L wurblpt_amd/lib -lwurblpt_hip -o cornellbox
 */
#include <cstdio>

#include <wurblpt/wurblpt.hpp>

using namespace WurblPT;

static void quad(Scene& scene, const Material* m, const vec3& a, const vec3& b, const vec3& c, const vec3& d, const vec3& n,
        HotSpotType hot = ColdSpot)
{
    scene.take(new MeshInstance(scene.take(new Mesh({ a, b, c, d }, { n, n, n, n },
                        { vec2(0.0f, 0.0f), vec2(1.0f, 0.0f), vec2(1.0f, 1.0f), vec2(0.0f, 1.0f) }, { 0, 1, 2, 0, 2, 3 })), m), hot);
}

int main(int argc, char* argv[])
{
    unsigned int size = argc > 1 ? atoi(argv[1]) : 512;
    unsigned int samplesSqrt = argc > 2 ? atoi(argv[2]) : 8;

    Scene scene;
    Material* white = scene.take(new MaterialLambertian(vec3(0.725f, 0.71f, 0.68f)));
    Material* red = scene.take(new MaterialLambertian(vec3(0.63f, 0.065f, 0.05f)));
    Material* green = scene.take(new MaterialLambertian(vec3(0.14f, 0.45f, 0.091f)));
    Material* light = scene.take(new LightDiffuse(vec3(4.0f)));
    Material* metal = scene.take(new MaterialGGX(vec3(1.0f), vec2(0.04f)));
    quad(scene, red, vec3(-1, 0, 1), vec3(-1, 0, -1), vec3(-1, 2, -1), vec3(-1, 2, 1), vec3(1, 0, 0));
    quad(scene, green, vec3(1, 0, -1), vec3(1, 0, 1), vec3(1, 2, 1), vec3(1, 2, -1), vec3(-1, 0, 0));
    quad(scene, white, vec3(-1, 0, 1), vec3(1, 0, 1), vec3(1, 0, -1), vec3(-1, 0, -1), vec3(0, 1, 0));
    quad(scene, white, vec3(-1, 2, 1), vec3(-1, 2, -1), vec3(1, 2, -1), vec3(1, 2, 1), vec3(0, -1, 0));
    quad(scene, white, vec3(-1, 0, -1), vec3(1, 0, -1), vec3(1, 2, -1), vec3(-1, 2, -1), vec3(0, 0, 1));
    /* a metal cube and a textured checker sphere built with the generators */
    scene.take(new MeshInstance(scene.take(generateCube(Transformation(vec3(-0.4f, 0.4f, -0.3f), toQuat(radians(20.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.3f, 0.4f, 0.3f)))), metal));
    Texture* checker = scene.take(new TextureChecker(vec3(0.8f), vec3(0.1f, 0.1f, 0.6f), 12, 6));
    Material* checkered = scene.take(new MaterialLambertian(vec3(0.5f), checker));
    scene.take(new MeshInstance(scene.take(generateSphere(Transformation(vec3(0.45f, 0.3f, 0.3f), quat::null(), vec3(0.3f)))), checkered));
    quad(scene, light, vec3(-0.24f, 1.98f, 0.16f), vec3(-0.24f, 1.98f, -0.22f), vec3(0.23f, 1.98f, -0.22f), vec3(0.23f, 1.98f, 0.16f),
            vec3(0, -1, 0), HotSpot);

    SensorRGB sensor(size, size);
    Optics optics(Projection(radians(50.0f), sensor.aspectRatio()));
    Camera camera(optics, Transformation::fromLookAt(vec3(0.0f, 1.0f, 3.2f), vec3(0.0f, 1.0f, -1.0f), vec3(0.0f, 1.0f, 0.0f)));
    Parameters params;
    scene.updateBVH();
    mcpt(sensor, camera, scene, samplesSqrt, 0.0f, 0.0f, params);

    const Array<float>& img = sensor.result();
    FILE* f = fopen("cornellbox.pfm", "wb");
    if (f) {
        fprintf(f, "PF\n%u %u\n-1.0\n", size, size); /* PFM stores rows bottom-up, like the sensor */
        fwrite(img.data(), sizeof(float), size_t(size) * size * 3, f);
        fclose(f);
    }
    double sum = 0.0;
    for (size_t i = 0; i < size_t(size) * size * 3; i++)
        sum += static_cast<const float*>(img.data())[i];
    printf("rendered %ux%u with %u spp on kernel %s, mean radiance %.6f\n", size, size, samplesSqrt * samplesSqrt,
            img.globalTagList().value("WURBLPT/DEVICE_KERNEL").c_str(), sum / (double(size) * size * 3));
    return 0;
}
