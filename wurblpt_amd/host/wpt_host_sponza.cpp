/*
 * wpt_host_sponza.cpp -- the Sponza-class procedural stand-in of BASELINE config 3.
 *
 * The real Crytek Sponza OBJ, its textures and the HDR environment map are not available
 * offline, so this builds a seeded scene of the same class with the generator
 * functions: a colonnaded courtyard (subdivided floor and walls, two rows of cylinder columns
 * with bases and capitals, beams, hanging curtains, vases) with what the OBJ importer would
 * produce (reference import.hpp:288-398): textured Lambertian and ModPhong materials, normal
 * maps on some of them, an alpha-textured two-sided curtain, plus a GGX and a mirror object.
 * Geometry is authored in "OBJ units" and baked through the transformation of
 * wurblpt-sponza.cpp:49-53 (rotate 90 degrees about Y, scale 0.01); camera of :145-148;
 * procedural sun + sky equirectangular float32 environment map with importance sampling.
 * `detail` scales the tessellation: 1.0 gives about 262 k triangles, tests use small values.
 */
#include <cmath>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "../../include/wurblpt/camera.hpp"
#include "../../include/wurblpt/generator.hpp"
#include "../../include/wurblpt/scene.hpp"
#include "../../include/wurblpt/sensor.hpp"

using namespace WurblPT;

struct wpt_host_scene;
wpt_host_scene* wptHostFinish(Scene* scene, unsigned int width, unsigned int height, float vfovRadians, const vec3& from,
        const vec3& at, float aperture, float focusDist);

namespace {

struct Rng {
    std::mt19937 g;
    explicit Rng(unsigned int seed) : g(seed) {}
    float u01() { return float(g() >> 8) * (1.0f / 16777216.0f); }
};

/* seeded value noise, bilinear over a lattice, a few octaves */
struct ValueNoise {
    int n;
    std::vector<float> lattice;
    ValueNoise(Rng& rng, int n_) : n(n_), lattice(size_t(n_) * n_)
    {
        for (float& v : lattice)
            v = rng.u01();
    }
    float at(float x, float y) const
    {
        x -= std::floor(x);
        y -= std::floor(y);
        float fx = x * n, fy = y * n;
        int x0 = int(fx) % n, y0 = int(fy) % n, x1 = (x0 + 1) % n, y1 = (y0 + 1) % n;
        float ax = fx - std::floor(fx), ay = fy - std::floor(fy);
        float a = lattice[y0 * n + x0] * (1.0f - ax) + lattice[y0 * n + x1] * ax;
        float b = lattice[y1 * n + x0] * (1.0f - ax) + lattice[y1 * n + x1] * ax;
        return a * (1.0f - ay) + b * ay;
    }
    float fbm(float x, float y) const { return 0.5f * at(x, y) + 0.3f * at(2.0f * x + 0.37f, 2.0f * y + 0.11f) + 0.2f * at(4.0f * x + 0.71f, 4.0f * y + 0.53f); }
};

unsigned char toByte(float v)
{
    v = v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v;
    return (unsigned char)(v * 255.0f + 0.5f);
}

/* sRGB colour texture: two base colours blended by noise, with a brick / tile pattern */
Texture* makeColorTexture(Scene& scene, Rng& rng, int size, const vec3& c0, const vec3& c1, int tilesX, int tilesY, bool withAlpha)
{
    ValueNoise noise(rng, 16);
    Array<uint8_t> img(size, size, withAlpha ? 4 : 3);
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++) {
            float u = (x + 0.5f) / size, v = (y + 0.5f) / size;
            float t = noise.fbm(u, v);
            float gu = u * tilesX - std::floor(u * tilesX), gv = v * tilesY - std::floor(v * tilesY);
            float mortar = (gu < 0.06f || gv < 0.08f) ? 0.55f : 1.0f;
            uint8_t* p = img.at(x, y);
            for (int k = 0; k < 3; k++)
                p[k] = toByte((c0[k] * (1.0f - t) + c1[k] * t) * mortar);
            if (withAlpha) {
                /* curtain with holes: a soft dot pattern in the alpha channel */
                float du = gu - 0.5f, dv = gv - 0.5f;
                p[3] = toByte((du * du + dv * dv < 0.06f) ? 0.15f + 0.5f * t : 1.0f);
            }
        }
    return scene.take(createTextureImage(img));
}

/* normal map (linear uint8): derivative of a noise height field */
Texture* makeNormalMap(Scene& scene, Rng& rng, int size, float strength)
{
    ValueNoise noise(rng, 24);
    Array<uint8_t> img(size, size, 3);
    float e = 1.0f / size;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++) {
            float u = (x + 0.5f) / size, v = (y + 0.5f) / size;
            float dx = (noise.fbm(u + e, v) - noise.fbm(u - e, v)) * strength;
            float dy = (noise.fbm(u, v + e) - noise.fbm(u, v - e)) * strength;
            vec3 n = normalize(vec3(-dx, -dy, 1.0f));
            uint8_t* p = img.at(x, y);
            for (int k = 0; k < 3; k++)
                p[k] = toByte(0.5f * n[k] + 0.5f);
        }
    return scene.take(createTextureImage(img, LinearizeSRGB_Off));
}

/* single-channel linear texture (shininess / opacity modulation) */
Texture* makeGreyTexture(Scene& scene, Rng& rng, int size, float lo, float hi)
{
    ValueNoise noise(rng, 12);
    Array<uint8_t> img(size, size, 1);
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            img.at(x, y)[0] = toByte(lo + (hi - lo) * noise.fbm((x + 0.5f) / size, (y + 0.5f) / size));
    return scene.take(createTextureImage(img, LinearizeSRGB_Off));
}

/* procedural sky: horizon-to-zenith gradient, ground, and a small bright sun; float32 RGB,
 * equirectangular, row 0 = v 0 = straight down */
Texture* makeSky(Scene& scene, int width, int height)
{
    Array<float> img(width, height, 3);
    const vec3 sunDir = normalize(vec3(0.35f, 0.8f, -0.45f));
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float lat = ((y + 0.5f) / height - 0.5f) * pi;
            float lon = ((x + 0.5f) / width) * 2.0f * pi;
            vec3 d(-std::cos(lat) * std::sin(lon), std::sin(lat), std::cos(lat) * std::cos(lon));
            float up = d.y();
            vec3 c = up > 0.0f ? mix(vec3(0.9f, 0.95f, 1.0f), vec3(0.25f, 0.45f, 0.9f), std::sqrt(up)) : vec3(0.2f, 0.18f, 0.15f);
            float s = dot(d, sunDir);
            if (s > 0.9995f)
                c = c + vec3(800.0f, 760.0f, 700.0f);
            else if (s > 0.99f)
                c = c + vec3(4.0f, 3.6f, 3.0f) * ((s - 0.99f) / 0.0095f);
            float* p = img.at(x, y);
            p[0] = c.x();
            p[1] = c.y();
            p[2] = c.z();
        }
    return scene.take(createTextureImage(img));
}

int scaled(float detail, int full, int least)
{
    int v = int(full * detail + 0.5f);
    return v < least ? least : v;
}

}

static wpt_host_scene* sponzaLike(unsigned int seed, float detail, unsigned int texSize, unsigned int envWidth,
        int importanceN, unsigned int width, unsigned int height, const char* rgl0, const char* rgl1)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    Rng rng(seed);
    /* wurblpt-sponza.cpp:49-53: OBJ space -> world */
    const Transformation objToWorld(vec3(0.0f), toQuat(radians(90.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.01f));
    const int ts = int(texSize);

    Texture* floorTex = makeColorTexture(scene, rng, ts, vec3(0.55f, 0.5f, 0.42f), vec3(0.8f, 0.76f, 0.68f), 8, 8, false);
    Texture* floorNrm = makeNormalMap(scene, rng, ts, 6.0f);
    Texture* wallTex = makeColorTexture(scene, rng, ts, vec3(0.62f, 0.45f, 0.33f), vec3(0.78f, 0.62f, 0.5f), 12, 24, false);
    Texture* wallNrm = makeNormalMap(scene, rng, ts, 10.0f);
    Texture* columnTex = makeColorTexture(scene, rng, ts, vec3(0.7f, 0.68f, 0.62f), vec3(0.86f, 0.84f, 0.8f), 1, 6, false);
    Texture* columnShi = makeGreyTexture(scene, rng, ts / 2 > 4 ? ts / 2 : 4, 0.2f, 1.0f);
    Texture* curtainTex[3] = {
        makeColorTexture(scene, rng, ts, vec3(0.7f, 0.08f, 0.08f), vec3(0.9f, 0.2f, 0.15f), 10, 10, true),
        makeColorTexture(scene, rng, ts, vec3(0.1f, 0.3f, 0.7f), vec3(0.2f, 0.5f, 0.9f), 10, 10, true),
        makeColorTexture(scene, rng, ts, vec3(0.1f, 0.5f, 0.15f), vec3(0.3f, 0.75f, 0.3f), 10, 10, true) };
    Texture* specTex = makeColorTexture(scene, rng, ts / 2 > 4 ? ts / 2 : 4, vec3(0.05f), vec3(0.35f), 4, 4, false);

    /* what the importer produces (import.hpp:328-386) */
    MaterialLambertian* floorMat = new MaterialLambertian(vec3(0.7f), floorTex);
    floorMat->normalTex = floorNrm;
    scene.take(floorMat, "floor");
    MaterialLambertian* wallMat = new MaterialLambertian(vec3(0.7f), wallTex);
    wallMat->normalTex = wallNrm;
    scene.take(wallMat, "bricks");
    MaterialModPhong* columnMat = new MaterialModPhong;
    columnMat->haveNIR = false;
    columnMat->diffuse = vec4(0.6f, 0.6f, 0.6f, 0.0f);
    columnMat->diffuseTex = columnTex;
    columnMat->specular = vec4(0.25f, 0.25f, 0.25f, 0.0f);
    columnMat->specularTex = specTex;
    columnMat->shininess = 60.0f;
    columnMat->shininessTex = columnShi;
    columnMat->opacity = 1.0f;
    scene.take(columnMat, "column");
    Material* curtainMat[3];
    for (int i = 0; i < 3; i++) {
        MaterialModPhong* m = new MaterialModPhong;
        m->haveNIR = false;
        m->diffuse = vec4(0.6f, 0.6f, 0.6f, 0.0f);
        m->diffuseTex = curtainTex[i];
        m->diffuseTexHasAlpha = true;
        m->specular = vec4(0.04f, 0.04f, 0.04f, 0.0f);
        m->shininess = 12.0f;
        m->opacity = 1.0f;
        m->transmissive = vec4(0.3f, 0.3f, 0.3f, 0.0f);
        scene.take(m, "fabric");
        curtainMat[i] = scene.take(new MaterialTwoSided(m, m), "fabric-two-sided");
    }
    Material* beamMat = scene.take(new MaterialModPhong(vec3(0.45f, 0.3f, 0.2f), vec3(0.1f), 30.0f), "wood");
    Material* vaseMat = scene.take(new MaterialGGX(vec3(0.95f, 0.75f, 0.4f), vec2(0.15f, 0.25f)), "brass");
    Material* columnUse = columnMat;
    Material* wallUse = wallMat;
    if (rgl0 && rgl1) {
        /* BASELINE config 5 flavour: measured BRDFs with normal maps on the large surfaces */
        MaterialRGL* r0 = new MaterialRGL(rgl0);
        r0->normalTex = wallNrm;
        wallUse = scene.take(r0, "measured-wall");
        MaterialRGL* r1 = new MaterialRGL(rgl1);
        r1->normalTex = floorNrm;
        columnUse = scene.take(r1, "measured-column");
        beamMat = scene.take(new MaterialRGL(rgl0), "measured-beam");
        vaseMat = scene.take(new MaterialRGL(rgl1), "measured-vase");
    }
    Material* mirrorMat = scene.take(new MaterialMirror(vec3(0.9f)), "mirror");

    auto T = [&](const vec3& t, const vec3& s, const quat& r = quat::null()) { return objToWorld * Transformation(t, r, s); };
    const quat layFlat = toQuat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f)); /* quad (XY) -> floor (XZ), facing up */
    const float halfL = 1400.0f, halfW = 600.0f, wallH = 1100.0f;

    /* floor: the largest share of the triangles, like Sponza's tiled floor */
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(0.0f), vec3(halfL, halfW, 1.0f), layFlat), scaled(detail, 270, 2))), floorMat));
    /* four walls */
    const int ws = scaled(detail, 64, 1);
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(0.0f, wallH * 0.5f, -halfW), vec3(halfL, wallH * 0.5f, 1.0f)), ws)), wallUse));
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(0.0f, wallH * 0.5f, halfW), vec3(halfL, wallH * 0.5f, 1.0f), toQuat(radians(180.0f), vec3(0.0f, 1.0f, 0.0f))), ws)), wallUse));
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(-halfL, wallH * 0.5f, 0.0f), vec3(halfW, wallH * 0.5f, 1.0f), toQuat(radians(90.0f), vec3(0.0f, 1.0f, 0.0f))), ws)), wallUse));
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(halfL, wallH * 0.5f, 0.0f), vec3(halfW, wallH * 0.5f, 1.0f), toQuat(radians(-90.0f), vec3(0.0f, 1.0f, 0.0f))), ws)), wallUse));
    /* a partial roof: two side strips, the middle is open to the sky */
    const quat faceDown = toQuat(radians(90.0f), vec3(1.0f, 0.0f, 0.0f));
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(0.0f, wallH, -halfW * 0.7f), vec3(halfL, halfW * 0.3f, 1.0f), faceDown), scaled(detail, 24, 1))), wallUse));
    scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(0.0f, wallH, halfW * 0.7f), vec3(halfL, halfW * 0.3f, 1.0f), faceDown), scaled(detail, 24, 1))), wallUse));
    /* two rows of columns with base and capital, beams on top */
    const int columns = 12;
    const int cylSlices = scaled(detail, 96, 6);
    for (int row = 0; row < 2; row++) {
        float z = (row == 0 ? -1.0f : 1.0f) * halfW * 0.45f;
        for (int i = 0; i < columns; i++) {
            float x = -halfL * 0.85f + i * (2.0f * halfL * 0.85f / (columns - 1));
            scene.take(new MeshInstance(scene.take(generateClosedCylinder(T(vec3(x, 330.0f, z), vec3(42.0f, 300.0f, 42.0f)), cylSlices)), columnUse));
            scene.take(new MeshInstance(scene.take(generateCube(T(vec3(x, 15.0f, z), vec3(60.0f, 15.0f, 60.0f)), scaled(detail, 4, 1))), columnUse));
            scene.take(new MeshInstance(scene.take(generateCube(T(vec3(x, 645.0f, z), vec3(58.0f, 15.0f, 58.0f)), scaled(detail, 4, 1))), columnUse));
        }
        scene.take(new MeshInstance(scene.take(generateCube(T(vec3(0.0f, 690.0f, z), vec3(halfL * 0.9f, 30.0f, 40.0f)), scaled(detail, 8, 1))), beamMat));
    }
    /* curtains between columns */
    for (int i = 0; i < 9; i++) {
        float x = -halfL * 0.7f + i * (2.0f * halfL * 0.7f / 8.0f);
        float z = (i % 2 == 0 ? -1.0f : 1.0f) * halfW * 0.45f;
        scene.take(new MeshInstance(scene.take(generateQuad(T(vec3(x, 430.0f, z), vec3(95.0f, 190.0f, 1.0f), toQuat(radians(6.0f * (rng.u01() - 0.5f)), vec3(0.0f, 1.0f, 0.0f))),
                            scaled(detail, 20, 1))), curtainMat[i % 3]));
    }
    /* vases and a mirror sphere on the floor */
    const int sphSlices = scaled(detail, 96, 8), sphStacks = scaled(detail, 48, 4);
    for (int i = 0; i < 6; i++) {
        float x = -halfL * 0.6f + i * (2.0f * halfL * 0.6f / 5.0f) + 40.0f * (rng.u01() - 0.5f);
        float z = (i % 2 == 0 ? -170.0f : 170.0f) + 60.0f * (rng.u01() - 0.5f);
        float r = 35.0f + 25.0f * rng.u01();
        scene.take(new MeshInstance(scene.take(generateSphere(T(vec3(x, r * 1.4f, z), vec3(r, r * 1.4f, r)), sphSlices, sphStacks)), i == 2 ? mirrorMat : vaseMat));
    }

    /* environment map with importance sampling */
    Texture* sky = makeSky(scene, int(envWidth), int(envWidth / 2));
    EnvironmentMap* env = scene.take(new EnvironmentMapEquiRect(sky));
    if (importanceN > 0)
        env->initializeImportanceSampling(importanceN);

    /* camera of wurblpt-sponza.cpp:145-148 */
    return wptHostFinish(scenePtr, width, height, radians(70.0f), vec3(0.0f, 1.7f, 0.0f), vec3(0.0f, 1.7f, -1.0f), 0.0f, 1.0f);
}

extern "C" wpt_host_scene* wpt_host_sponza_like(unsigned int seed, float detail, unsigned int texSize, unsigned int envWidth,
        int importanceN, unsigned int width, unsigned int height)
{
    return sponzaLike(seed, detail, texSize, envWidth, importanceN, width, height, nullptr, nullptr);
}

/* BASELINE config 5 stand-in ("Bistro-class": measured BRDFs + normal maps, environment importance
 * sampling): the Sponza-class architecture with MaterialRGL on walls, columns, beams and vases */
extern "C" wpt_host_scene* wpt_host_measured_like(unsigned int seed, float detail, unsigned int texSize, unsigned int envWidth,
        int importanceN, const char* rgl0, const char* rgl1, unsigned int width, unsigned int height)
{
    return sponzaLike(seed, detail, texSize, envWidth, importanceN, width, height, rgl0, rgl1);
}

/* BASELINE config 4 stand-in ("San-Miguel-class", wurblpt-san-miguel.cpp:36-44,58-70): a
 * courtyard whose triangle count is dominated by foliage -- clouds of small two-sided leaf quads
 * around tree crowns and hedges -- plus ground, facades, arcade columns and furniture.  As in the
 * reference application every material is two-sided (ImportBitTwoSidedMaterials), there are no
 * light sources, and the environment is a constant texture without importance sampling, so all
 * light comes from rays that escape.  `triangles` is the approximate total. */
extern "C" wpt_host_scene* wpt_host_courtyard_like(unsigned int seed, unsigned int triangles, unsigned int texSize,
        unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    Rng rng(seed);
    const int ts = int(texSize);
    const float detail = std::sqrt(float(triangles) / 1.0e7f); /* tessellation of the architecture */

    Texture* groundTex = makeColorTexture(scene, rng, ts, vec3(0.5f, 0.45f, 0.38f), vec3(0.72f, 0.68f, 0.6f), 16, 16, false);
    Texture* groundNrm = makeNormalMap(scene, rng, ts, 5.0f);
    Texture* facadeTex = makeColorTexture(scene, rng, ts, vec3(0.75f, 0.6f, 0.4f), vec3(0.9f, 0.8f, 0.62f), 10, 6, false);
    Texture* leafTex = makeColorTexture(scene, rng, ts / 2 > 4 ? ts / 2 : 4, vec3(0.08f, 0.3f, 0.05f), vec3(0.3f, 0.6f, 0.15f), 2, 2, true);
    Texture* barkTex = makeColorTexture(scene, rng, ts / 2 > 4 ? ts / 2 : 4, vec3(0.25f, 0.17f, 0.1f), vec3(0.4f, 0.3f, 0.2f), 1, 8, false);

    auto twoSided = [&](Material* m) { return scene.take(new MaterialTwoSided(m, m), "two-sided"); };
    MaterialLambertian* groundBase = new MaterialLambertian(vec3(0.7f), groundTex);
    groundBase->normalTex = groundNrm;
    Material* groundMat = twoSided(scene.take(groundBase, "ground"));
    MaterialModPhong* facadeBase = new MaterialModPhong;
    facadeBase->haveNIR = false;
    facadeBase->diffuse = vec4(0.7f, 0.7f, 0.7f, 0.0f);
    facadeBase->diffuseTex = facadeTex;
    facadeBase->specular = vec4(0.05f, 0.05f, 0.05f, 0.0f);
    facadeBase->shininess = 20.0f;
    facadeBase->opacity = 1.0f;
    Material* facadeMat = twoSided(scene.take(facadeBase, "facade"));
    MaterialModPhong* leafBase = new MaterialModPhong;
    leafBase->haveNIR = false;
    leafBase->diffuse = vec4(0.6f, 0.6f, 0.6f, 0.0f);
    leafBase->diffuseTex = leafTex;
    leafBase->diffuseTexHasAlpha = true;
    leafBase->specular = vec4(0.06f, 0.06f, 0.06f, 0.0f);
    leafBase->shininess = 25.0f;
    leafBase->opacity = 1.0f;
    Material* leafMat = twoSided(scene.take(leafBase, "leaf"));
    Material* barkMat = twoSided(scene.take(new MaterialLambertian(vec3(0.6f), barkTex), "bark"));
    Material* woodMat = twoSided(scene.take(new MaterialModPhong(vec3(0.5f, 0.35f, 0.22f), vec3(0.12f), 40.0f), "wood"));

    size_t soFar = 0; /* triangles of the architecture */
    auto addInstance = [&](Mesh* mesh, const Material* material) {
        soFar += mesh->triangleCount();
        scene.take(new MeshInstance(mesh, material));
    };
    const quat layFlat = toQuat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f));
    const float halfX = 14.0f, halfZ = 10.0f, wallH = 9.0f;
    auto T = [&](const vec3& t, const vec3& s, const quat& r = quat::null()) { return Transformation(t, r, s); };
    addInstance(scene.take(generateQuad(T(vec3(0.0f), vec3(halfX, halfZ, 1.0f), layFlat), scaled(detail, 300, 2))), groundMat);
    const int ws = scaled(detail, 120, 1);
    addInstance(scene.take(generateQuad(T(vec3(0.0f, wallH * 0.5f, -halfZ), vec3(halfX, wallH * 0.5f, 1.0f)), ws)), facadeMat);
    addInstance(scene.take(generateQuad(T(vec3(0.0f, wallH * 0.5f, halfZ), vec3(halfX, wallH * 0.5f, 1.0f), toQuat(radians(180.0f), vec3(0.0f, 1.0f, 0.0f))), ws)), facadeMat);
    addInstance(scene.take(generateQuad(T(vec3(-halfX, wallH * 0.5f, 0.0f), vec3(halfZ, wallH * 0.5f, 1.0f), toQuat(radians(90.0f), vec3(0.0f, 1.0f, 0.0f))), ws)), facadeMat);
    addInstance(scene.take(generateQuad(T(vec3(halfX, wallH * 0.5f, 0.0f), vec3(halfZ, wallH * 0.5f, 1.0f), toQuat(radians(-90.0f), vec3(0.0f, 1.0f, 0.0f))), ws)), facadeMat);
    /* arcade columns along the long sides, tables in the middle */
    const int cylSlices = scaled(detail, 160, 6);
    for (int side = 0; side < 2; side++)
        for (int i = 0; i < 10; i++) {
            float x = -halfX * 0.9f + i * (2.0f * halfX * 0.9f / 9.0f);
            float z = (side == 0 ? -1.0f : 1.0f) * halfZ * 0.8f;
            addInstance(scene.take(generateClosedCylinder(T(vec3(x, 1.8f, z), vec3(0.22f, 1.8f, 0.22f)), cylSlices)), facadeMat);
        }
    for (int i = 0; i < 8; i++) {
        float x = -halfX * 0.5f + (i % 4) * (halfX / 3.0f) + 0.4f * (rng.u01() - 0.5f);
        float z = (i < 4 ? -1.5f : 1.5f) + 0.4f * (rng.u01() - 0.5f);
        addInstance(scene.take(generateCube(T(vec3(x, 0.72f, z), vec3(0.6f, 0.03f, 0.6f)), scaled(detail, 6, 1))), woodMat);
        addInstance(scene.take(generateClosedCylinder(T(vec3(x, 0.36f, z), vec3(0.05f, 0.36f, 0.05f)), scaled(detail, 24, 6))), woodMat);
    }
    /* trees: trunk + a crown of leaf quads; hedges along the short sides: the same leaf soup in a box */
    const size_t leafQuads = triangles / 2 > soFar / 2 + 64 ? triangles / 2 - soFar / 2 : 64;
    const int trees = 12, hedges = 4;
    const size_t perCloud = leafQuads / size_t(trees + hedges);
    auto leafCloud = [&](const vec3& centre, const vec3& radii, bool box, size_t count) {
        std::vector<vec3> pos, nrm;
        std::vector<vec2> tc;
        std::vector<unsigned int> ind;
        pos.reserve(count * 4); nrm.reserve(count * 4); tc.reserve(count * 4); ind.reserve(count * 6);
        for (size_t q = 0; q < count; q++) {
            vec3 p;
            if (box) {
                p = vec3(2.0f * rng.u01() - 1.0f, 2.0f * rng.u01() - 1.0f, 2.0f * rng.u01() - 1.0f);
            } else {
                do {
                    p = vec3(2.0f * rng.u01() - 1.0f, 2.0f * rng.u01() - 1.0f, 2.0f * rng.u01() - 1.0f);
                } while (dot(p, p) > 1.0f);
            }
            p = centre + p * radii;
            vec3 n = normalize(vec3(rng.u01() - 0.5f, rng.u01() - 0.2f, rng.u01() - 0.5f) + vec3(0.0f, 1e-3f, 0.0f));
            vec3 a = normalize(cross(n, std::fabs(n.y()) < 0.9f ? vec3(0.0f, 1.0f, 0.0f) : vec3(1.0f, 0.0f, 0.0f)));
            vec3 b = cross(n, a);
            const float size = 0.025f + 0.035f * rng.u01();
            const unsigned int base = (unsigned int)pos.size();
            pos.push_back(p - size * a - size * b); pos.push_back(p + size * a - size * b);
            pos.push_back(p - size * a + size * b); pos.push_back(p + size * a + size * b);
            for (int k = 0; k < 4; k++)
                nrm.push_back(n);
            tc.push_back(vec2(0.0f, 0.0f)); tc.push_back(vec2(1.0f, 0.0f)); tc.push_back(vec2(0.0f, 1.0f)); tc.push_back(vec2(1.0f, 1.0f));
            const unsigned int t[6] = { base, base + 1, base + 2, base + 1, base + 3, base + 2 };
            ind.insert(ind.end(), t, t + 6);
        }
        addInstance(scene.take(new Mesh(pos, nrm, tc, ind)), leafMat);
    };
    for (int i = 0; i < trees; i++) {
        float x = -halfX * 0.75f + (i % 6) * (2.0f * halfX * 0.75f / 5.0f) + 0.8f * (rng.u01() - 0.5f);
        float z = (i < 6 ? -1.0f : 1.0f) * (halfZ * 0.42f + 0.8f * rng.u01());
        float trunkH = 2.2f + 1.2f * rng.u01();
        addInstance(scene.take(generateClosedCylinder(T(vec3(x, trunkH * 0.5f, z), vec3(0.16f, trunkH * 0.5f, 0.16f)), scaled(detail, 48, 6))), barkMat);
        leafCloud(vec3(x, trunkH + 1.3f, z), vec3(1.7f + 0.5f * rng.u01(), 1.4f, 1.7f + 0.5f * rng.u01()), false, perCloud);
    }
    for (int i = 0; i < hedges; i++) {
        float x = (i < 2 ? -1.0f : 1.0f) * halfX * 0.93f;
        float z = (i % 2 == 0 ? -1.0f : 1.0f) * halfZ * 0.35f;
        leafCloud(vec3(x, 0.6f, z), vec3(0.4f, 0.6f, halfZ * 0.3f), true, perCloud);
    }

    /* wurblpt-san-miguel.cpp:41-43: a constant environment, no importance sampling */
    Texture* envTex = scene.take(new TextureConstant(vec4(1.0f)));
    scene.take(new EnvironmentMapEquiRect(envTex));
    /* a standing viewer in a corner of the courtyard looking across it (45 degrees, :58) */
    return wptHostFinish(scenePtr, width, height, radians(45.0f), vec3(-halfX * 0.8f, 1.6f, halfZ * 0.6f), vec3(0.0f, 1.8f, -halfZ * 0.3f), 0.0f, 1.0f);
}

/* wurblpt-furnace-test.cpp:36-86 with the tessellated sphere the file keeps as its alternative
 * (the analytic Sphere hitable is a "next" row): a unit sphere of one material inside a constant
 * environment of radiance 1, camera at (0,0,5), 40 degrees.  With a cosine-sampled Lambertian
 * of albedo a every sample of a pixel on the sphere is a * 1 exactly (up to rounding), which
 * pins scatter / attenuation / pdf and the environment term without any reference build.
 * material: 0 Lambertian 0.42, 1 Lambertian 1, 2 ModPhong(1,0), 3 ModPhong(0,1), 4 ModPhong(.5,.5),
 * 5 GGX albedo 1 roughness 0.5 */
extern "C" wpt_host_scene* wpt_host_furnace(int material, int slices, unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    Texture* tex = scene.take(new TextureConstant(vec4(1.0f)));
    scene.take(new EnvironmentMapEquiRect(tex));
    Material* mat;
    switch (material) {
    case 0: mat = scene.take(new MaterialLambertian(vec3(0.42f))); break;
    case 1: mat = scene.take(new MaterialLambertian(vec3(1.0f))); break;
    case 2: mat = scene.take(new MaterialModPhong(vec3(1.0f), vec3(0.0f))); break;
    case 3: mat = scene.take(new MaterialModPhong(vec3(0.0f), vec3(1.0f))); break;
    case 4: mat = scene.take(new MaterialModPhong(vec3(0.5f), vec3(0.5f))); break;
    case 5: mat = scene.take(new MaterialGGX(vec3(1.0f), vec2(0.5f, 0.5f))); break;
    /* on an analytic sphere (no facets, so no total internal reflection at facet edges): clear glass and a perfect mirror
     * neither absorb nor emit, every path leaves with attenuation exactly 1 */
    case 6: mat = scene.take(new MaterialGlass(vec4(0.0f), 1.5f)); break;
    case 7: mat = scene.take(new MaterialMirror(vec3(1.0f))); break;
    default: mat = nullptr; break;
    }
    if (material == 6 || material == 7) {
        scene.take(new Sphere(vec3(0.0f), 1.0f, mat));
        return wptHostFinish(scenePtr, width, height, radians(40.0f), vec3(0.0f, 0.0f, 5.0f), vec3(0.0f, 0.0f, 0.0f), 0.0f, 1.0f);
    }
    scene.take(new MeshInstance(scene.take(generateSphere(Transformation(), slices, slices / 2)), mat));
    return wptHostFinish(scenePtr, width, height, radians(40.0f), vec3(0.0f, 0.0f, 5.0f), vec3(0.0f, 0.0f, 0.0f), 0.0f, 1.0f);
}

wpt_host_scene* wptHostFinishAnimated(Scene* scene, unsigned int width, unsigned int height, float vfovRadians,
        const Animation* cameraAnimation, float t0, float t1, float aperture, float focusDist);

/* A room with moving things for the exposure interval [t0, t1] (animation.hpp, animation_keyframes.hpp, the way
 * wurblpt-animations.cpp sets such scenes up): floor, back wall and a side wall; a GGX cube that turns and drifts
 * (three key frames, scaled), a textured two-sided panel that swings about the z axis, a quad light that slides and
 * tilts (a hot spot, so its pdfValue / direction move too), a static glass block; the camera dollies sideways and
 * pans.  variant bit 0: thin lens as well; bit 1: the camera stands still (only the instances move);
 * bit 2: nothing but the camera moves; bit 3: rolling marbles as well (wurblpt-rolling-marbles.cpp: unit spheres
 * placed, scaled and turned by their animation alone -- a checkered one, a glass one, and a glowing one that is a
 * hot spot). */
extern "C" wpt_host_scene* wpt_host_animated(int variant, float t0, float t1, unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    const bool moveInstances = !(variant & 4);
    Material* white = scene.take(new MaterialLambertian(vec3(0.73f, 0.72f, 0.69f)));
    Material* blue = scene.take(new MaterialLambertian(vec3(0.15f, 0.25f, 0.6f)));
    Material* metal = scene.take(new MaterialGGX(vec3(0.95f, 0.8f, 0.5f), vec2(0.15f)));
    Material* glass = scene.take(new MaterialGlass(vec3(0.1f), 1.5f));
    Texture* checker = scene.take(new TextureChecker(vec3(0.85f, 0.3f, 0.2f), vec3(0.9f, 0.9f, 0.85f), 6, 6));
    Material* panel = scene.take(new MaterialTwoSided(scene.take(new MaterialLambertian(vec3(0.7f), checker)), blue));
    Material* light = scene.take(new LightDiffuse(vec3(9.0f)));

    AnimationKeyframes* cubeMotion = new AnimationKeyframes;
    cubeMotion->addKeyframe(0.0f, Transformation(vec3(-0.6f, 0.35f, -0.2f), toQuat(radians(10.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.35f)));
    cubeMotion->addKeyframe(0.5f, Transformation(vec3(-0.45f, 0.4f, -0.1f), toQuat(radians(55.0f), vec3(0.2f, 1.0f, 0.0f)), vec3(0.38f)));
    cubeMotion->addKeyframe(1.0f, Transformation(vec3(-0.2f, 0.5f, 0.1f), toQuat(radians(130.0f), vec3(0.3f, 1.0f, 0.1f)), vec3(0.3f, 0.4f, 0.3f)));
    AnimationKeyframes* panelMotion = new AnimationKeyframes(0.0f, Transformation(vec3(0.7f, 0.8f, -0.6f), toQuat(radians(-25.0f), vec3(0.0f, 0.0f, 1.0f)), vec3(0.4f)),
            1.0f, Transformation(vec3(0.7f, 0.8f, -0.6f), toQuat(radians(35.0f), vec3(0.0f, 0.0f, 1.0f)), vec3(0.4f)));
    AnimationKeyframes* lightMotion = new AnimationKeyframes(0.0f, Transformation(vec3(-0.3f, 1.9f, -0.2f), toQuat(radians(90.0f), vec3(1.0f, 0.0f, 0.0f)), vec3(0.3f)),
            1.0f, Transformation(vec3(0.3f, 1.85f, 0.1f), toQuat(radians(70.0f), vec3(1.0f, 0.1f, 0.0f)), vec3(0.3f)));
    const int cubeAnim = scene.take(cubeMotion);
    const int panelAnim = scene.take(panelMotion);
    const int lightAnim = scene.take(lightMotion);

    /* static room */
    scene.take(new MeshInstance(scene.take(generateQuad()), white,
                Transformation(vec3(0.0f, 0.0f, 0.0f), toQuat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f)), vec3(2.0f))));
    scene.take(new MeshInstance(scene.take(generateQuad()), white, Transformation(vec3(0.0f, 1.0f, -1.5f), quat::null(), vec3(2.0f, 1.0f, 1.0f))));
    scene.take(new MeshInstance(scene.take(generateQuad()), blue,
                Transformation(vec3(-1.5f, 1.0f, 0.0f), toQuat(radians(90.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(2.0f, 1.0f, 1.0f))));
    scene.take(new MeshInstance(scene.take(generateCube()), glass, Transformation(vec3(0.5f, 0.3f, 0.4f), toQuat(radians(20.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.25f, 0.3f, 0.25f))));
    /* moving things: an instance transformation and an animation on top of it, an animation alone, a moving light */
    if (moveInstances) {
        scene.take(new MeshInstance(scene.take(generateCube(Transformation(), 2)), metal, Transformation(vec3(0.0f), toQuat(radians(15.0f), vec3(0.0f, 0.0f, 1.0f)), vec3(1.0f, 0.8f, 1.0f)), cubeAnim));
        scene.take(new MeshInstance(scene.take(generateQuad(Transformation(), 3)), panel, panelAnim));
        scene.take(new MeshInstance(scene.take(generateQuad()), light, lightAnim), HotSpot);
    } else {
        scene.take(new MeshInstance(scene.take(generateCube(Transformation(), 2)), metal, cubeMotion->at(0.3f)));
        scene.take(new MeshInstance(scene.take(generateQuad(Transformation(), 3)), panel, panelMotion->at(0.3f)));
        scene.take(new MeshInstance(scene.take(generateQuad()), light, lightMotion->at(0.3f)), HotSpot);
    }
    if (variant & 8) {
        Texture* marbleTex = scene.take(new TextureChecker(vec3(0.1f, 0.5f, 0.2f), vec3(0.9f), 8, 4));
        Material* marble = scene.take(new MaterialLambertian(vec3(0.8f), marbleTex));
        Material* glow = scene.take(new LightDiffuse(vec3(6.0f, 5.0f, 3.0f)));
        AnimationKeyframes* roll = new AnimationKeyframes;
        for (int k = 0; k <= 4; k++) /* rolls along +x: the rotation matches the distance travelled */
            roll->addKeyframe(0.25f * k, Transformation(vec3(-0.9f + 0.3f * k, 0.15f, 0.9f), toQuat(radians(-115.0f * k), vec3(0.0f, 0.0f, 1.0f)), vec3(0.15f)));
        AnimationKeyframes* bounce = new AnimationKeyframes(0.0f, Transformation(vec3(0.2f, 0.2f, 1.0f), quat::null(), vec3(0.2f)),
                1.0f, Transformation(vec3(0.35f, 0.6f, 0.8f), toQuat(radians(40.0f), vec3(1.0f, 0.0f, 0.0f)), vec3(0.16f, 0.22f, 0.16f)));
        AnimationKeyframes* drift = new AnimationKeyframes(0.0f, Transformation(vec3(-0.8f, 1.2f, 0.3f), quat::null(), vec3(0.08f)),
                1.0f, Transformation(vec3(-0.5f, 1.0f, 0.6f), quat::null(), vec3(0.1f)));
        scene.take(new Sphere(marble, scene.take(roll)));
        scene.take(new Sphere(glass, scene.take(bounce)));
        scene.take(new Sphere(glow, scene.take(drift)), HotSpot);
    }
    AnimationKeyframes* cameraMotion = new AnimationKeyframes;
    const vec3 up(0.0f, 1.0f, 0.0f);
    cameraMotion->addKeyframe(0.0f, Transformation::fromLookAt(vec3(0.1f, 0.9f, 3.0f), vec3(0.0f, 0.7f, 0.0f), up));
    if (!(variant & 2)) {
        cameraMotion->addKeyframe(0.6f, Transformation::fromLookAt(vec3(0.25f, 0.95f, 2.9f), vec3(0.05f, 0.7f, 0.0f), up));
        cameraMotion->addKeyframe(1.0f, Transformation::fromLookAt(vec3(0.45f, 1.0f, 2.7f), vec3(0.1f, 0.75f, 0.0f), up));
    }
    return wptHostFinishAnimated(scenePtr, width, height, radians(45.0f), cameraMotion, t0, t1, (variant & 1) ? 0.05f : 0.0f, 3.0f);
}

/* Scenes with analytic spheres (hitable_sphere.hpp), the "next" row f2 of the scope table:
 * variant 0  ground + Lambertian (checker texture, rotated frame) / GGX / glass / mirror spheres,
 *            lit by a spherical light AND a quad light, both hot spots (mixed sphere and triangle
 *            next-event estimation, as in wurblpt-mis-test)
 * variant 1  the same objects without lights under a cube environment map (envmap.hpp:250-285)
 * variant 2  wurblpt-furnace-test.cpp as written: analytic sphere, Lambertian 0.42, constant
 *            equirect environment -- every pixel on the sphere is exactly 0.42
 * variant 3  camera inside a large emitting sphere that is a hot spot (pdfValue's inside branch)
 * variant 4  variant 1 with importance sampling of the cube map (next-event estimation towards it) */
extern "C" wpt_host_scene* wpt_host_spheres(int variant, unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    if (variant == 2) {
        Texture* tex = scene.take(new TextureConstant(vec4(1.0f)));
        scene.take(new EnvironmentMapEquiRect(tex));
        scene.take(new Sphere(vec3(0.0f), 1.0f, scene.take(new MaterialLambertian(vec3(0.42f)))));
        return wptHostFinish(scenePtr, width, height, radians(40.0f), vec3(0.0f, 0.0f, 5.0f), vec3(0.0f, 0.0f, 0.0f), 0.0f, 1.0f);
    }
    Texture* checker = scene.take(new TextureChecker(vec3(0.8f, 0.2f, 0.2f), vec3(0.9f, 0.9f, 0.8f), 8, 4));
    Texture* groundTex = scene.take(new TextureChecker(vec3(0.3f), vec3(0.7f), 10, 10));
    Material* ground = scene.take(new MaterialLambertian(vec3(0.7f), groundTex));
    Material* textured = scene.take(new MaterialLambertian(vec3(0.7f), checker));
    Material* ggx = scene.take(new MaterialGGX(vec3(0.9f, 0.7f, 0.3f), vec2(0.2f, 0.2f)));
    Material* glass = scene.take(new MaterialGlass(vec4(0.1f), vec4(1.5f), vec4(1.0f)));
    Material* mirror = scene.take(new MaterialMirror(vec3(0.9f)));
    const quat layFlat = toQuat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f));
    scene.take(new MeshInstance(scene.take(generateQuad(Transformation(vec3(0.0f), layFlat, vec3(6.0f, 6.0f, 1.0f)), 4)), ground));
    scene.take(new Sphere(textured, Transformation(vec3(-2.2f, 0.8f, 0.0f), toQuat(radians(35.0f), normalize(vec3(0.3f, 1.0f, 0.2f))), vec3(0.8f))));
    scene.take(new Sphere(vec3(-0.5f, 0.6f, 0.8f), 0.6f, ggx));
    scene.take(new Sphere(vec3(1.0f, 0.7f, 0.2f), 0.7f, glass));
    scene.take(new Sphere(mirror, Transformation(vec3(2.6f, 0.5f, -0.6f), quat::null(), vec3(0.5f, 0.25f, 0.4f)))); /* radius = max(scaling) */
    if (variant == 0) {
        Material* light = scene.take(new LightDiffuse(vec3(12.0f, 11.0f, 9.0f)));
        scene.take(new Sphere(vec3(0.0f, 4.0f, 0.5f), 0.4f, light), HotSpot);
        Material* light2 = scene.take(new LightDiffuse(vec3(2.0f, 3.0f, 5.0f)));
        scene.take(new MeshInstance(scene.take(generateQuad(Transformation(vec3(-4.0f, 2.0f, -2.0f), toQuat(radians(60.0f), vec3(0.0f, 1.0f, 0.0f)), vec3(0.8f, 0.8f, 1.0f)), 1)), light2), HotSpot);
    } else if (variant == 1 || variant == 4) {
        Texture* side[6] = {
            scene.take(new TextureConstant(vec4(0.9f, 0.3f, 0.2f, 0.5f))), scene.take(new TextureConstant(vec4(0.2f, 0.8f, 0.3f, 0.4f))),
            scene.take(new TextureChecker(vec3(1.5f, 1.5f, 2.0f), vec3(0.4f, 0.5f, 0.9f), 6, 6)), scene.take(new TextureConstant(vec4(0.15f, 0.12f, 0.1f, 0.1f))),
            scene.take(new TextureChecker(vec3(0.9f, 0.9f, 0.2f), vec3(0.2f, 0.2f, 0.9f), 3, 5)), scene.take(new TextureConstant(vec4(0.6f, 0.6f, 0.6f, 0.6f))) };
        EnvironmentMap* env = scene.take(new EnvironmentMapCube(side[0], side[1], side[2], side[3], side[4], side[5]));
        if (variant == 4) /* importance sampled: the importance maps are independent of the parameterisation (envmap.hpp:40-42) */
            env->initializeImportanceSampling(24);
    } else {
        Material* glow = scene.take(new LightDiffuse(vec3(0.8f, 0.9f, 1.0f)));
        Material* glowTwoSided = scene.take(new MaterialTwoSided(glow, glow));
        scene.take(new Sphere(vec3(0.0f, 1.0f, 0.0f), 9.0f, glowTwoSided), HotSpot);
    }
    return wptHostFinish(scenePtr, width, height, radians(45.0f), vec3(0.0f, 2.0f, 6.0f), vec3(0.0f, 0.7f, 0.0f), 0.0f, 1.0f);
}

/* The scene of the reference's statistical test for multiple importance sampling (wurblpt-mis-test.cpp:32-97, after
 * Veach's four plates under four lights): a white room, four GGX plates of growing roughness tilted towards the
 * camera, four sphere lights of growing size and equal radiance.  withHotSpots = 0 leaves the lights to material
 * sampling alone; the two renderings must converge to the same image (:118-133).  lightMask: bit i = light i is there
 * (15 = the reference's scene). */
extern "C" wpt_host_scene* wpt_host_mis_test(int withHotSpots, unsigned int lightMask, unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    Material* white = scene.take(new MaterialLambertian(vec4(0.8f)));
    /* walls: unit quads scaled by five; where each stands, and how it is turned to face the room */
    struct Wall { vec3 where; float degrees; vec3 axis; };
    const vec3 yAxis(0.0f, 1.0f, 0.0f), xAxis(1.0f, 0.0f, 0.0f);
    const Wall walls[6] = {
        { vec3(-2.6f, 0.0f, 0.0f), +90.0f, yAxis }, { vec3(+2.6f, 0.0f, 0.0f), -90.0f, yAxis },
        { vec3(0.0f, 0.0f, +5.0f), 180.0f, yAxis }, { vec3(0.0f, 0.0f, -4.6f), 0.0f, yAxis },
        { vec3(0.0f, -2.499, 0.0f), +90.0f, xAxis }, { vec3(0.0f, -5.0f, 0.0f), -90.0f, xAxis } };
    Mesh* wallMesh[6];
    for (int i = 0; i < 6; i++) {
        Transformation T;
        T.translate(walls[i].where);
        T.scale(vec3(5.0f));
        T.rotate(toQuat(radians(walls[i].degrees), walls[i].axis));
        wallMesh[i] = scene.take(generateQuad(T));
    }
    for (int i = 0; i < 6; i++)
        scene.take(new MeshInstance(wallMesh[i], white));
    /* plates: roughness, position, tilt about x */
    const float roughness[4] = { 0.001f, 0.008, 0.03f, 0.1f };
    const vec3 platePosition[4] = { vec3(0.0f, -4.2f, -4.2f), vec3(0.0f, -4.6f, -3.8f), vec3(0.0f, -4.8f, -3.4f), vec3(0.0f, -4.9f, -3.0f) };
    const float plateTilt[4] = { -35.0f, -47.0f, -59.0f, -71.0f };
    Material* plateMaterial[4];
    for (int i = 0; i < 4; i++)
        plateMaterial[i] = scene.take(new MaterialGGX(vec3(1.0f), vec2(roughness[i])));
    for (int i = 0; i < 4; i++) {
        const Transformation T(platePosition[i], toQuat(radians(plateTilt[i]), xAxis), vec3(2.0f, 0.3f, 1.0f));
        scene.take(new MeshInstance(scene.take(generateQuad(T)), plateMaterial[i]));
    }
    /* lights: x position and radius */
    const float lightX[4] = { -1.5f, -0.5f, +0.5f, +1.5f };
    const float lightRadius[4] = { 0.032f, 0.08f, 0.2f, 0.5f };
    Material* lightMaterial[4];
    for (int i = 0; i < 4; i++)
        lightMaterial[i] = scene.take(new LightDiffuse(vec3(4.0f)));
    for (int i = 0; i < 4; i++)
        if (lightMask & (1u << i))
            scene.take(new Sphere(lightMaterial[i], Transformation(vec3(lightX[i], -3.5f, -4.0f), quat::null(), vec3(lightRadius[i]))),
                    withHotSpots ? HotSpot : ColdSpot);
    /* camera at (0, -4.5, -1.2) looking down -z (Transformation without rotation), 50 degrees */
    return wptHostFinish(scenePtr, width, height, radians(50.0f), vec3(0.0f, -4.5f, -1.2f), vec3(0.0f, -4.5f, -2.2f), 0.0f, 1.0f);
}

/* A probe for what the camera sees directly: a quad light with an emission texture (an 8-bit sRGB image of 7 x 5 texels
 * with coordinate factor / offset and value factor / offset) in the middle of the view, and around it an equirectangular
 * environment of float texels (16 x 8).  With one sample through every pixel centre a pixel is LightDiffuse::emitted or
 * EnvironmentMapEquiRect::L of its ray and nothing else, so TextureImage::value (texel decoding, the half-texel shift, the
 * clamp at the far edges, fract of the transformed coordinates) and L's direction-to-coordinate mapping can be checked
 * against an evaluation written independently from the reference's text (tests/test_scene_and_integrator.py).
 * compat: 0 Mitsuba, 1 surround video. */
extern "C" wpt_host_scene* wpt_host_texture_probe(int compat, unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    unsigned int state = 12345u;
    auto next = [&state]() { state = state * 1664525u + 1013904223u; return state >> 8; };
    Array<float> sky(16, 8, 3);
    for (size_t i = 0; i < sky.elementCount(); i++)
        for (int c = 0; c < 3; c++)
            sky[i][c] = float(next() & 0xffffu) * (1.0f / 65536.0f) * 2.0f;
    Array<uint8_t> picture(7, 5, 3);
    for (size_t i = 0; i < picture.elementCount(); i++)
        for (int c = 0; c < 3; c++)
            picture[i][c] = uint8_t(next() & 0xffu);
    Texture* skyTex = scene.take(createTextureImage(sky, LinearizeSRGB_Off));
    scene.take(new EnvironmentMapEquiRect(skyTex, compat == 0 ? EnvironmentMapEquiRect::CompatibilityMitsuba : EnvironmentMapEquiRect::CompatibilitySurroundVideo));
    Texture* pictureTex = scene.take(createTextureImage(picture, LinearizeSRGB_Auto, vec2(2.0f, 3.0f), vec2(0.25f, 0.1f),
                vec4(0.9f, 0.8f, 0.7f, 1.0f), vec4(0.05f, 0.02f, 0.01f, 0.0f)));
    Material* light = scene.take(new LightDiffuse(vec3(1.5f, 1.2f, 0.9f), pictureTex));
    scene.take(new MeshInstance(scene.take(generateQuad(Transformation(vec3(0.1f, -0.05f, -2.0f), quat::null(), vec3(0.6f, 0.4f, 1.0f)))), light));
    return wptHostFinish(scenePtr, width, height, radians(60.0f), vec3(0.0f, 0.0f, 0.0f), vec3(0.3f, 0.2f, -1.0f), 0.0f, 1.0f);
}

/* Builds the tables of a measured BRDF file as MaterialRGL does (include/wurblpt/rgl.hpp).
 * Returns the number of floats of the table pool (0 on error, message on stderr); copies at most
 * `capacity` of them. */
extern "C" unsigned long long wpt_host_rgl_build(const char* filename, wpt_rgl_brdf* brdf, float* pool, unsigned long long capacity)
{
    std::vector<float> p;
    std::string error;
    wpt_rgl_brdf b;
    if (!buildRglBrdf(filename, p, b, error)) {
        fprintf(stderr, "wpt_host: %s\n", error.c_str());
        return 0;
    }
    if (brdf)
        *brdf = b;
    if (pool)
        memcpy(pool, p.data(), sizeof(float) * (p.size() < capacity ? p.size() : capacity));
    return p.size();
}

/* Measured-BRDF scenes (material_rgl.hpp): variant 0 = the furnace test with its RGL option
 * (wurblpt-furnace-test.cpp:67) on a tessellated sphere in a constant environment; variant 1 =
 * a ground plane with a quad light (hot spot, so next-event estimation evaluates the BRDF with
 * scatterToDirection), one sphere of the first file, one analytic sphere of the second file with
 * a normal map, under a constant environment of low radiance */
extern "C" wpt_host_scene* wpt_host_rgl_scene(int variant, const char* file0, const char* file1, unsigned int width, unsigned int height)
{
    Scene* scenePtr = new Scene;
    Scene& scene = *scenePtr;
    MaterialRGL* m0 = new MaterialRGL(file0);
    scene.take(m0, "rgl0");
    if (variant == 0) {
        Texture* tex = scene.take(new TextureConstant(vec4(1.0f)));
        scene.take(new EnvironmentMapEquiRect(tex));
        scene.take(new MeshInstance(scene.take(generateSphere(Transformation(), 48, 24)), m0));
        return wptHostFinish(scenePtr, width, height, radians(40.0f), vec3(0.0f, 0.0f, 5.0f), vec3(0.0f, 0.0f, 0.0f), 0.0f, 1.0f);
    }
    MaterialRGL* m1 = new MaterialRGL(file1);
    Rng rng(5);
    m1->normalTex = makeNormalMap(scene, rng, 32, 4.0f);
    scene.take(m1, "rgl1");
    Texture* envTex = scene.take(new TextureConstant(vec4(0.05f, 0.06f, 0.08f, 0.06f)));
    scene.take(new EnvironmentMapEquiRect(envTex));
    Texture* groundTex = scene.take(new TextureChecker(vec3(0.3f), vec3(0.7f), 10, 10));
    Material* ground = scene.take(new MaterialLambertian(vec3(0.7f), groundTex));
    const quat layFlat = toQuat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f));
    scene.take(new MeshInstance(scene.take(generateQuad(Transformation(vec3(0.0f), layFlat, vec3(6.0f, 6.0f, 1.0f)), 4)), ground));
    scene.take(new MeshInstance(scene.take(generateSphere(Transformation(vec3(-1.1f, 0.8f, 0.0f), quat::null(), vec3(0.8f)), 32, 16)), m0));
    scene.take(new Sphere(vec3(1.1f, 0.8f, 0.2f), 0.8f, m1));
    Material* light = scene.take(new LightDiffuse(vec3(14.0f, 13.0f, 12.0f)));
    scene.take(new MeshInstance(scene.take(generateQuad(Transformation(vec3(0.0f, 4.0f, 1.0f), toQuat(radians(90.0f), vec3(1.0f, 0.0f, 0.0f)), vec3(0.7f, 0.7f, 1.0f)), 1)), light), HotSpot);
    return wptHostFinish(scenePtr, width, height, radians(45.0f), vec3(0.0f, 2.0f, 6.0f), vec3(0.0f, 0.7f, 0.0f), 0.0f, 1.0f);
}
