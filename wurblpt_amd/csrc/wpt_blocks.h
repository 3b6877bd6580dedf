/*
 * wpt_blocks.h -- the per-path logic of the integrator as three device functions that a
 * kernel runs for the lanes whose path is at that point:
 *
 *   blockNew     start the pixel's next sample: jitter, Camera::getRay (wurblpt.hpp:348-360)
 *   blockShade   one path component after its ray came back: emission, scatter, next-event
 *                set-up with MIS (wurblpt.hpp:131-252)
 *   blockNeeEnd  the next-event ray came back: add its contribution (wurblpt.hpp:203-218,236-250)
 *
 * each followed by `advancePath` (wurblpt.hpp:254-273) where the path continues.  They return
 * what the lane needs next: a ray to be traced (ps.ray / ps.rayKind are set), a new sample, or
 * nothing more (all samples done).  The traversal of the ray is the kernel's business.
 */
#ifndef WPT_BLOCKS_H
#define WPT_BLOCKS_H

#include "wpt_device.h"
#include "wpt_lens.h"

namespace wptk {

using namespace wptd;

constexpr uint32_t NO_HIT = 0xffffffffu;

enum { RAY_PATH = 0, RAY_NEE_LIGHT = 1, RAY_NEE_ENV = 2 };
enum { NEXT_TRACE = 0, NEXT_NEW = 1, NEXT_DONE = 2 };

struct LaneCounters {
    uint32_t rays, nodes, leaves, pdfs, scatters;
    /* COUNT builds: shader clock this lane spent in the sections of blockShade (profiling only):
     * [0] hit record + material, [1] scatter, [2] emission, [3] light pdf of the scattered direction,
     * [4] light sample, [5] light pdf of the light direction, [6] evaluation towards the light,
     * [7] environment sampling / continuation */
    unsigned long long shadeClock[8];
};

/* everything a pixel's path carries between blocks (registers) */
struct PathState {
    Prng prng;
    float acc0, acc1, acc2;
    uint32_t px, py;
    uint32_t sampleIndex, pathComponent;
    int rayKind;
    Ray ray;
    f4 att;
    f3 opl; /* opticalPathLength; SensorRGB reads channels 0..2 only */
    /* continuation of the path while a next-event ray is in flight */
    f4 nextAtt;
    f4 neeFactor; /* attenuation * directSR.attenuation / directPdf * weight (wurblpt.hpp:211,243) */
    f3 srDir;
    uint32_t chosenPrim;
    float time; /* FEAT_ANIM: the path's time (Ray::time; the thread's AnimationCache is set to it, wurblpt.hpp:361) */
    /* FEAT_ANIM: the lane's AnimationCache, one entry deep -- the matrix of the animation it used last at ps.time.
     * Consecutive leaf tests of a walk mostly hit triangles of one instance, and the lights share few animations. */
    int animCached;
    float animM[16];
};

/* AnimationCache::getM(ai) at the path's time.  Spheres do not go through it: measured on the test scene, sharing the
 * entry with them lets sphere and triangle leaves evict each other (162 -> 141 Msamples/s) and an entry of their own
 * is evicted by the next sphere (158), so their transformation is evaluated where it is needed. */
WPT_D wptanim::Trs animationTrs(const SceneView& sv, PathState& ps, int ai)
{
    return animationAt(sv, ai, ps.time);
}
WPT_D const float* animationMatrix(const SceneView& sv, PathState& ps, int ai)
{
    if (ps.animCached != ai) {
        wptanim::toMat4(animationAt(sv, ai, ps.time), ps.animM);
        ps.animCached = ai;
    }
    return ps.animM;
}
/* an animated sphere at the path's time, through the lane's cache: as hit() / direction() place it */
template<uint32_t F> WPT_D wpt_sphere sphereNow(const SceneView& sv, PathState& ps, const wpt_sphere& sp)
{
    if ((F & FEAT_ANIM) && sp.animation >= 0)
        return sphereMoved(sp, animationTrs(sv, ps, sp.animation));
    return sp;
}

struct FrameArgs {
    wpt_camera cam;
    wpt_params par;
    uint32_t width, height, samplesSqrt;
};

WPT_D void pathStateInit(PathState& ps, uint32_t pixel, uint32_t width)
{
    prngSeed(ps.prng, pixel);
    ps.acc0 = ps.acc1 = ps.acc2 = 0.0f;
    ps.px = pixel % width;
    ps.py = pixel / width;
    ps.sampleIndex = 0;
    ps.pathComponent = 0;
    ps.rayKind = RAY_PATH;
    ps.ray.o = mk3(0.0f, 0.0f, 0.0f);
    ps.ray.d = mk3(0.0f, 0.0f, 1.0f);
    ps.ray.ri = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    ps.att = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    ps.opl = mk3(0.0f, 0.0f, 0.0f);
    ps.nextAtt = ps.att;
    ps.neeFactor = ps.att;
    ps.srDir = ps.ray.d;
    ps.chosenPrim = NO_HIT;
    ps.time = 0.0f;
    ps.animCached = -1;
}

/* HitableTriangle::pdfValue (hitable_triangle.hpp:405-423) for one hot spot */
WPT_D float hotSpotPdfValue(float4 g0, float4 g1, float4 g2, f3 org, f3 dir, const RayAux& h)
{
    const f3 v0 = mk3(g0.x, g0.y, g0.z), v1 = mk3(g1.x, g1.y, g1.z), v2 = mk3(g2.x, g2.y, g2.z);
    Candidate c;
    float value = 0.0f;
    if (triangleTest(v0, v1, v2, org, h, 0.0f, k_maxval, c)) {
        f3 edgeCross = cross(sub(v1, v0), sub(v2, v0));
        float edgeCrossLength = __builtin_sqrtf(dot(edgeCross, edgeCross));
        f3 faceNormal = divs(edgeCross, edgeCrossLength);
        float faceArea = 0.5f * edgeCrossLength;
        float cosine = __builtin_fabsf(dot(faceNormal, neg(dir)));
        float distance_squared = c.a * c.a;
        value = distance_squared / (cosine * faceArea);
    }
    return value;
}

/* HitableSphere::pdfValue (hitable_sphere.hpp:149-186) */
WPT_CALL float spherePdfValue(const wpt_sphere& sp, const wpt_sphere& spHit, f3 org, f3 dir)
{
    const f3 cmo = sub(ld3(sp.center), org);
    const float distanceSquared = dot(cmo, cmo);
    const float radiusSquared = sp.radius * sp.radius;
    float value = 0.0f;
    if (distanceSquared <= radiusSquared) {
        value = 0.25f * k_inv_pi; /* inside: any direction hits */
    } else {
        float a;
        if (sphereTest(spHit, org, dir, 0.0f, k_maxval, a)) { /* this->hit(): an animated sphere is placed differently there */
            const float discriminant = 1.0f - radiusSquared / distanceSquared;
            const float cosThetaMax = discriminant > 0.0f ? __builtin_sqrtf(discriminant) : 0.0f;
            const float solidAngle = 2.0f * k_pi * (1.0f - cosThetaMax);
            value = 1.0f / solidAngle;
        }
    }
    return value;
}

/* HitableSphere::direction (hitable_sphere.hpp:188-219) */
WPT_CALL f3 sphereDirection(const wpt_sphere& sp, f3 org, Prng& prng)
{
    const f3 cmo = sub(ld3(sp.center), org);
    const float distanceSquared = dot(cmo, cmo);
    const float radiusSquared = sp.radius * sp.radius;
    if (distanceSquared <= radiusSquared)
        return onUnitSphere(in01x2(prng));
    const float discriminant = 1.0f - radiusSquared / distanceSquared;
    const float cosThetaMax = discriminant > 0.0f ? __builtin_sqrtf(discriminant) : 0.0f;
    return toSphere(normalize(cmo), cosThetaMax, in01x2(prng));
}

/* mean pdf over all hot spots of hitting them from org along dir (wurblpt.hpp:181-184) */
template<uint32_t F, bool COUNT, class Tri4>
WPT_D float hotSpotsMeanPdf(const SceneView& sv, Tri4 tri4, f3 org, f3 dir, PathState& ps, LaneCounters& lc)
{
    const RayAux h = rayAux(dir);
    float sum = 0.0f;
    for (uint32_t i = 0; i < sv.hotspotCount; i++) {
        const uint32_t p = sv.hotspots[i].prim;
        if ((F & FEAT_SPHERES) && sv.hotspots[i].kind == WPT_HOTSPOT_SPHERE) {
            const wpt_sphere& sp = sv.spheres[p];
            if ((F & FEAT_ANIM) && sp.animation >= 0) {
                const wptanim::Trs T = animationTrs(sv, ps, sp.animation);
                sum += spherePdfValue(sphereMovedForPdf(sp, T), sphereMoved(sp, T), org, dir);
            } else {
                sum += spherePdfValue(sp, sp, org, dir);
            }
        } else if ((F & FEAT_ANIM) && sv.hotspots[i].animation >= 0) {
            /* the light moves: its corners at the path's time (hitable_triangle.hpp:209-218,405-423) */
            float4 g0 = tri4(3 * p), g1 = tri4(3 * p + 1), g2 = tri4(3 * p + 2);
            const float* animationM = animationMatrix(sv, ps, sv.hotspots[i].animation);
            const f3 v0 = animatePoint(animationM, mk3(g0.x, g0.y, g0.z)), v1 = animatePoint(animationM, mk3(g1.x, g1.y, g1.z)),
                     v2 = animatePoint(animationM, mk3(g2.x, g2.y, g2.z));
            g0.x = v0.x; g0.y = v0.y; g0.z = v0.z;
            g1.x = v1.x; g1.y = v1.y; g1.z = v1.z;
            g2.x = v2.x; g2.y = v2.y; g2.z = v2.z;
            sum += hotSpotPdfValue(g0, g1, g2, org, dir, h);
        } else
            sum += hotSpotPdfValue(tri4(3 * p), tri4(3 * p + 1), tri4(3 * p + 2), org, dir, h);
        if (COUNT)
            lc.pdfs++;
    }
    sum *= 1.0f / (float)sv.hotspotCount;
    return sum;
}

/* SensorRGB::accumulateRadiance (sensor_rgb.hpp:63-80) */
WPT_D void accumulateRadiance(const wpt_params& par, f3 opl, float distanceToLight, f4 radiance, PathState& ps)
{
    const bool dOk = distanceToLight >= par.min_dist_to_light && distanceToLight <= par.max_dist_to_light;
    if (dOk && opl.x >= par.min_path_len && opl.x <= par.max_path_len)
        ps.acc0 += radiance.x;
    if (dOk && opl.y >= par.min_path_len && opl.y <= par.max_path_len)
        ps.acc1 += radiance.y;
    if (dOk && opl.z >= par.min_path_len && opl.z <= par.max_path_len)
        ps.acc2 += radiance.z;
}

/* wurblpt.hpp:254-273: continue along the scattered direction (ps.ray.o already is the hit
 * position, ps.ray.ri the index to continue with), Russian roulette */
WPT_D int advancePath(const wpt_params& par, PathState& ps)
{
    ps.att = ps.nextAtt;
    ps.ray.d = ps.srDir;
    const float mx = max4(ps.att);
    if (mx < par.rr_threshold && ps.pathComponent >= 5) {
        const float q = clampr(1.0f - mx, 0.0f, 0.95f);
        if (in01(ps.prng) < q)
            return NEXT_NEW;
        const float rrWeight = 1.0f / (1.0f - q);
        ps.att = sclr(ps.att, rrWeight);
    }
    ps.pathComponent++;
    ps.rayKind = RAY_PATH;
    return NEXT_TRACE;
}

/* wurblpt.hpp:348-360 + Camera::getRay (camera.hpp:123-185), pinhole or thin lens */
template<uint32_t F>
WPT_D int blockNew(const FrameArgs& fa, PathState& ps, const SceneView* sv = nullptr)
{
    const uint32_t samples = fa.samplesSqrt * fa.samplesSqrt;
    if (ps.sampleIndex >= samples)
        return NEXT_DONE;
    float u = (float)ps.px, v = (float)ps.py;
    if (fa.par.randomize_ray_over_pixel) {
        /* stratified jitter; the reference compiler draws the vertical stratum first */
        const uint32_t j = ps.sampleIndex / fa.samplesSqrt;
        const uint32_t i = ps.sampleIndex % fa.samplesSqrt;
        const float fj = (float)j + in01(ps.prng);
        const float fi = (float)i + in01(ps.prng);
        const float invSamplesSqrt = 1.0f / (float)fa.samplesSqrt;
        u += fi * invSamplesSqrt;
        v += fj * invSamplesSqrt;
    } else {
        u += 0.5f;
        v += 0.5f;
    }
    u *= 1.0f / (float)fa.width;
    v *= 1.0f / (float)fa.height;
    /* Camera::getRay (camera.hpp:123-185) */
    float stereoscopicShift = 0.0f;
    if ((F & FEAT_LENS) && fa.cam.stereoscopic_distance > 0.0f) {
        v *= 2.0f; /* left view in the upper half, right view in the lower half */
        if (v < 1.0f) {
            stereoscopicShift = -0.5f * fa.cam.stereoscopic_distance;
        } else {
            v -= 1.0f;
            stereoscopicShift = +0.5f * fa.cam.stereoscopic_distance;
        }
    }
    f3 O, D;
    if ((F & FEAT_LENS) && fa.cam.surround_mode != WPT_SURROUND_OFF) {
        /* direction from longitude and latitude; the optics are ignored */
        float lon = (2.0f * u - 1.0f) * k_pi;
        if (fa.cam.surround_mode == WPT_SURROUND_180)
            lon *= 0.5f;
        const float lat = (v - 0.5f) * k_pi;
        const float clat = wptm::cosf_(lat), slat = wptm::sinf_(lat), clon = wptm::cosf_(lon), slon = wptm::sinf_(lon);
        D = mk3(clat * slon, slat, -clat * clon);
        O = sclr(mk3(-clon, 0.0f, -slon), stereoscopicShift);
    } else {
        /* the samples lie in the distorted output image: rays are made from the undistorted coordinates */
        if ((F & FEAT_LENS) && fa.cam.distortion_type != WPT_DISTORTION_NONE)
            wptlens::undistort(fa.cam, u, v, fa.width, fa.height);
        f3 P = mk3(mixr(fa.cam.l, fa.cam.r, u), mixr(fa.cam.b, fa.cam.t, v), -1.0f);
        O = mk3(0.0f, 0.0f, 0.0f);
        if ((F & FEAT_LENS) && fa.cam.lens_radius > 0.0f) {
            P = sclr(P, fa.cam.focus_dist);
            f2 d = inUnitDisk(in01x2(ps.prng));
            O = mk3(fa.cam.lens_radius * d.x, fa.cam.lens_radius * d.y, 0.0f);
        }
        D = sub(P, O);
        O = add(O, mk3(stereoscopicShift, 0.0f, 0.0f));
    }
    if ((F & FEAT_ANIM) && fa.par.t0 != fa.par.t1) {
        /* camera.hpp:175-184: the ray draws its time in the exposure interval; a moving camera is taken at that time */
        const float t = fa.par.t0 + in01(ps.prng) * (fa.par.t1 - fa.par.t0);
        ps.time = t;
        ps.animCached = -1; /* AnimationCache::init(r.time) */
        if (fa.cam.animation >= 0 && sv) {
            const wptanim::Trs T = animationAt(*sv, fa.cam.animation, t);
            ps.ray.o = add(ld3(T.t), quatRotate(T.q, mul(O, ld3(T.s))));
            ps.ray.d = normalize(quatRotate(T.q, D));
        } else {
            ps.ray.o = add(ld3(fa.cam.translation), quatRotate(fa.cam.rotation, mul(O, ld3(fa.cam.scaling))));
            ps.ray.d = normalize(quatRotate(fa.cam.rotation, D));
        }
    } else {
        if (F & FEAT_ANIM)
            ps.time = fa.par.t0;
        ps.ray.o = add(ld3(fa.cam.translation), quatRotate(fa.cam.rotation, mul(O, ld3(fa.cam.scaling))));
        ps.ray.d = normalize(quatRotate(fa.cam.rotation, D));
    }
    ps.ray.ri = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    ps.att = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    ps.opl = mk3(0.0f, 0.0f, 0.0f);
    ps.pathComponent = 0;
    ps.sampleIndex++;
    ps.rayKind = RAY_PATH;
    return NEXT_TRACE;
}

/* tracePath, one path component (wurblpt.hpp:131-252); `best` is the path ray's result */
template<uint32_t F, bool COUNT, class Tri4>
WPT_D int blockShade(const SceneView& sv, const wpt_params& par, Tri4 tri4, PathState& ps, const Candidate& best, LaneCounters& lc)
{
    const bool haveEnv = (F & FEAT_ENVMAP) && sv.envType != WPT_ENV_NONE;
    if (best.prim == NO_HIT) {
        if (haveEnv) {
            f4 rad = mul(ps.att, envL(sv, ps.ray.d));
            accumulateRadiance(par, mk3(k_maxval, k_maxval, k_maxval), k_maxval, rad, ps);
        }
        return NEXT_NEW;
    }
    ps.opl = add(ps.opl, scl(best.a, mk3(ps.ray.ri.x, ps.ray.ri.y, ps.ray.ri.z)));
    if (!(ps.pathComponent + 1 < par.max_path_components))
        return NEXT_NEW;
    long long tSection = 0;
    auto section = [&](int k) { /* COUNT builds: close section k */
        if (COUNT) {
            const long long now = clock64();
            lc.shadeClock[k] += (unsigned long long)(now - tSection);
            tSection = now;
        }
    };
    if (COUNT)
        tSection = clock64();
    Hit h = finishHit<F>(sv, best, ps.ray.o, ps.ray.d, ps.time);
    const wpt_material& m = resolveMaterial<F>(sv, h.material, h);
    if (COUNT)
        lc.scatters++;
    section(0);
    const Scatter sr = materialScatter<F>(sv, m, ps.ray, h, ps.prng);
    section(1);
    {
        f4 rad = mul(ps.att, materialEmitted<F>(sv, m, h));
        accumulateRadiance(par, ps.opl, (ps.pathComponent == 0 ? 0.0f : h.a), rad, ps);
    }
    section(2);
    if (sr.type == SCATTER_NONE)
        return NEXT_NEW;
    ps.nextAtt = mul(ps.att, sr.att);
    if (sr.type == SCATTER_RANDOM) {
        if (sr.pdf > 0.0f)
            ps.nextAtt = divs(ps.nextAtt, sr.pdf);
        else
            ps.nextAtt = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    ps.srDir = sr.dir;
    if (sr.type == SCATTER_RANDOM && sv.hotspotCount > 0) {
        /* light sampling with MIS (wurblpt.hpp:179-220) */
        const float hotSpotsPdf = hotSpotsMeanPdf<F, COUNT>(sv, tri4, h.p, sr.dir, ps, lc);
        ps.nextAtt = sclr(ps.nextAtt, powerHeuristicWeight(sr.pdf, hotSpotsPdf));
        section(3);
        uint32_t idx = (uint32_t)(in01(ps.prng) * (float)sv.hotspotCount);
        idx = idx < sv.hotspotCount - 1 ? idx : sv.hotspotCount - 1;
        const wpt_hotspot& hs = sv.hotspots[idx];
        f3 directDir;
        uint32_t hotSpotPrim = hs.prim;
        if ((F & FEAT_SPHERES) && hs.kind == WPT_HOTSPOT_SPHERE) {
            directDir = sphereDirection(sphereNow<F>(sv, ps, sv.spheres[hs.prim]), h.p, ps.prng);
            hotSpotPrim = PRIM_SPHERE | hs.prim;
        } else {
            /* HitableTriangle::direction (hitable_triangle.hpp:425-443) */
            const f3 bary = inTriangle(in01x2(ps.prng));
            f3 p = add(add(scl(bary.x, ld3(hs.p0)), scl(bary.y, ld3(hs.p1))), scl(bary.z, ld3(hs.p2)));
            if (hs.transform)
                p = mat4mulPoint(hs.M, p);
            if ((F & FEAT_ANIM) && hs.animation >= 0)
                p = animatePoint(animationMatrix(sv, ps, hs.animation), p);
            directDir = normalize(sub(p, h.p));
        }
        section(4);
        const float directPdf = hotSpotsMeanPdf<F, COUNT>(sv, tri4, h.p, directDir, ps, lc);
        section(5);
        if (directPdf > 0.0f) {
            float dpdf;
            f4 directAtt;
            materialEval<F>(sv, m, ps.ray, h, directDir, directAtt, dpdf);
            if (dpdf > 0.0f) {
                ps.neeFactor = sclr(divs(mul(ps.att, directAtt), directPdf), powerHeuristicWeight(directPdf, dpdf));
                ps.chosenPrim = hotSpotPrim;
                ps.ray.o = h.p;
                ps.ray.d = directDir;
                ps.rayKind = RAY_NEE_LIGHT;
                section(6);
                return NEXT_TRACE;
            }
        }
    } else if ((F & FEAT_ENVMAP) && sr.type == SCATTER_RANDOM && haveEnv && sv.envN > 0) {
        /* environment sampling with MIS (wurblpt.hpp:221-252) */
        const float lightsP = envP(sv, sr.dir);
        ps.nextAtt = sclr(ps.nextAtt, powerHeuristicWeight(sr.pdf, lightsP));
        const f3 lightDir = envD(sv, ps.prng);
        const float directPdf = envP(sv, lightDir);
        float dpdf;
        f4 directAtt;
        materialEval<F>(sv, m, ps.ray, h, lightDir, directAtt, dpdf);
        if (dpdf > 0.0f) {
            ps.neeFactor = sclr(divs(mul(ps.att, directAtt), directPdf), powerHeuristicWeight(directPdf, dpdf));
            ps.ray.o = h.p;
            ps.ray.d = lightDir;
            ps.rayKind = RAY_NEE_ENV;
            section(7);
            return NEXT_TRACE;
        }
    }
    /* No next-event ray.  The scattered ray's refractive index: every ScatterRandom record
     * carries the incoming ray's index unchanged (material_lambertian.hpp:83, material_ggx.hpp:224,
     * material_modphong.hpp:307), so while a next-event ray is in flight ray.ri already is the
     * value to continue with; only explicit scattering (glass, transparent ModPhong) changes it. */
    ps.ray.o = h.p;
    ps.ray.ri = sr.ri;
    const int next = advancePath(par, ps);
    section(7);
    return next;
}

/* the next-event ray's result (wurblpt.hpp:208-218: only the CHOSEN hot spot as nearest hit
 * counts; :240-250: the environment counts if nothing was hit), then the path continues */
template<uint32_t F>
WPT_D int blockNeeEnd(const SceneView& sv, const wpt_params& par, PathState& ps, const Candidate& best)
{
    if (ps.rayKind == RAY_NEE_LIGHT) {
        if (best.prim == ps.chosenPrim) {
            Hit lh = finishHit<F>(sv, best, ps.ray.o, ps.ray.d, ps.time);
            const wpt_material& lm = resolveMaterial<F>(sv, lh.material, lh);
            f4 rad = mul(ps.neeFactor, materialEmitted<F>(sv, lm, lh));
            f3 oplLight = add(ps.opl, scl(lh.a, mk3(ps.ray.ri.x, ps.ray.ri.y, ps.ray.ri.z)));
            accumulateRadiance(par, oplLight, lh.a, rad, ps);
        }
    } else if (F & FEAT_ENVMAP) {
        if (best.prim == NO_HIT) {
            f4 rad = mul(ps.neeFactor, envL(sv, ps.ray.d));
            accumulateRadiance(par, mk3(k_maxval, k_maxval, k_maxval), k_maxval, rad, ps);
        }
    }
    return advancePath(par, ps);
}

} /* namespace wptk */

#endif
