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
 * what the lane needs next: a ray to be traced (ps.o / ps.d / ps.rayKind are set), a new sample, or
 * nothing more (all samples done).  The traversal of the ray is the kernel's business.
 *
 * Where a path's state lives.  A lane walks one pixel's paths; between two blocks it traverses, and all it
 * needs for that is the ray.  Everything else a path carries (generator, accumulators, attenuations, the
 * continuation while a next-event ray is in flight: 32 words) is COLD while the lane traverses, and in the
 * blocks each word is needed at one or two places only.  The path tracing kernels therefore keep the cold
 * words in LDS (PathLds: eight 16-byte slots per lane, lane-major, so that every access is a conflict-free
 * ds_read_b128 / ds_write_b128) and the blocks below fetch a slot where they use it and store it where its
 * value is final.  Registers then hold what the running block works on, not what other lanes of the wave
 * will need later: that is what used to spill to scratch memory at four waves per SIMD.  The ground truth
 * kernel, one ray per lane, uses the same blocks over plain registers (PathRegs).
 */
#ifndef WPT_BLOCKS_H
#define WPT_BLOCKS_H

#include "wpt_device.h"
#include "wpt_lens.h"

namespace wptk {

using namespace wptd;

constexpr uint32_t NO_HIT = 0xffffffffu;

enum { RAY_PATH = 0, RAY_NEE_LIGHT = 1, RAY_NEE_ENV = 2, RAY_PATH_WAITED = 3 /* a path ray's hit that has stood back once */ };
enum { NEXT_TRACE = 0, NEXT_NEW = 1, NEXT_DONE = 2, NEXT_WAIT = 3 };

/* COUNT builds: how often each stretch of the kernel's code runs, for the instruction budget (tools/instruction_budget.py multiplies
 * them with the stretches' instruction counts from the assembly): executions by a wave (at least one lane in it) and by lanes */
enum { SEC_NODE_STEP = 0, SEC_LEAF_TEST, SEC_BEGIN_RAY, SEC_MISS, SEC_HIT_RECORD, SEC_SCATTER_LAMBERT, SEC_SCATTER_GGX, SEC_SCATTER_GLASS, SEC_SCATTER_OTHER,
       SEC_EMISSION, SEC_LIGHT_SAMPLE, SEC_PDF_SETUP, SEC_PDF_LIGHT, SEC_EVAL_LAMBERT, SEC_EVAL_GGX, SEC_EVAL_OTHER, SEC_NEE_SETUP, SEC_ADVANCE, SEC_NEE_END,
       SEC_NEE_END_LIGHT, SEC_NEW_SAMPLE, SEC_PIXEL_DONE, SEC_LOOK, SEC_LOOK_INNER, SEC_COUNT };

struct LaneCounters {
    uint32_t rays, nodes, leaves, pdfs, scatters;
    /* COUNT builds: shader clock this lane spent in the sections of blockShade (profiling only):
     * [0] hit record + material, [1] scatter, [2] emission, [3] light pdf of the scattered direction,
     * [4] light sample, [5] light pdf of the light direction, [6] evaluation towards the light,
     * [7] environment sampling / continuation */
    unsigned long long shadeClock[8];
    uint32_t secWave[SEC_COUNT], secLane[SEC_COUNT];
};
constexpr LaneCounters LANE_COUNTERS_ZERO = {};

/* called by every lane that is active around stretch k; `mine`: this lane runs it */
template<bool COUNT> WPT_D void sec(LaneCounters& lc, int k, bool mine = true)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (COUNT) {
        const unsigned long long lanes = __ballot(mine);
        if (mine) {
            lc.secLane[k]++;
            const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if ((int)lane == __ffsll((long long)lanes) - 1)
                lc.secWave[k]++;
        }
    }
#endif
}

/* ---- the cold words of a path, by slot (x, y, z, w) ----
 *   0 prng s0 s1 s2 s3        1 att                      2 ray.ri (refractiveIndex)    3 nextAtt
 *   4 acc r g b, sample       5 opl x y z, pathComponent 6 neeFactor r g b, chosenPrim 7 srDir x y z, pixel
 * sample = the next sample's stratum, column | row << 16 (row == samplesSqrt: all done); pixel = x | y << 16
 * (launches check samples_sqrt, width and height against 65535): no division per sample.
 * opticalPathLength and the next-event factor have three channels here: SensorRGB reads channels 0..2 only
 * (sensor_rgb.hpp:63-80), and nothing else reads their fourth. */
enum { SLOT_PRNG = 0, SLOT_ATT = 1, SLOT_RI = 2, SLOT_NEXTATT = 3, SLOT_ACC = 4, SLOT_OPL = 5, SLOT_NEE = 6, SLOT_SRDIR = 7, SLOT_COUNT = 8 };

struct Slot {
    float x, y, z;
    uint32_t w;
};

/* what both kinds of path state share: the words that stay in registers */
struct PathHot {
    f3 o, d;      /* the ray being traced (Ray::origin, Ray::direction) */
    int rayKind;
    /* wavefront form (blockShade<..., MERGED>): the light ray of the hit just shaded, which travels BESIDE the path's continuation;
     * neeKind: 0 = none, RAY_NEE_LIGHT, RAY_NEE_ENV */
    f3 neeO, neeD;
    int neeKind;
    float time;   /* FEAT_ANIM: the path's time (Ray::time; the thread's AnimationCache is set to it, wurblpt.hpp:361) */
    /* FEAT_ANIM: the lane's AnimationCache, one entry deep -- the matrix of the animation it used last at `time`.
     * Consecutive leaf tests of a walk mostly hit triangles of one instance, and the lights share few animations. */
    int animCached;
    float animM[16];
};

/* cold words in LDS: `base` is this lane's first slot, slot k lies k * STRIDE float4 further */
template<int STRIDE> struct PathLds : PathHot {
    float4* base;
    WPT_D f4 get4(int k) const { const float4 v = base[k * STRIDE]; return mk4(v.x, v.y, v.z, v.w); }
    WPT_D void set4(int k, f4 v) { base[k * STRIDE] = make_float4(v.x, v.y, v.z, v.w); }
    WPT_D Slot get(int k) const
    {
        const float4 v = base[k * STRIDE];
        Slot s;
        s.x = v.x; s.y = v.y; s.z = v.z; s.w = __float_as_uint(v.w);
        return s;
    }
    WPT_D void set(int k, Slot s) { base[k * STRIDE] = make_float4(s.x, s.y, s.z, __uint_as_float(s.w)); }
    WPT_D void set3(int k, f3 v) /* x, y, z only: w keeps its value */
    {
        float* p = reinterpret_cast<float*>(base + k * STRIDE);
        p[0] = v.x; p[1] = v.y; p[2] = v.z;
    }
    WPT_D uint32_t getW(int k) const { return reinterpret_cast<const uint32_t*>(base + k * STRIDE)[3]; }
    WPT_D void setW(int k, uint32_t w) { reinterpret_cast<uint32_t*>(base + k * STRIDE)[3] = w; }
};

/* cold words in registers (one ray per lane: the ground truth kernel) */
struct PathRegs : PathHot {
    Slot slot[SLOT_COUNT];
    WPT_D f4 get4(int k) const { return mk4(slot[k].x, slot[k].y, slot[k].z, __uint_as_float(slot[k].w)); }
    WPT_D void set4(int k, f4 v) { slot[k].x = v.x; slot[k].y = v.y; slot[k].z = v.z; slot[k].w = __float_as_uint(v.w); }
    WPT_D Slot get(int k) const { return slot[k]; }
    WPT_D void set(int k, Slot s) { slot[k] = s; }
    WPT_D void set3(int k, f3 v) { slot[k].x = v.x; slot[k].y = v.y; slot[k].z = v.z; }
    WPT_D uint32_t getW(int k) const { return slot[k].w; }
    WPT_D void setW(int k, uint32_t w) { slot[k].w = w; }
};

template<class PS> WPT_D Prng loadPrng(const PS& ps)
{
    const Slot s = ps.get(SLOT_PRNG);
    Prng p;
    p.s0 = __float_as_uint(s.x); p.s1 = __float_as_uint(s.y); p.s2 = __float_as_uint(s.z); p.s3 = s.w;
    return p;
}
template<class PS> WPT_D void storePrng(PS& ps, const Prng& p)
{
    Slot s;
    s.x = __uint_as_float(p.s0); s.y = __uint_as_float(p.s1); s.z = __uint_as_float(p.s2); s.w = p.s3;
    ps.set(SLOT_PRNG, s);
}

/* AnimationCache::getM(ai) at the path's time.  Spheres do not go through it: measured on the test scene, sharing the
 * entry with them lets sphere and triangle leaves evict each other (162 -> 141 Msamples/s) and an entry of their own
 * is evicted by the next sphere (158), so their transformation is evaluated where it is needed. */
WPT_D wptanim::Trs animationTrs(const SceneView& sv, const PathHot& ps, int ai)
{
    return animationAt(sv, ai, ps.time);
}
WPT_D const float* animationMatrix(const SceneView& sv, PathHot& ps, int ai)
{
    if (ps.animCached != ai) {
        wptanim::toMat4(animationAt(sv, ai, ps.time), ps.animM);
        ps.animCached = ai;
    }
    return ps.animM;
}
/* an animated sphere at the path's time, through the lane's cache: as hit() / direction() place it */
template<uint32_t F> WPT_D wpt_sphere sphereNow(const SceneView& sv, const PathHot& ps, const wpt_sphere& sp)
{
    if ((F & FEAT_ANIM) && sp.animation >= 0)
        return sphereMoved(sp, animationTrs(sv, ps, sp.animation));
    return sp;
}

struct FrameArgs {
    wpt_camera cam;
    wpt_params par;
    uint32_t width, height, samplesSqrt;
    /* 1.0f / (float)width, / height, / samplesSqrt: divided once on the host (IEEE, the same bits) */
    float invWidth, invHeight, invSamplesSqrt;
};

/* A uniform value as the compiler must take it where it stands (a scalar register, no instruction).  Values that
 * depend only on the launch arguments are loop invariant, float arithmetic on them has no scalar form on gfx950,
 * so the compiler computes e.g. a pinhole camera's origin once in front of the kernel's loop into VECTOR registers
 * that then stay occupied (or spill) for the whole kernel.  Behind this fence the arithmetic stays where it is used. */
WPT_D float here(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* (the value is the same in all lanes; should the compiler hold it in a vector register, this takes it from there) */
    int bits = __builtin_amdgcn_readfirstlane(__float_as_int(x));
    asm volatile("" : "+s"(bits));
    x = __int_as_float(bits);
#endif
    return x;
}

template<class PS> WPT_D void pathStateInit(PS& ps, uint32_t pixel, uint32_t px, uint32_t py)
{
    Prng prng;
    prngSeed(prng, pixel);
    storePrng(ps, prng);
    const f4 one = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    ps.set4(SLOT_ATT, one);
    ps.set4(SLOT_RI, one);
    ps.set4(SLOT_NEXTATT, one);
    Slot z;
    z.x = z.y = z.z = 0.0f;
    z.w = 0;
    ps.set(SLOT_ACC, z); /* stratum (0, 0) */
    ps.set(SLOT_OPL, z); /* pathComponent 0 */
    Slot n;
    n.x = n.y = n.z = 1.0f;
    n.w = NO_HIT;
    ps.set(SLOT_NEE, n);
    Slot s;
    s.x = 0.0f; s.y = 0.0f; s.z = 1.0f;
    s.w = px | (py << 16);
    ps.set(SLOT_SRDIR, s);
    ps.o = mk3(0.0f, 0.0f, 0.0f);
    ps.d = mk3(0.0f, 0.0f, 1.0f);
    ps.rayKind = RAY_PATH;
    ps.neeKind = 0;
    ps.time = 0.0f;
    ps.animCached = -1;
}

/* What HitableTriangle::pdfValue (hitable_triangle.hpp:405-423) derives from the corners alone: the unit face normal and
 * the face area.  Evaluated once per hot spot at upload (wpt_capi.hip) with exactly these operations; a moving light's
 * corners change, there they are evaluated per call. */
WPT_D float4 hotSpotFace(f3 v0, f3 v1, f3 v2)
{
    f3 edgeCross = cross(sub(v1, v0), sub(v2, v0));
    float edgeCrossLength = __builtin_sqrtf(dot(edgeCross, edgeCross));
    f3 faceNormal = divs(edgeCross, edgeCrossLength);
    float faceArea = 0.5f * edgeCrossLength;
    return make_float4(faceNormal.x, faceNormal.y, faceNormal.z, faceArea);
}

/* HitableTriangle::pdfValue (hitable_triangle.hpp:405-423) for one hot spot; FACE: `face` holds hotSpotFace() of the corners */
template<bool FACE>
WPT_D float hotSpotPdfValue(float4 g0, float4 g1, float4 g2, f3 org, f3 dir, const RayAux& h, float4 face)
{
    const f3 v0 = mk3(g0.x, g0.y, g0.z), v1 = mk3(g1.x, g1.y, g1.z), v2 = mk3(g2.x, g2.y, g2.z);
    Candidate c;
    float value = 0.0f;
    if (triangleTest(v0, v1, v2, org, h, 0.0f, k_maxval, c)) {
        if (!FACE)
            face = hotSpotFace(v0, v1, v2);
        float cosine = __builtin_fabsf(dot(mk3(face.x, face.y, face.z), neg(dir)));
        float distance_squared = c.a * c.a;
        value = distance_squared / (cosine * face.w);
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

/* HitableSphere::direction (hitable_sphere.hpp:188-219).  The generator travels by value, in and out: a reference
 * would pin the caller's copy to scratch memory, because this is a real call. */
struct DirectionDraw {
    f3 dir;
    Prng prng;
};
WPT_CALL DirectionDraw sphereDirection(const wpt_sphere& sp, f3 org, Prng prng)
{
    DirectionDraw r;
    const f3 cmo = sub(ld3(sp.center), org);
    const float distanceSquared = dot(cmo, cmo);
    const float radiusSquared = sp.radius * sp.radius;
    if (distanceSquared <= radiusSquared) {
        r.dir = onUnitSphere(in01x2(prng));
    } else {
        const float discriminant = 1.0f - radiusSquared / distanceSquared;
        const float cosThetaMax = discriminant > 0.0f ? __builtin_sqrtf(discriminant) : 0.0f;
        r.dir = toSphere(normalize(cmo), cosThetaMax, in01x2(prng));
    }
    r.prng = prng;
    return r;
}

/* Mean pdf over all hot spots of hitting them from org, for two directions at once: the scattered direction and the
 * direction towards the sampled light (wurblpt.hpp:181-184 and :196-199).  Every light's corners are fetched once, and
 * the two tests, which share nothing but them, run side by side; each direction's sum adds its terms in the order of the
 * hot spots, as the reference's loop does. */
template<uint32_t F, bool COUNT, class Tri4>
WPT_D void hotSpotsMeanPdfPair(const SceneView& sv, Tri4 tri4, f3 org, f3 dirA, f3 dirB, PathHot& ps, LaneCounters& lc, float& meanA, float& meanB)
{
    const RayAux hA = rayAux<true>(dirA), hB = rayAux<true>(dirB);
    float sumA = 0.0f, sumB = 0.0f;
    for (uint32_t i = 0; i < sv.hotspotCount; i++) {
        sec<COUNT>(lc, SEC_PDF_LIGHT);
        const uint32_t p = sv.hotspots[i].prim;
        if ((F & FEAT_SPHERES) && sv.hotspots[i].kind == WPT_HOTSPOT_SPHERE) {
            const wpt_sphere& sp = sv.spheres[p];
            if ((F & FEAT_ANIM) && sp.animation >= 0) {
                const wptanim::Trs T = animationTrs(sv, ps, sp.animation);
                const wpt_sphere forPdf = sphereMovedForPdf(sp, T), moved = sphereMoved(sp, T);
                sumA += spherePdfValue(forPdf, moved, org, dirA);
                sumB += spherePdfValue(forPdf, moved, org, dirB);
            } else {
                sumA += spherePdfValue(sp, sp, org, dirA);
                sumB += spherePdfValue(sp, sp, org, dirB);
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
            sumA += hotSpotPdfValue<false>(g0, g1, g2, org, dirA, hA, g0);
            sumB += hotSpotPdfValue<false>(g0, g1, g2, org, dirB, hB, g0);
        } else {
            const float4 g0 = tri4(3 * p), g1 = tri4(3 * p + 1), g2 = tri4(3 * p + 2), face = sv.hotspotFace[i];
            sumA += hotSpotPdfValue<true>(g0, g1, g2, org, dirA, hA, face);
            sumB += hotSpotPdfValue<true>(g0, g1, g2, org, dirB, hB, face);
        }
        if (COUNT)
            lc.pdfs += 2;
    }
    meanA = sumA * here(sv.invHotspotCount);
    meanB = sumB * here(sv.invHotspotCount);
}

/* SensorRGB::accumulateRadiance (sensor_rgb.hpp:63-80): read - add - write of the accumulator slot; a closed
 * distance gate adds nothing, so the slot is not touched then */
template<class PS> WPT_D void accumulateRadiance(const wpt_params& par, f3 opl, float distanceToLight, f4 radiance, PS& ps)
{
    const bool dOk = distanceToLight >= par.min_dist_to_light && distanceToLight <= par.max_dist_to_light;
    if (!dOk)
        return;
    Slot acc = ps.get(SLOT_ACC);
    if (opl.x >= par.min_path_len && opl.x <= par.max_path_len)
        acc.x += radiance.x;
    if (opl.y >= par.min_path_len && opl.y <= par.max_path_len)
        acc.y += radiance.y;
    if (opl.z >= par.min_path_len && opl.z <= par.max_path_len)
        acc.z += radiance.z;
    ps.set3(SLOT_ACC, mk3(acc.x, acc.y, acc.z));
}

/* wurblpt.hpp:254-273: continue along the scattered direction (ps.o already is the hit position, the ray's
 * refractive index the one to continue with), Russian roulette.  nextAtt / srDir / pathComponent are the
 * path's values, handed over by the caller who has them at hand; prng is the caller's copy of the generator
 * (the caller stores it afterwards). */
template<class PS> WPT_D int advancePath(const wpt_params& par, PS& ps, f4 nextAtt, f3 srDir, uint32_t pathComponent, Prng& prng)
{
    f4 att = nextAtt;
    ps.d = srDir;
    const float mx = max4(att);
    if (mx < par.rr_threshold && pathComponent >= 5) {
        const float q = clampr(1.0f - mx, 0.0f, 0.95f);
        if (in01(prng) < q)
            return NEXT_NEW;
        const float rrWeight = 1.0f / (1.0f - q);
        att = sclr(att, rrWeight);
    }
    ps.set4(SLOT_ATT, att);
    ps.setW(SLOT_OPL, pathComponent + 1);
    ps.rayKind = RAY_PATH;
    return NEXT_TRACE;
}

/* wurblpt.hpp:348-360 + Camera::getRay (camera.hpp:123-185), pinhole or thin lens */
template<uint32_t F, class PS>
WPT_D int blockNew(const FrameArgs& fa, PS& ps, const SceneView& sv)
{
    const uint32_t stratum = ps.getW(SLOT_ACC);
    const uint32_t i = stratum & 0xffffu, j = stratum >> 16; /* sampleIndex % samplesSqrt, sampleIndex / samplesSqrt */
    if (j >= fa.samplesSqrt)
        return NEXT_DONE;
    const uint32_t pxy = ps.getW(SLOT_SRDIR);
    Prng prng = loadPrng(ps);
    float u = (float)(pxy & 0xffffu), v = (float)(pxy >> 16);
    if (fa.par.randomize_ray_over_pixel) {
        /* stratified jitter; the reference compiler draws the vertical stratum first */
        const float fj = (float)j + in01(prng);
        const float fi = (float)i + in01(prng);
        u += fi * fa.invSamplesSqrt;
        v += fj * fa.invSamplesSqrt;
    } else {
        u += 0.5f;
        v += 0.5f;
    }
    u *= fa.invWidth;
    v *= fa.invHeight;
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
            wptlens::undistort(fa.cam, u, v, fa.width, fa.height); /* a real call; the coefficients travel by value */
        f3 P = mk3(mixr(here(fa.cam.l), here(fa.cam.r), u), mixr(here(fa.cam.b), here(fa.cam.t), v), -1.0f);
        O = mk3(0.0f, 0.0f, 0.0f);
        if ((F & FEAT_LENS) && fa.cam.lens_radius > 0.0f) {
            P = sclr(P, fa.cam.focus_dist);
            f2 d = inUnitDisk(in01x2(prng));
            O = mk3(fa.cam.lens_radius * d.x, fa.cam.lens_radius * d.y, 0.0f);
        }
        D = sub(P, O);
        O = add(O, mk3(stereoscopicShift, 0.0f, 0.0f));
    }
    if ((F & FEAT_ANIM) && fa.par.t0 != fa.par.t1) {
        /* camera.hpp:175-184: the ray draws its time in the exposure interval; a moving camera is taken at that time */
        const float t = fa.par.t0 + in01(prng) * (fa.par.t1 - fa.par.t0);
        ps.time = t;
        ps.animCached = -1; /* AnimationCache::init(r.time) */
        if (fa.cam.animation >= 0) {
            /* (the scene view comes by reference: a POINTER to it, as this function once took, made the compiler keep a private
             * copy of all kernel arguments in scratch memory -- 880 bytes per lane in the ground truth kernel, 670 - 930 in the
             * path tracing kernels for moving scenes) */
            const wptanim::Trs T = animationAt(sv, fa.cam.animation, t);
            ps.o = add(ld3(T.t), quatRotate(T.q, mul(O, ld3(T.s))));
            ps.d = normalize(quatRotate(T.q, D));
        } else {
            ps.o = add(ld3(fa.cam.translation), quatRotate(fa.cam.rotation, mul(O, ld3(fa.cam.scaling))));
            ps.d = normalize(quatRotate(fa.cam.rotation, D));
        }
    } else {
        if (F & FEAT_ANIM)
            ps.time = fa.par.t0;
        const float rotation[4] = { here(fa.cam.rotation[0]), here(fa.cam.rotation[1]), here(fa.cam.rotation[2]), here(fa.cam.rotation[3]) };
        const f3 translation = mk3(here(fa.cam.translation[0]), here(fa.cam.translation[1]), here(fa.cam.translation[2]));
        const f3 scaling = mk3(here(fa.cam.scaling[0]), here(fa.cam.scaling[1]), here(fa.cam.scaling[2]));
        ps.o = add(translation, quatRotate(rotation, mul(O, scaling)));
        ps.d = normalize(quatRotate(rotation, D));
    }
    storePrng(ps, prng);
    const f4 one = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    ps.set4(SLOT_RI, one);
    ps.set4(SLOT_ATT, one);
    Slot opl;
    opl.x = opl.y = opl.z = 0.0f;
    opl.w = 0; /* pathComponent */
    ps.set(SLOT_OPL, opl);
    ps.setW(SLOT_ACC, i + 1 < fa.samplesSqrt ? stratum + 1 : (j + 1) << 16);
    ps.rayKind = RAY_PATH;
    return NEXT_TRACE;
}

/* tracePath, one path component (wurblpt.hpp:131-252); `best` is the path ray's result.
 * MERGED (the wavefront form): a light ray does not hold the path up.  What follows its end in the reference -- the Russian roulette
 * and the continuation, wurblpt.hpp:254-273 -- depends on the scatter alone, not on the light ray's answer, and the roulette's draw
 * is the generator's next one either way (nothing draws while the light ray is traced); so the block goes on to advancePath at once
 * and leaves the light ray in ps.neeO / neeD / neeKind to be traced beside the continuation.  Its answer is added by blockNeeResult
 * at the start of the pixel's next shading, before anything else touches the accumulator: the additions keep the reference's order.
 * What that needs of THIS hit -- the factor and the chosen hot spot (SLOT_NEE), the optical path length and the refractive index --
 * waits in the two slots the single kernel uses for the continuation (SLOT_NEXTATT, SLOT_SRDIR x y z), which are free here. */
template<uint32_t F, bool COUNT, class Tri4, class PS, bool MERGED = false>
WPT_D int blockShade(const SceneView& sv, const wpt_params& par, Tri4 tri4, PS& ps, const Candidate& best, LaneCounters& lc, int waitBelow = 0)
{
    const bool haveEnv = (F & FEAT_ENVMAP) && sv.envType != WPT_ENV_NONE;
    sec<COUNT>(lc, SEC_MISS, best.prim == NO_HIT);
    if (best.prim == NO_HIT) {
        if (haveEnv) {
            f4 rad = mul(ps.get4(SLOT_ATT), envL(sv, ps.d));
            accumulateRadiance(par, mk3(k_maxval, k_maxval, k_maxval), k_maxval, rad, ps);
        }
        return NEXT_NEW;
    }
    Ray ray;
    ray.o = ps.o;
    ray.d = ps.d;
    ray.ri = ps.get4(SLOT_RI);
    const Slot oplSlot = ps.get(SLOT_OPL);
    const uint32_t pathComponent = oplSlot.w;
    const f3 opl = add(mk3(oplSlot.x, oplSlot.y, oplSlot.z), scl(best.a, mk3(ray.ri.x, ray.ri.y, ray.ri.z)));
    if (!(pathComponent + 1 < par.max_path_components))
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
    sec<COUNT>(lc, SEC_HIT_RECORD);
    Hit h = finishHit<F>(sv, best, ray.o, ray.d, ps.time, tri4);
    const wpt_material& m = resolveMaterial<F>(sv, h.material, h);
    if ((F & ~(FEAT_GGX | FEAT_GLASS)) == 0 && waitBelow > 0) { /* the all-features builds have no register to spare for it */
        /* Each kind of material is its own stretch of code below, as long for one lane as for
         * forty.  A kind with few lanes in this round, next to lanes of other kinds, stands back
         * once (nothing has been written yet): the next round then runs it for two rounds' worth
         * of lanes.  Which lanes run together never changes what a lane computes. */
        const bool waited = ps.rayKind == RAY_PATH_WAITED;
        const unsigned long long here = __ballot(true);
        bool wait = false;
        for (uint32_t kind = WPT_MAT_LAMBERTIAN; kind <= WPT_MAT_RGL; kind++) {
            if (kind == WPT_MAT_LIGHT_DIFFUSE || kind == WPT_MAT_TWOSIDED) /* a light ends the path at once; two-sided has been resolved */
                continue;
            const unsigned long long lanes = __ballot(m.type == kind);
            if (lanes != 0 && lanes != here && __popcll(lanes) < waitBelow && __ballot(m.type == kind && waited) == 0 && m.type == kind)
                wait = true;
        }
        if (wait) {
            ps.rayKind = RAY_PATH_WAITED;
            return NEXT_WAIT;
        }
    }
    ps.set3(SLOT_OPL, opl);
    if (COUNT)
        lc.scatters++;
    section(0);
    sec<COUNT>(lc, SEC_SCATTER_LAMBERT, m.type == WPT_MAT_LAMBERTIAN);
    sec<COUNT>(lc, SEC_SCATTER_GGX, m.type == WPT_MAT_GGX);
    sec<COUNT>(lc, SEC_SCATTER_GLASS, m.type == WPT_MAT_GLASS || m.type == WPT_MAT_MIRROR);
    sec<COUNT>(lc, SEC_SCATTER_OTHER, m.type != WPT_MAT_LAMBERTIAN && m.type != WPT_MAT_GGX && m.type != WPT_MAT_GLASS && m.type != WPT_MAT_MIRROR && m.type != WPT_MAT_LIGHT_DIFFUSE);
    sec<COUNT>(lc, SEC_EMISSION);
    Prng prng = loadPrng(ps);
#ifdef WPT_MATERIAL_CACHE
    /* what scatter reads from the material's textures is kept for the evaluation towards the light (the wavefront shade
     * kernels: +2 %) */
    MatCache mc = matCacheEmpty();
    const Scatter sr = materialScatter<F>(sv, m, ray, h, prng, mc);
#else
    /* The single kernel lets every evaluation read its textures itself: the twelve kept values live across the light sampling,
     * where that kernel has no register to spare -- its scratch went from 272 to 336 bytes per lane and the frame's HBM writes
     * from 57 to 636 GB (Sponza-class, profiles/r03_pmc_sponza.txt of that build) for +0.3 % of speed. */
    MatCache mcScatter = matCacheEmpty(), mc = matCacheEmpty();
    const Scatter sr = materialScatter<F>(sv, m, ray, h, prng, mcScatter);
#endif
    section(1);
    const f4 att = ps.get4(SLOT_ATT);
    {
        f4 rad = mul(att, materialEmitted<F>(sv, m, h));
        accumulateRadiance(par, opl, (pathComponent == 0 ? 0.0f : h.a), rad, ps);
    }
    section(2);
    if (sr.type == SCATTER_NONE) {
        storePrng(ps, prng);
        return NEXT_NEW;
    }
    f4 nextAtt = mul(att, sr.att);
    if (sr.type == SCATTER_RANDOM) {
        if (sr.pdf > 0.0f)
            nextAtt = divs(nextAtt, sr.pdf);
        else
            nextAtt = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    sec<COUNT>(lc, SEC_LIGHT_SAMPLE, sr.type == SCATTER_RANDOM && sv.hotspotCount > 0);
    if (sr.type == SCATTER_RANDOM && sv.hotspotCount > 0) {
        /* light sampling with MIS (wurblpt.hpp:179-220).  The pdf of the scattered direction draws nothing from the
         * generator, so it can wait for the direction towards the light and be evaluated together with that one's. */
        uint32_t idx = (uint32_t)(in01(prng) * (float)sv.hotspotCount);
        idx = idx < sv.hotspotCount - 1 ? idx : sv.hotspotCount - 1;
        const wpt_hotspot& hs = sv.hotspots[idx];
        f3 directDir;
        uint32_t hotSpotPrim = hs.prim;
        if ((F & FEAT_SPHERES) && hs.kind == WPT_HOTSPOT_SPHERE) {
            const DirectionDraw draw = sphereDirection(sphereNow<F>(sv, ps, sv.spheres[hs.prim]), h.p, prng);
            directDir = draw.dir;
            prng = draw.prng;
            hotSpotPrim = PRIM_SPHERE | hs.prim;
        } else {
            /* HitableTriangle::direction (hitable_triangle.hpp:425-443) */
            const f3 bary = inTriangle(in01x2(prng));
            f3 p = add(add(scl(bary.x, ld3(hs.p0)), scl(bary.y, ld3(hs.p1))), scl(bary.z, ld3(hs.p2)));
            if (hs.transform)
                p = mat4mulPoint(hs.M, p);
            if ((F & FEAT_ANIM) && hs.animation >= 0)
                p = animatePoint(animationMatrix(sv, ps, hs.animation), p);
            directDir = normalize(sub(p, h.p));
        }
        section(4);
        float hotSpotsPdf, directPdf;
        sec<COUNT>(lc, SEC_PDF_SETUP);
        hotSpotsMeanPdfPair<F, COUNT>(sv, tri4, h.p, sr.dir, directDir, ps, lc, hotSpotsPdf, directPdf);
        nextAtt = sclr(nextAtt, powerHeuristicWeight(sr.pdf, hotSpotsPdf));
        section(5);
        sec<COUNT>(lc, SEC_EVAL_LAMBERT, directPdf > 0.0f && m.type == WPT_MAT_LAMBERTIAN);
        sec<COUNT>(lc, SEC_EVAL_GGX, directPdf > 0.0f && m.type == WPT_MAT_GGX);
        sec<COUNT>(lc, SEC_EVAL_OTHER, directPdf > 0.0f && m.type != WPT_MAT_LAMBERTIAN && m.type != WPT_MAT_GGX);
        if (directPdf > 0.0f) {
            float dpdf;
            f4 directAtt;
            materialEval<F>(sv, m, ray, h, directDir, directAtt, dpdf, mc);
            sec<COUNT>(lc, SEC_NEE_SETUP, dpdf > 0.0f);
            if (dpdf > 0.0f) {
                const f4 neeFactor = sclr(divs(mul(att, directAtt), directPdf), powerHeuristicWeight(directPdf, dpdf));
                Slot nee;
                nee.x = neeFactor.x; nee.y = neeFactor.y; nee.z = neeFactor.z;
                nee.w = hotSpotPrim;
                ps.set(SLOT_NEE, nee);
                if (MERGED) {
                    ps.set3(SLOT_NEXTATT, opl);
                    ps.set3(SLOT_SRDIR, mk3(ray.ri.x, ray.ri.y, ray.ri.z));
                    ps.neeO = h.p;
                    ps.neeD = directDir;
                    ps.neeKind = RAY_NEE_LIGHT;
                } else {
                    ps.set4(SLOT_NEXTATT, nextAtt);
                    ps.set3(SLOT_SRDIR, sr.dir);
                    storePrng(ps, prng);
                    ps.o = h.p;
                    ps.d = directDir;
                    ps.rayKind = RAY_NEE_LIGHT;
                    section(6);
                    return NEXT_TRACE;
                }
            }
        }
    } else if ((F & FEAT_ENVMAP) && sr.type == SCATTER_RANDOM && haveEnv && sv.envN > 0) {
        /* environment sampling with MIS (wurblpt.hpp:221-252) */
        const float lightsP = envP(sv, sr.dir);
        nextAtt = sclr(nextAtt, powerHeuristicWeight(sr.pdf, lightsP));
        const f3 lightDir = envD(sv, prng);
        const float directPdf = envP(sv, lightDir);
        float dpdf;
        f4 directAtt;
        materialEval<F>(sv, m, ray, h, lightDir, directAtt, dpdf, mc);
        if (dpdf > 0.0f) {
            const f4 neeFactor = sclr(divs(mul(att, directAtt), directPdf), powerHeuristicWeight(directPdf, dpdf));
            ps.set3(SLOT_NEE, mk3(neeFactor.x, neeFactor.y, neeFactor.z));
            if (MERGED) {
                ps.neeO = h.p;
                ps.neeD = lightDir;
                ps.neeKind = RAY_NEE_ENV; /* (its end needs no path lengths: the environment is infinitely far, wurblpt.hpp:244-248) */
            } else {
                ps.set4(SLOT_NEXTATT, nextAtt);
                ps.set3(SLOT_SRDIR, sr.dir);
                storePrng(ps, prng);
                ps.o = h.p;
                ps.d = lightDir;
                ps.rayKind = RAY_NEE_ENV;
                section(7);
                return NEXT_TRACE;
            }
        }
    }
    /* No next-event ray.  The scattered ray's refractive index: every ScatterRandom record
     * carries the incoming ray's index unchanged (material_lambertian.hpp:83, material_ggx.hpp:224,
     * material_modphong.hpp:307), so the slot already holds the value to continue with, also while a
     * next-event ray is in flight; only explicit scattering (glass, mirror, transparent ModPhong) sets it. */
    sec<COUNT>(lc, SEC_ADVANCE);
    ps.o = h.p;
    if (sr.type == SCATTER_EXPLICIT)
        ps.set4(SLOT_RI, sr.ri);
    const int next = advancePath(par, ps, nextAtt, sr.dir, pathComponent, prng);
    storePrng(ps, prng);
    section(7);
    return next;
}

/* the next-event ray's result (wurblpt.hpp:208-218: only the CHOSEN hot spot as nearest hit
 * counts; :240-250: the environment counts if nothing was hit), then the path continues */
template<uint32_t F, class Tri4, class PS>
WPT_D int blockNeeEnd(const SceneView& sv, const wpt_params& par, Tri4 tri4, PS& ps, const Candidate& best)
{
    const Slot oplSlot = ps.get(SLOT_OPL);
    if (ps.rayKind == RAY_NEE_LIGHT) {
        const Slot nee = ps.get(SLOT_NEE);
        if (best.prim == nee.w) {
            Hit lh = finishHit<F>(sv, best, ps.o, ps.d, ps.time, tri4);
            const wpt_material& lm = resolveMaterial<F>(sv, lh.material, lh);
            f4 rad = mul(mk4(nee.x, nee.y, nee.z, 0.0f), materialEmitted<F>(sv, lm, lh));
            const f4 ri = ps.get4(SLOT_RI);
            f3 oplLight = add(mk3(oplSlot.x, oplSlot.y, oplSlot.z), scl(lh.a, mk3(ri.x, ri.y, ri.z)));
            accumulateRadiance(par, oplLight, lh.a, rad, ps);
        }
    } else if (F & FEAT_ENVMAP) {
        if (best.prim == NO_HIT) {
            const Slot nee = ps.get(SLOT_NEE);
            f4 rad = mul(mk4(nee.x, nee.y, nee.z, 0.0f), envL(sv, ps.d));
            accumulateRadiance(par, mk3(k_maxval, k_maxval, k_maxval), k_maxval, rad, ps);
        }
    }
    const f4 nextAtt = ps.get4(SLOT_NEXTATT);
    const Slot sd = ps.get(SLOT_SRDIR);
    /* the generator is fetched (and stored again) only when the roulette will draw: advancePath's own test */
    const bool roulette = max4(nextAtt) < par.rr_threshold && oplSlot.w >= 5;
    Prng prng;
    prng.s0 = prng.s1 = prng.s2 = prng.s3 = 0;
    if (roulette)
        prng = loadPrng(ps);
    const int next = advancePath(par, ps, nextAtt, mk3(sd.x, sd.y, sd.z), oplSlot.w, prng);
    if (roulette)
        storePrng(ps, prng);
    return next;
}

/* MERGED: the answer of the light ray that travelled beside the continuation (ps.neeKind, ps.neeO, ps.neeD; `neeBest` its result):
 * the first half of blockNeeEnd (wurblpt.hpp:208-218,240-250) with this hit's optical path length and refractive index from
 * the slots blockShade<MERGED> left them in */
template<uint32_t F, class Tri4, class PS>
WPT_D void blockNeeResult(const SceneView& sv, const wpt_params& par, Tri4 tri4, PS& ps, const Candidate& neeBest)
{
    const Slot nee = ps.get(SLOT_NEE);
    if (ps.neeKind == RAY_NEE_LIGHT) {
        if (neeBest.prim == nee.w) {
            Hit lh = finishHit<F>(sv, neeBest, ps.neeO, ps.neeD, ps.time, tri4);
            const wpt_material& lm = resolveMaterial<F>(sv, lh.material, lh);
            f4 rad = mul(mk4(nee.x, nee.y, nee.z, 0.0f), materialEmitted<F>(sv, lm, lh));
            const Slot opl = ps.get(SLOT_NEXTATT), ri = ps.get(SLOT_SRDIR);
            f3 oplLight = add(mk3(opl.x, opl.y, opl.z), scl(lh.a, mk3(ri.x, ri.y, ri.z)));
            accumulateRadiance(par, oplLight, lh.a, rad, ps);
        }
    } else if (F & FEAT_ENVMAP) {
        if (neeBest.prim == NO_HIT) {
            f4 rad = mul(mk4(nee.x, nee.y, nee.z, 0.0f), envL(sv, ps.neeD));
            accumulateRadiance(par, mk3(k_maxval, k_maxval, k_maxval), k_maxval, rad, ps);
        }
    }
}

} /* namespace wptk */

#endif
