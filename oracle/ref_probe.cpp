/*
 * ref_probe.cpp -- golden-vector generator that runs THE REFERENCE'S OWN CODE.
 *
 * TEST INFRASTRUCTURE.  Compiled only in the build container, only when /root/reference
 * exists, by oracle/Makefile, against the reference headers where they lie
 * (-I/root/reference/libwurblpt); the binary goes to oracle/_ref/ (git-ignored) and its
 * output to tests/golden/ref_golden.json (committed: data only, no reference source).
 *
 * Only reference headers that compile from the reference tree alone are used:
 * gvm, prng, sampler, tangentspace, fresnel, ray, aabb, hitable, bvh, transformation,
 * animation, optics, camera, geometryproc.  Everything that (transitively) includes
 * <tgd/array.hpp> -- texture, material*, mesh, hitable_triangle, envmap, sensor, scene,
 * wurblpt.hpp -- needs the external libtgd, which is not in this image, and is therefore NOT
 * built (no stand-in headers are written for it).
 *
 * Floats are written as their 32-bit patterns (hex) so that comparisons are bit-exact.
 */
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "prng.hpp"
#include "sampler.hpp"
#include "tangentspace.hpp"
#include "fresnel.hpp"
#include "aabb.hpp"
#include "hitable.hpp"
#include "bvh.hpp"
#include "transformation.hpp"
#include "optics.hpp"
#include "camera.hpp"
#include "geometryproc.hpp"
#include "hitable_sphere.hpp"
#include "color.hpp"
#include "animation_keyframes.hpp"
#define POWITACQ_IMPLEMENTATION
#include "powitacq_rgb.h"
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

using namespace WurblPT;

static FILE* out;
static bool firstKey = true;

static uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static void key(const char* name)
{
    fprintf(out, "%s\n\"%s\": ", firstKey ? "" : ",", name);
    firstKey = false;
}

static void floats(const char* name, const std::vector<float>& v)
{
    key(name);
    fprintf(out, "[");
    for (size_t i = 0; i < v.size(); i++)
        fprintf(out, "%s\"%08x\"", i ? "," : "", bits(v[i]));
    fprintf(out, "]");
}

static void ints(const char* name, const std::vector<long long>& v)
{
    key(name);
    fprintf(out, "[");
    for (size_t i = 0; i < v.size(); i++)
        fprintf(out, "%s%lld", i ? "," : "", v[i]);
    fprintf(out, "]");
}

static void push3(std::vector<float>& v, const vec3& a)
{
    v.push_back(a.x());
    v.push_back(a.y());
    v.push_back(a.z());
}

/* FNV-1a over a byte range, to pin large arrays with a few bytes */
static uint64_t fnv1a(const void* data, size_t n)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

/* A hitable for exercising the reference's BVH build and BVH::hit: a box with a fixed
 * "hit distance".  hit() reports a hit at that distance iff it lies in [amin, amax], and logs
 * the visit, so the log is the reference's leaf visiting order with its amax shrinking. */
class ProbeHitable final : public Hitable
{
public:
    AABB box;
    float a;
    static std::vector<long long>* log;
    static const ProbeHitable* base;

    virtual AABB aabb(AnimationCache&, AnimationCache&) const override { return box; }
    virtual HitRecord hit(const Ray&, const RayIntersectionHelper&, float amin, float amax, float, AnimationCache&, Prng&) const override
    {
        if (log)
            log->push_back(this - base);
        if (a >= amin && a <= amax) {
            HitRecord hr(a);
            hr.hitable = this;
            return hr;
        }
        return HitRecord();
    }
};
std::vector<long long>* ProbeHitable::log = nullptr;
const ProbeHitable* ProbeHitable::base = nullptr;

/* BVH exposes its nodes only privately; the flattened order is observed instead through
 * BVH::hit's visiting order with a ray-independent all-pass query (see below), and the
 * boxes through aabb().  For the node array itself we rebuild the same tree with the
 * reference's BVHNode::buildBVH and walk it. */
struct FlatNode {
    float lo[3], hi[3];
    long long link;
    long long kind;
};

class BVHNodeWalker
{
public:
    /* BVHNode's members are private with `friend class BVH`; measure() is public and
     * build is deterministic, so we reproduce the flatten order by a second walk that uses
     * only public behaviour: a BVH over the same hitables, queried with rays that pass every
     * box, visits the leaves in depth-first flatten order. */
};

static std::vector<float> randomBoxes(unsigned int n, unsigned int seed, bool degenerate)
{
    std::mt19937 rng(seed);
    auto u01 = [&rng]() { return float(rng() >> 8) * (1.0f / 16777216.0f); };
    std::vector<float> b(6 * n);
    for (unsigned int i = 0; i < n; i++) {
        float c[3], e[3];
        for (int k = 0; k < 3; k++) {
            c[k] = u01() * 2.0f - 1.0f;
            e[k] = degenerate && (rng() & 3) == 0 ? 0.0f : 0.01f + 0.1f * u01();
            if (degenerate && (rng() & 7) == 0)
                c[k] = float(int(c[k] * 4.0f)) * 0.25f; /* many equal centres -> sort ties */
        }
        for (int k = 0; k < 3; k++) {
            b[6 * i + k] = c[k] - e[k];
            b[6 * i + 3 + k] = c[k] + e[k];
        }
    }
    return b;
}

/* what the vendored tinyobjloader makes of an OBJ file, configured as the importer configures it (import.hpp:214-218) */
static void dumpObj(const std::string& objFile)
{
    {
        tinyobj::ObjReaderConfig conf;
        conf.triangulate = true;
        conf.vertex_color = false;
        tinyobj::ObjReader reader;
        reader.ParseFromFile(objFile, conf);
        std::vector<long long> valid(1, reader.Valid() ? 1 : 0);
        ints("obj_valid", valid);
        const tinyobj::attrib_t& attrib = reader.GetAttrib();
        floats("obj_vertices", std::vector<float>(attrib.vertices.begin(), attrib.vertices.end()));
        floats("obj_normals", std::vector<float>(attrib.normals.begin(), attrib.normals.end()));
        floats("obj_texcoords", std::vector<float>(attrib.texcoords.begin(), attrib.texcoords.end()));
        auto chars = [](std::vector<long long>& dst, const std::string& str) {
            dst.push_back((long long)str.size());
            for (char ch : str)
                dst.push_back((unsigned char)ch);
        };
        std::vector<long long> shapeNames, shapeSizes, indices, materialIds;
        for (const tinyobj::shape_t& sh : reader.GetShapes()) {
            chars(shapeNames, sh.name);
            shapeSizes.push_back((long long)sh.mesh.indices.size());
            for (const tinyobj::index_t& ix : sh.mesh.indices) {
                indices.push_back(ix.vertex_index);
                indices.push_back(ix.normal_index);
                indices.push_back(ix.texcoord_index);
            }
            for (int id : sh.mesh.material_ids)
                materialIds.push_back(id);
        }
        ints("obj_shape_names", shapeNames);       /* per shape: length, bytes */
        ints("obj_shape_index_counts", shapeSizes);
        ints("obj_indices", indices);              /* vertex, normal, texcoord per corner */
        ints("obj_material_ids", materialIds);     /* per triangle */
        std::vector<long long> matStrings;
        std::vector<float> matFloats;
        for (const tinyobj::material_t& M : reader.GetMaterials()) {
            chars(matStrings, M.name);
            const std::string* names[7] = { &M.diffuse_texname, &M.specular_texname, &M.specular_highlight_texname, &M.bump_texname,
                &M.alpha_texname, &M.emissive_texname, &M.normal_texname };
            const tinyobj::texture_option_t* opts[7] = { &M.diffuse_texopt, &M.specular_texopt, &M.specular_highlight_texopt, &M.bump_texopt,
                &M.alpha_texopt, &M.emissive_texopt, &M.normal_texopt };
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
                for (int k = 0; k < 3; k++) matFloats.push_back(opts[t]->origin_offset[k]);
                matFloats.push_back(opts[t]->bump_multiplier);
            }
        }
        ints("obj_material_strings", matStrings);  /* per material: name, then 7 texture names (diffuse specular shininess bump alpha emissive normal) */
        floats("obj_material_floats", matFloats);  /* per material: Kd Ks Ke Tf (3 each) Ns d Ni, then per texture scale(3) offset(3) bm */
    }

}

int main(int argc, char* argv[])
{
    if (argc == 4 && std::string(argv[1]) == "--obj") { /* ref_probe --obj file.obj out.json: only the obj_* entries, for any file */
        out = fopen(argv[3], "w");
        if (!out)
            return 1;
        fprintf(out, "{");
        key("generator");
        fprintf(out, "\"oracle/ref_probe.cpp --obj\"");
        dumpObj(argv[2]);
        fprintf(out, "\n}\n");
        fclose(out);
        return 0;
    }
    out = argc > 1 ? fopen(argv[1], "w") : stdout;
    if (!out)
        return 1;
    fprintf(out, "{");
    key("generator");
    fprintf(out, "\"oracle/ref_probe.cpp over /root/reference/libwurblpt headers, g++ %d.%d.%d -O2 -ffp-contract=off\"", __GNUC__, __GNUC_MINOR__, __GNUC_PATCHLEVEL__);

    /* ---- Prng (prng.hpp:47-101) ---- */
    {
        const unsigned int pixels[5] = { 0, 1, 7, 65535, 1048575 };
        std::vector<long long> px;
        std::vector<float> v, v2;
        for (unsigned int p : pixels) {
            px.push_back(p);
            Prng prng(p);
            for (int i = 0; i < 64; i++)
                v.push_back(prng.in01());
            Prng prng2(p);
            for (int i = 0; i < 8; i++) {
                vec2 xy = prng2.in01x2();
                v2.push_back(xy.x());
                v2.push_back(xy.y());
            }
        }
        ints("prng_pixels", px);
        floats("prng_in01", v);
        floats("prng_in01x2", v2);
    }

    /* ---- Sampler (sampler.hpp:39-123), 32x32 grid plus the special points ---- */
    {
        std::vector<float> u, disk, tri, cosd;
        for (int j = 0; j < 32; j++)
            for (int i = 0; i < 32; i++) {
                u.push_back((i + 0.37f) / 32.0f);
                u.push_back((j + 0.61f) / 32.0f);
            }
        const float special[][2] = { { 0.5f, 0.5f }, { 0.0f, 0.0f }, { 0.5f, 0.25f }, { 0.25f, 0.5f }, { 0.99999994f, 0.99999994f }, { 0.0f, 0.5f }, { 0.5f, 0.0f } };
        for (auto& s : special) {
            u.push_back(s[0]);
            u.push_back(s[1]);
        }
        for (size_t i = 0; i < u.size() / 2; i++) {
            vec2 uu(u[2 * i], u[2 * i + 1]);
            vec2 d = Sampler::inUnitDisk(uu);
            disk.push_back(d.x());
            disk.push_back(d.y());
            push3(tri, Sampler::inTriangle(uu));
            push3(cosd, Sampler::cosineDirection(uu));
        }
        floats("sampler_u", u);
        floats("sampler_inUnitDisk", disk);
        floats("sampler_inTriangle", tri);
        floats("sampler_cosineDirection", cosd);
    }

    std::mt19937 rng(12345);
    auto u01 = [&rng]() { return float(rng() >> 8) * (1.0f / 16777216.0f); };
    auto randDir = [&]() {
        for (;;) {
            vec3 d(u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f);
            float l = dot(d, d);
            if (l > 0.01f && l <= 1.0f)
                return normalize(d);
        }
    };

    /* ---- TangentSpace (tangentspace.hpp:57-136) ---- */
    {
        std::vector<float> nrm, vec, res;
        for (int i = 0; i < 256; i++) {
            vec3 n = randDir();
            if (i == 0) n = vec3(0.0f, 0.0f, 1.0f);
            if (i == 1) n = vec3(0.0f, 0.0f, -1.0f);
            if (i == 2) n = vec3(1.0f, 0.0f, 0.0f);
            if (i == 3) n = vec3(0.0f, -1.0f, 0.0f);
            vec3 v = randDir();
            TangentSpace ts(n);
            push3(nrm, n);
            push3(vec, v);
            push3(res, ts.tangent);
            push3(res, ts.bitangent);
            push3(res, ts.toWorldSpace(v));
            push3(res, ts.toTangentSpace(v));
        }
        floats("tangentspace_normals", nrm);
        floats("tangentspace_vecs", vec);
        floats("tangentspace_out", res);
    }

    /* ---- Fresnel (fresnel.hpp:48-72), reflect / refract (gvm.hpp:1213-1223) ---- */
    {
        std::vector<float> in, res;
        for (int i = 0; i < 256; i++) {
            float a = u01(), b = u01(), c = 1.0f + u01(), d = 1.0f + u01();
            in.insert(in.end(), { a, b, c, d });
            res.push_back(fresnelUnpolarized(a, b, c, d));
            vec4 s = fresnelSchlick(vec4(a, b, c, d), b);
            res.insert(res.end(), { s.x(), s.y(), s.z(), s.w() });
        }
        floats("fresnel_in", in);
        floats("fresnel_out", res);
        std::vector<float> in7, res6;
        for (int i = 0; i < 256; i++) {
            vec3 dI = randDir(), n = randDir();
            float eta = (i & 1) ? 1.5f : 1.0f / 1.5f;
            if (i % 7 == 0) eta = 1.0f;
            push3(in7, dI);
            push3(in7, n);
            in7.push_back(eta);
            push3(res6, reflect(dI, n));
            push3(res6, refract(dI, n, eta));
        }
        floats("reflect_refract_in", in7);
        floats("reflect_refract_out", res6);
    }

    /* ---- RayIntersectionHelper (hitable.hpp:66-113), AABB::mayHit (aabb.hpp:70-86) ---- */
    {
        std::vector<float> rays, helper;
        for (int i = 0; i < 512; i++) {
            vec3 o(u01() * 4.0f - 2.0f, u01() * 4.0f - 2.0f, u01() * 4.0f - 2.0f);
            vec3 d = randDir();
            if (i < 6) { d = vec3(0.0f); d[i % 3] = (i < 3 ? 1.0f : -1.0f); } /* axis aligned: inf in invDirection */
            if (i == 6) d = normalize(vec3(1.0f, 1.0f, 0.0f));
            if (i == 7) d = normalize(vec3(0.0f, -1.0f, 1.0f));
            Ray r(o, d, 0.0f, vec4(1.0f));
            RayIntersectionHelper h(r);
            push3(rays, o);
            push3(rays, d);
            push3(helper, h.invDirection);
            helper.insert(helper.end(), { float(h.k.x()), float(h.k.y()), float(h.k.z()) });
            push3(helper, h.S);
        }
        floats("rayhelper_rays", rays);
        floats("rayhelper_out", helper);
        std::vector<float> boxes, brays;
        std::vector<long long> hit;
        for (int i = 0; i < 2048; i++) {
            vec3 c(u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f);
            vec3 e(u01() * 0.5f, u01() * 0.5f, u01() * 0.5f);
            if (i % 16 == 0) e[i / 16 % 3] = 0.0f; /* flat boxes */
            AABB box(c - e, c + e);
            vec3 o(u01() * 4.0f - 2.0f, u01() * 4.0f - 2.0f, u01() * 4.0f - 2.0f);
            vec3 d = randDir();
            if (i % 8 == 1) { d = vec3(0.0f); d[i / 8 % 3] = (i & 64) ? 1.0f : -1.0f; }
            if (i % 32 == 1) o[(i / 8 % 3 + 1) % 3] = box.lo[(i / 8 % 3 + 1) % 3]; /* origin on a slab plane with zero direction: 0 * inf = NaN */
            if (i % 8 == 2) d = normalize(c - o); /* aimed at the box */
            float amin = (i % 5 == 0) ? 0.0f : 1e-5f;
            float amax = (i % 3 == 0) ? 1.5f : maxval;
            Ray r(o, d, 0.0f, vec4(1.0f));
            RayIntersectionHelper h(r);
            push3(boxes, box.lo);
            push3(boxes, box.hi);
            push3(brays, o);
            push3(brays, d);
            brays.push_back(amin);
            brays.push_back(amax);
            hit.push_back(box.mayHit(r, amin, amax, h.invDirection) ? 1 : 0);
        }
        floats("aabb_boxes", boxes);
        floats("aabb_rays", brays);
        ints("aabb_mayhit", hit);
    }

    /* ---- BVH build + traversal (bvh.hpp:93-311) over probe hitables ---- */
    {
        struct Case { unsigned int n, seed; bool degenerate; };
        const Case cases[] = { { 1, 1, false }, { 2, 2, false }, { 3, 3, false }, { 7, 4, true }, { 36, 5, false }, { 200, 6, true }, { 1000, 7, false }, { 40000, 8, true } };
        int ci = 0;
        for (const Case& cs : cases) {
            std::vector<float> b = randomBoxes(cs.n, cs.seed, cs.degenerate);
            std::vector<ProbeHitable> hitables(cs.n);
            std::vector<const Hitable*> ptrs(cs.n);
            std::mt19937 arng(cs.seed * 977u);
            for (unsigned int i = 0; i < cs.n; i++) {
                hitables[i].box = AABB(vec3(b.data() + 6 * i), vec3(b.data() + 6 * i + 3));
                hitables[i].a = 0.5f + 4.0f * float(arng() >> 8) * (1.0f / 16777216.0f);
                ptrs[i] = &hitables[i];
            }
            ProbeHitable::base = hitables.data();
            std::vector<const Animation*> noAnimations;
            AnimationCache c0(noAnimations, 0.0f), c1(noAnimations, 0.0f);
            BVH bvh;
            bvh.build(ptrs, c0, c1);
            Prng prng(0);
            /* (1) flatten order of the leaves: a query that passes every box and never shrinks
             * amax.  Ray from far away along +x with invDirection made irrelevant by amin=-inf.. is not
             * possible; instead use the property that mayHit() is true for every box when the
             * ray origin is inside... not general either.  So: log visits for MANY rays and
             * keep the full logs; together with the leaf boxes this pins order and pruning. */
            std::vector<float> qrays;
            std::vector<long long> visitLog, finalHit;
            std::vector<float> finalA;
            int nq = cs.n >= 40000 ? 64 : 128;
            for (int q = 0; q < nq; q++) {
                vec3 o(u01() * 3.0f - 1.5f, u01() * 3.0f - 1.5f, u01() * 3.0f - 1.5f);
                vec3 d = randDir();
                if (q % 16 == 3) { d = vec3(0.0f); d[q / 16 % 3] = 1.0f; }
                float amin = 1e-5f;
                float amax = (q % 4 == 0) ? 2.0f : maxval;
                Ray r(o, d, 0.0f, vec4(1.0f));
                std::vector<long long> log;
                ProbeHitable::log = &log;
                HitRecord hr = bvh.hit(r, RayIntersectionHelper(r), amin, amax, amin, c0, prng);
                ProbeHitable::log = nullptr;
                push3(qrays, o);
                push3(qrays, d);
                qrays.push_back(amin);
                qrays.push_back(amax);
                visitLog.push_back(log.size());
                visitLog.insert(visitLog.end(), log.begin(), log.end());
                finalHit.push_back(hr.haveHit ? static_cast<const ProbeHitable*>(hr.hitable) - hitables.data() : -1);
                finalA.push_back(hr.haveHit ? hr.a : 0.0f);
            }
            char name[64];
            std::vector<long long> meta = { (long long)cs.n, (long long)cs.seed, cs.degenerate ? 1 : 0 };
            snprintf(name, sizeof(name), "bvh%d_meta", ci);
            ints(name, meta);
            snprintf(name, sizeof(name), "bvh%d_rays", ci);
            floats(name, qrays);
            if (cs.n <= 1000) {
                snprintf(name, sizeof(name), "bvh%d_boxes", ci);
                floats(name, b);
                std::vector<float> as;
                for (unsigned int i = 0; i < cs.n; i++)
                    as.push_back(hitables[i].a);
                snprintf(name, sizeof(name), "bvh%d_leaf_a", ci);
                floats(name, as);
                snprintf(name, sizeof(name), "bvh%d_visitlog", ci);
                ints(name, visitLog);
            } else {
                /* too large to commit: the generator is seeded (randomBoxes is restated in the
                 * test), so commit only a hash of the visit logs */
                snprintf(name, sizeof(name), "bvh%d_visitlog_fnv", ci);
                std::vector<long long> h = { (long long)(fnv1a(visitLog.data(), visitLog.size() * sizeof(long long)) >> 1) };
                ints(name, h);
            }
            snprintf(name, sizeof(name), "bvh%d_final", ci);
            ints(name, finalHit);
            snprintf(name, sizeof(name), "bvh%d_final_a", ci);
            floats(name, finalA);
            /* root box */
            AABB root = bvh.aabb(c0, c1);
            std::vector<float> rb;
            push3(rb, root.lo);
            push3(rb, root.hi);
            snprintf(name, sizeof(name), "bvh%d_rootbox", ci);
            floats(name, rb);
            ci++;
        }
    }

    /* ---- Transformation (transformation.hpp:80-137), Projection (optics.hpp:49-55) ---- */
    {
        std::vector<float> in, res;
        const float cases[][9] = {
            { 0.0f, 1.0f, 3.2f, 0.0f, 1.0f, -1.0f, 0.0f, 1.0f, 0.0f },  /* Cornell camera */
            { 0.0f, 1.7f, 0.0f, 0.0f, 1.7f, -1.0f, 0.0f, 1.0f, 0.0f },  /* Sponza camera */
            { 0.3f, 0.4f, 3.5f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f },
            { 13.0f, 2.0f, 3.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f },
            { 1.0f, 2.0f, 3.0f, -2.0f, 0.5f, 1.0f, 0.1f, 0.9f, 0.2f },
            { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 0.0f },  /* looking along +z: 180 degree case */
        };
        for (auto& c : cases) {
            in.insert(in.end(), c, c + 9);
            Transformation T = Transformation::fromLookAt(vec3(c[0], c[1], c[2]), vec3(c[3], c[4], c[5]), normalize(vec3(c[6], c[7], c[8])));
            push3(res, T.translation);
            res.insert(res.end(), { T.rotation.x, T.rotation.y, T.rotation.z, T.rotation.w });
            push3(res, T.scaling);
            mat4 M = T.toMat4();
            for (int k = 0; k < 16; k++)
                res.push_back(M.values[k]);
            mat3 N = T.toNormalMatrix();
            for (int k = 0; k < 9; k++)
                res.push_back(N.values[k]);
        }
        floats("lookat_in", in);
        floats("lookat_out", res);
        std::vector<float> tin, tout;
        for (int i = 0; i < 64; i++) {
            vec3 t(u01() * 4.0f - 2.0f, u01() * 4.0f - 2.0f, u01() * 4.0f - 2.0f);
            float angle = u01() * 6.0f - 3.0f;
            vec3 axis = randDir() * (0.5f + u01());
            vec3 s(0.1f + u01() * 2.0f, 0.1f + u01() * 2.0f, 0.1f + u01() * 2.0f);
            vec3 v(u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f, u01() * 2.0f - 1.0f);
            if (i == 0) { angle = radians(90.0f); axis = vec3(0.0f, 1.0f, 0.0f); s = vec3(0.01f); t = vec3(0.0f); } /* Sponza import transformation */
            push3(tin, t);
            tin.push_back(angle);
            push3(tin, axis);
            push3(tin, s);
            push3(tin, v);
            Transformation T(t, toQuat(angle, axis), s);
            mat4 M = T.toMat4();
            for (int k = 0; k < 16; k++)
                tout.push_back(M.values[k]);
            mat3 N = T.toNormalMatrix();
            for (int k = 0; k < 9; k++)
                tout.push_back(N.values[k]);
            push3(tout, T * v);
        }
        floats("transformation_in", tin);
        floats("transformation_out", tout);
        std::vector<float> pin, pout;
        const float pc[][2] = { { radians(50.0f), 1.0f }, { radians(70.0f), 1920.0f / 1080.0f }, { radians(60.0f), 4.0f / 3.0f }, { radians(35.0f), 2.0f } };
        for (auto& c : pc) {
            Projection P(c[0], c[1]);
            pin.insert(pin.end(), { c[0], c[1] });
            pout.insert(pout.end(), { P.l, P.r, P.b, P.t });
        }
        floats("projection_in", pin);
        floats("projection_out", pout);
    }

    /* ---- Camera::getRay (camera.hpp:123-185): pinhole and thin lens ---- */
    {
        for (int variant = 0; variant < 2; variant++) {
            float aperture = variant == 0 ? 0.0f : 0.1f;
            Optics optics(Projection(radians(50.0f), 1.0f), LensDistortion(), LensDepthOfField(aperture, 3.2f));
            Camera camera(optics, Transformation::fromLookAt(vec3(0.0f, 1.0f, 3.2f), vec3(0.0f, 1.0f, -1.0f), vec3(0.0f, 1.0f, 0.0f)));
            Camera::RayHelper rh = camera.getRayHelper(0.0f, 64, 64);
            Prng prng(0);
            std::vector<float> pq, res;
            for (int j = 0; j < 16; j++)
                for (int i = 0; i < 16; i++) {
                    float p = (i + 0.3f) / 16.0f, q = (j + 0.7f) / 16.0f;
                    Ray r = camera.getRay(p, q, 0.0f, 0.0f, rh, prng);
                    pq.push_back(p);
                    pq.push_back(q);
                    push3(res, r.origin);
                    push3(res, r.direction);
                }
            floats(variant == 0 ? "camera_pinhole_pq" : "camera_lens_pq", pq);
            floats(variant == 0 ? "camera_pinhole_rays" : "camera_lens_rays", res);
            if (variant == 1) {
                /* the form getGroundTruth uses (wurblpt.hpp:660-662): no randomness, so no depth of field offset */
                std::vector<float> plain;
                for (size_t k = 0; k < pq.size(); k += 2) {
                    Ray r = camera.getRay(pq[k], pq[k + 1], 0.0f, 0.0f, rh, prng, false);
                    push3(plain, r.origin);
                    push3(plain, r.direction);
                }
                floats("camera_lens_rays_without_randomness", plain);
            }
            std::vector<float> camdesc = { optics.projection.l, optics.projection.r, optics.projection.b, optics.projection.t,
                camera.transformation.translation.x(), camera.transformation.translation.y(), camera.transformation.translation.z(),
                camera.transformation.rotation.x, camera.transformation.rotation.y, camera.transformation.rotation.z, camera.transformation.rotation.w,
                camera.transformation.scaling.x(), camera.transformation.scaling.y(), camera.transformation.scaling.z(),
                optics.depthOfField.lensRadius, optics.depthOfField.focusDist };
            floats(variant == 0 ? "camera_pinhole_desc" : "camera_lens_desc", camdesc);
        }
    }

    /* ---- LensDistortion (optics.hpp:112-309): undistort, distort of the result, and Camera::getRay
     * through it, for the three models on an off-centre projection ---- */
    {
        const unsigned int W = 640, H = 480;
        Projection proj(W, H, vec2(311.5f, 250.25f), vec2(520.0f, 515.0f));
        LensDistortion models[3] = { LensDistortion(-0.21f, 0.07f, 0.0012f, -0.0009f), LensDistortion(-0.18f, 0.05f, -0.01f),
            LensDistortion(-0.25f, 0.09f, -0.015f, 0.0011f, -0.0007f) };
        const char* names[3] = { "radial_and_planar", "radial_only", "opencv" };
        for (int m = 0; m < 3; m++) {
            LensDistortion::Helper helper = models[m].getHelper(proj, W, H);
            Optics optics(proj, models[m], LensDepthOfField());
            Camera camera(optics, Transformation::fromLookAt(vec3(0.5f, 1.0f, 3.0f), vec3(0.0f, 0.8f, -1.0f), vec3(0.0f, 1.0f, 0.0f)));
            Camera::RayHelper rh = camera.getRayHelper(0.0f, W, H);
            Prng prng(0);
            std::vector<float> pq, und, rays;
            for (int j = 0; j < 12; j++)
                for (int i = 0; i < 16; i++) {
                    float p = (i + 0.41f) / 16.0f, q = (j + 0.67f) / 12.0f;
                    pq.push_back(p);
                    pq.push_back(q);
                    float up = p, uq = q;
                    models[m].undistort(up, uq, helper);
                    und.push_back(up);
                    und.push_back(uq);
                    models[m].distort(up, uq, helper);
                    und.push_back(up);
                    und.push_back(uq);
                    Ray r = camera.getRay(p, q, 0.0f, 0.0f, rh, prng);
                    push3(rays, r.origin);
                    push3(rays, r.direction);
                }
            std::vector<float> desc = { proj.l, proj.r, proj.b, proj.t,
                camera.transformation.translation.x(), camera.transformation.translation.y(), camera.transformation.translation.z(),
                camera.transformation.rotation.x, camera.transformation.rotation.y, camera.transformation.rotation.z, camera.transformation.rotation.w,
                camera.transformation.scaling.x(), camera.transformation.scaling.y(), camera.transformation.scaling.z(),
                float(int(models[m].type)), models[m].k1, models[m].k2, models[m].k3, models[m].p1, models[m].p2,
                models[m].b1, models[m].b2, models[m].b3, models[m].b4,
                helper.center.x(), helper.center.y(), helper.focalLength.x(), helper.focalLength.y(),
                helper.inverseFocalLength.x(), helper.inverseFocalLength.y(), float(W), float(H) };
            floats((std::string("lens_") + names[m] + "_desc").c_str(), desc);
            floats((std::string("lens_") + names[m] + "_pq").c_str(), pq);
            floats((std::string("lens_") + names[m] + "_undistort_distort").c_str(), und);
            floats((std::string("lens_") + names[m] + "_rays").c_str(), rays);
            /* what getGroundTruth does with a hit position (wurblpt.hpp:679,706-717): world space -> camera space with
             * the inverse camera transformation, camera space -> (distorted) image space */
            {
                Camera::ImageSpaceHelper ish = camera.getImageSpaceHelper(W, H);
                Transformation inv = inverse(camera.at(0.0f));
                Prng pointPrng(17 + m);
                std::vector<float> ws, cs, ic;
                for (int k = 0; k < 256; k++) {
                    /* points on rays through the image, 0.5 .. 8.5 units away */
                    Ray r = camera.getRay(pointPrng.in01(), pointPrng.in01(), 0.0f, 0.0f, rh, prng, false);
                    vec3 wsPos = r.origin + (0.5f + 8.0f * pointPrng.in01()) * r.direction;
                    vec3 csPos = inv.rotation * wsPos + inv.translation;
                    vec2 imageCoord = camera.cameraSpaceToImageSpace(csPos, ish);
                    push3(ws, wsPos);
                    push3(cs, csPos);
                    ic.push_back(imageCoord.x());
                    ic.push_back(imageCoord.y());
                }
                floats((std::string("lens_") + names[m] + "_world_points").c_str(), ws);
                floats((std::string("lens_") + names[m] + "_camera_points").c_str(), cs);
                floats((std::string("lens_") + names[m] + "_image_coords").c_str(), ic);
            }
        }
    }

    /* ---- Camera::getRay in the surround and stereoscopic modes (camera.hpp:128-170) ---- */
    {
        const struct { Camera::SurroundMode mode; float stereo; const char* name; } modes[4] = {
            { Camera::Surround_180, 0.0f, "surround180" }, { Camera::Surround_360, 0.0f, "surround360" },
            { Camera::Surround_360, 0.065f, "surround360_stereo" }, { Camera::Surround_Off, 0.065f, "stereo" } };
        for (const auto& md : modes) {
            Optics optics(Projection(radians(50.0f), 1.5f), LensDistortion(), LensDepthOfField());
            Camera camera(md.mode, md.stereo, optics, Transformation::fromLookAt(vec3(0.5f, 1.0f, 3.0f), vec3(0.0f, 0.8f, -1.0f), vec3(0.0f, 1.0f, 0.0f)));
            Camera::RayHelper rh = camera.getRayHelper(0.0f, 96, 64);
            Prng prng(0);
            std::vector<float> pq, rays;
            for (int j = 0; j < 12; j++)
                for (int i = 0; i < 16; i++) {
                    float p = (i + 0.41f) / 16.0f, q = (j + 0.67f) / 12.0f;
                    pq.push_back(p);
                    pq.push_back(q);
                    Ray r = camera.getRay(p, q, 0.0f, 0.0f, rh, prng);
                    push3(rays, r.origin);
                    push3(rays, r.direction);
                }
            std::vector<float> desc = { optics.projection.l, optics.projection.r, optics.projection.b, optics.projection.t,
                camera.transformation.translation.x(), camera.transformation.translation.y(), camera.transformation.translation.z(),
                camera.transformation.rotation.x, camera.transformation.rotation.y, camera.transformation.rotation.z, camera.transformation.rotation.w,
                camera.transformation.scaling.x(), camera.transformation.scaling.y(), camera.transformation.scaling.z(),
                float(int(md.mode)), md.stereo };
            floats((std::string("camera_") + md.name + "_desc").c_str(), desc);
            floats((std::string("camera_") + md.name + "_pq").c_str(), pq);
            floats((std::string("camera_") + md.name + "_rays").c_str(), rays);
        }
    }

    /* ---- colour conversions the output side is built from (color.hpp:226-310) ---- */
    {
        std::mt19937 crng(99);
        auto cu01 = [&crng]() { return float(crng() >> 8) * (1.0f / 16777216.0f); };
        std::vector<float> in, out;
        std::vector<long long> bytes;
        for (int i = 0; i < 2048; i++) {
            float scale = (i % 4 == 0) ? 30.0f : (i % 4 == 1 ? 1.0f : (i % 4 == 2 ? 0.01f : 3.0f));
            vec3 rgb(cu01() * scale, cu01() * scale, cu01() * scale);
            if (i % 64 == 5) rgb = vec3(0.0f);
            if (i % 64 == 6) rgb = vec3(0.0031308f, 1.0f, 0.5f);
            float newY = cu01() * 100.0f;
            push3(in, rgb);
            in.push_back(newY);
            vec3 xyz = rgb_to_xyz(rgb);
            push3(out, xyz);
            push3(out, xyz_to_rgb(xyz));
            push3(out, adjust_y(xyz, newY));
            for (int c = 0; c < 3; c++) {
                float v = min(rgb[c], 1.0f);
                float s = rgb_to_srgb_helper(v);
                out.push_back(s);
                bytes.push_back(float_to_byte(s));
            }
        }
        floats("color_in", in);     /* rgb(3) newY */
        floats("color_out", out);   /* xyz(3) back to rgb(3) adjust_y(3) srgb of min(c,1) (3) */
        ints("color_srgb_bytes", bytes);
    }

    /* ---- computeTangents / computeNormals (geometryproc.hpp:58-226) ---- */
    {
        /* a Cornell wall quad and a small random indexed mesh */
        std::vector<vec3> pos = { vec3(-1.01f, 0.0f, 0.99f), vec3(-0.99f, 0.0f, -1.04f), vec3(-1.02f, 1.99f, -1.04f), vec3(-1.02f, 1.99f, 0.99f) };
        std::vector<vec3> nrm = { vec3(0.9999874f, 0.005025057f, 0.0f), vec3(0.9998379f, 0.01507292f, 0.009850611f), vec3(0.9999874f, 0.005025057f, 0.0f), vec3(0.9999874f, 0.005025057f, 0.0f) };
        std::vector<vec2> tc = { vec2(0.0f, 0.0f), vec2(1.0f, 0.0f), vec2(1.0f, 1.0f), vec2(0.0f, 1.0f) };
        std::vector<unsigned int> ind = { 0, 1, 2, 0, 2, 3 };
        int grid = 6;
        unsigned int base = pos.size();
        for (int j = 0; j <= grid; j++)
            for (int i = 0; i <= grid; i++) {
                pos.push_back(vec3(i / float(grid) + 0.05f * u01(), 0.3f * u01(), j / float(grid) + 0.05f * u01()) + vec3(3.0f, 0.0f, 0.0f));
                nrm.push_back(vec3(0.0f));
                tc.push_back(vec2(i / float(grid), j / float(grid)));
            }
        for (int j = 0; j < grid; j++)
            for (int i = 0; i < grid; i++) {
                unsigned int a = base + j * (grid + 1) + i, b = a + 1, c = a + grid + 1, d = c + 1;
                ind.insert(ind.end(), { a, b, c, b, d, c });
            }
        std::vector<vec3> n2 = computeNormals(pos, ind);
        for (size_t i = base; i < pos.size(); i++)
            nrm[i] = n2[i];
        std::vector<vec3> tng = computeTangents(pos, nrm, tc, ind);
        std::vector<vec3> n1 = computeNormals(pos, ind, NormalsFromFaceAverage);
        std::vector<float> fpos, fnrm, ftc, ftng, fn0, fn1;
        std::vector<long long> find(ind.begin(), ind.end());
        for (size_t i = 0; i < pos.size(); i++) {
            push3(fpos, pos[i]);
            push3(fnrm, nrm[i]);
            ftc.push_back(tc[i].x());
            ftc.push_back(tc[i].y());
            push3(ftng, tng[i]);
            push3(fn0, n2[i]);
            push3(fn1, n1[i]);
        }
        floats("geom_pos", fpos);
        floats("geom_nrm", fnrm);
        floats("geom_tc", ftc);
        ints("geom_ind", find);
        floats("geom_tangents", ftng);
        floats("geom_normals_weighted", fn0);
        floats("geom_normals_average", fn1);
    }

    /* ---- HitableSphere (hitable_sphere.hpp:32-220) and the samplers it uses ---- */
    {
        std::mt19937 srng(4242);
        auto su01 = [&srng]() { return float(srng() >> 8) * (1.0f / 16777216.0f); };
        auto sdir = [&]() {
            for (;;) {
                vec3 d(su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f);
                float l = dot(d, d);
                if (l > 1e-4f && l <= 1.0f)
                    return normalize(d);
            }
        };
        std::vector<const Animation*> noAnimations;
        AnimationCache cache(noAnimations, 0.0f);
        std::vector<float> spheres, rays, hits, pdfs, dirs, onSphereU, onSphereOut, toSphereIn, toSphereOut;
        std::vector<long long> seeds;
        const int n = 2048;
        for (int i = 0; i < n; i++) {
            vec3 center(su01() * 4.0f - 2.0f, su01() * 4.0f - 2.0f, su01() * 4.0f - 2.0f);
            float radius = 0.05f + 1.5f * su01();
            quat rot = (i % 3 == 0) ? quat::null() : toQuat(radians(360.0f * su01()), sdir());
            /* non-uniform scaling on purpose: the radius is max(scaling) */
            vec3 scaling = (i % 5 == 0) ? vec3(radius * 0.5f, radius, radius * 0.25f) : vec3(radius);
            HitableSphere sphere(Transformation(center, rot, scaling), nullptr);
            /* origins: far away, close to the surface, inside, exactly on the centre */
            vec3 origin;
            switch (i % 8) {
            case 0: origin = center; break;
            case 1: origin = center + (radius * 0.5f) * sdir(); break;
            case 2: origin = center + (radius * (1.0f + 1e-4f)) * sdir(); break;
            case 3: origin = center + (radius * (1.0f - 1e-4f)) * sdir(); break;
            default: origin = center + (radius * (1.5f + 6.0f * su01())) * sdir(); break;
            }
            /* directions: towards the sphere (with jitter that also produces grazing rays and misses) or random */
            vec3 direction = (i % 4 == 3) ? sdir() : normalize(center + (radius * 1.05f * su01()) * sdir() - origin + vec3(1e-6f));
            float amin = (i % 7 == 0) ? 0.0f : 1e-5f;
            float amax = (i % 11 == 0) ? 3.0f : maxval;
            Ray ray(origin, direction, 0.0f, 1.0f);
            Prng unused(0);
            HitRecord hr = sphere.hit(ray, RayIntersectionHelper(ray), amin, amax, 0.0f, cache, unused);
            push3(spheres, center);
            spheres.push_back(radius);
            spheres.push_back(rot.x); spheres.push_back(rot.y); spheres.push_back(rot.z); spheres.push_back(rot.w);
            spheres.push_back(scaling.x()); spheres.push_back(scaling.y()); spheres.push_back(scaling.z());
            push3(rays, origin);
            push3(rays, direction);
            rays.push_back(amin);
            rays.push_back(amax);
            hits.push_back(hr.haveHit ? 1.0f : 0.0f);
            hits.push_back(hr.haveHit ? hr.a : 0.0f);
            push3(hits, hr.haveHit ? hr.position : vec3(0.0f));
            push3(hits, hr.haveHit ? hr.normal : vec3(0.0f));
            push3(hits, hr.haveHit ? hr.tangent : vec3(0.0f));
            hits.push_back(hr.haveHit ? hr.texcoords.x() : 0.0f);
            hits.push_back(hr.haveHit ? hr.texcoords.y() : 0.0f);
            hits.push_back(hr.haveHit && hr.backside ? 1.0f : 0.0f);
            pdfs.push_back(sphere.pdfValue(origin, direction, cache, unused));
            Prng prng(1000 + i);
            seeds.push_back(1000 + i);
            push3(dirs, sphere.direction(origin, cache, prng));
        }
        /* the same with an animation on every sphere (hitable_sphere.hpp:118-127,161-166,196-203): hit() and direction()
         * add the animation's translation to the centre, pdfValue() applies the whole transformation to it */
        {
            std::vector<float> aSpheres, aKeys, aRays, aHits, aPdfs, aDirs;
            std::vector<long long> aSeeds;
            const float time = 0.37f;
            std::mt19937 arng(777); /* its own generator: the vectors after this block keep their values */
            auto su01 = [&arng]() { return float(arng() >> 8) * (1.0f / 16777216.0f); };
            auto sdir = [&]() {
                for (;;) {
                    vec3 d(su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f);
                    float l = dot(d, d);
                    if (l > 1e-4f && l <= 1.0f)
                        return normalize(d);
                }
            };
            for (int i = 0; i < 768; i++) {
                vec3 center = (i % 2 == 0) ? vec3(0.0f) : vec3(su01() - 0.5f, su01() - 0.5f, su01() - 0.5f);
                float radius = (i % 2 == 0) ? 1.0f : 0.2f + su01();
                quat rot = (i % 2 == 0) ? quat::null() : toQuat(radians(360.0f * su01()), sdir());
                AnimationKeyframes* anim = new AnimationKeyframes(0.0f,
                        Transformation(vec3(su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f), toQuat(radians(360.0f * su01()), sdir()),
                            vec3(0.3f + su01(), 0.3f + su01(), 0.3f + su01())),
                        1.0f,
                        Transformation(vec3(su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f, su01() * 2.0f - 1.0f), toQuat(radians(360.0f * su01()), sdir()),
                            vec3(0.3f + su01(), 0.3f + su01(), 0.3f + su01())));
                std::vector<const Animation*> animations(1, anim);
                AnimationCache acache(animations, time);
                HitableSphere sphere(Transformation(center, rot, vec3(radius)), nullptr, 0);
                const Transformation T = anim->at(time);
                const vec3 movedCenter = center + T.translation;
                const float movedRadius = radius * max(T.scaling);
                vec3 origin;
                switch (i % 6) {
                case 0: origin = movedCenter + (movedRadius * 0.5f) * sdir(); break;
                case 1: origin = T * center + (movedRadius * 0.3f) * sdir(); break;      /* inside as pdfValue() sees it */
                default: origin = movedCenter + (movedRadius * (1.5f + 5.0f * su01())) * sdir(); break;
                }
                vec3 direction = (i % 4 == 3) ? sdir() : normalize(movedCenter + (movedRadius * 1.05f * su01()) * sdir() - origin + vec3(1e-6f));
                float amin = 1e-5f, amax = maxval;
                Ray ray(origin, direction, time, 1.0f);
                Prng unused(0);
                HitRecord hr = sphere.hit(ray, RayIntersectionHelper(ray), amin, amax, 0.0f, acache, unused);
                push3(aSpheres, center);
                aSpheres.push_back(radius);
                aSpheres.push_back(rot.x); aSpheres.push_back(rot.y); aSpheres.push_back(rot.z); aSpheres.push_back(rot.w);
                aSpheres.push_back(radius); aSpheres.push_back(radius); aSpheres.push_back(radius);
                for (const AnimationKeyframes::Keyframe& k : anim->keyframes()) {
                    aKeys.push_back(k.t);
                    push3(aKeys, k.transformation.translation);
                    aKeys.push_back(k.transformation.rotation.x); aKeys.push_back(k.transformation.rotation.y);
                    aKeys.push_back(k.transformation.rotation.z); aKeys.push_back(k.transformation.rotation.w);
                    push3(aKeys, k.transformation.scaling);
                }
                push3(aRays, origin);
                push3(aRays, direction);
                aRays.push_back(amin);
                aRays.push_back(amax);
                aHits.push_back(hr.haveHit ? 1.0f : 0.0f);
                aHits.push_back(hr.haveHit ? hr.a : 0.0f);
                push3(aHits, hr.haveHit ? hr.position : vec3(0.0f));
                push3(aHits, hr.haveHit ? hr.normal : vec3(0.0f));
                push3(aHits, hr.haveHit ? hr.tangent : vec3(0.0f));
                aHits.push_back(hr.haveHit ? hr.texcoords.x() : 0.0f);
                aHits.push_back(hr.haveHit ? hr.texcoords.y() : 0.0f);
                aHits.push_back(hr.haveHit && hr.backside ? 1.0f : 0.0f);
                aPdfs.push_back(sphere.pdfValue(origin, direction, acache, unused));
                Prng prng(5000 + i);
                aSeeds.push_back(5000 + i);
                push3(aDirs, sphere.direction(origin, acache, prng));
                delete anim;
            }
            floats("sphere_anim_records", aSpheres);    /* as sphere_records */
            floats("sphere_anim_keyframes", aKeys);     /* two key frames per sphere, 11 floats each; evaluated at t = 0.37 */
            floats("sphere_anim_rays", aRays);
            floats("sphere_anim_hits", aHits);
            floats("sphere_anim_pdf", aPdfs);
            ints("sphere_anim_direction_seeds", aSeeds);
            floats("sphere_anim_direction", aDirs);
        }
        floats("sphere_records", spheres); /* centre(3) radius rotation(xyzw) scaling(3) */
        floats("sphere_rays", rays);       /* origin(3) direction(3) amin amax */
        floats("sphere_hits", hits);       /* haveHit a position(3) normal(3) tangent(3) texcoords(2) backside */
        floats("sphere_pdf", pdfs);
        ints("sphere_direction_seeds", seeds);
        floats("sphere_direction", dirs);
        for (int j = 0; j < 16; j++)
            for (int i = 0; i < 16; i++) {
                vec2 u((i + 0.37f) / 16.0f, (j + 0.61f) / 16.0f);
                onSphereU.push_back(u.x());
                onSphereU.push_back(u.y());
                push3(onSphereOut, Sampler::onUnitSphere(u));
                vec3 d = sdir();
                float cosThetaMax = su01();
                push3(toSphereIn, d);
                toSphereIn.push_back(cosThetaMax);
                push3(toSphereOut, Sampler::toSphere(d, cosThetaMax, u));
            }
        floats("sampler_sphere_u", onSphereU);
        floats("sampler_on_unit_sphere", onSphereOut);
        floats("sampler_to_sphere_in", toSphereIn); /* direction(3) cosThetaMax */
        floats("sampler_to_sphere", toSphereOut);
    }

    /* ---- powitacq_rgb::BRDF (powitacq_rgb.inl:856-1185), the model behind MaterialRGL, on the
     * synthetic tensor files of tests/golden/ (argv[2] = that directory) ---- */
    if (argc > 2) {
        const char* files[2] = { "synthetic_iso.bsdf", "synthetic_aniso.bsdf" };
        for (int f = 0; f < 2; f++) {
            powitacq_rgb::BRDF brdf(std::string(argv[2]) + "/" + files[f]);
            std::mt19937 brng(777 + f);
            auto bu01 = [&brng]() { return float(brng() >> 8) * (1.0f / 16777216.0f); };
            auto hemi = [&](bool allowBelow) {
                for (;;) {
                    vec3 d(bu01() * 2.0f - 1.0f, bu01() * 2.0f - 1.0f, allowBelow ? bu01() * 2.0f - 1.0f : bu01());
                    float l = dot(d, d);
                    if (l > 1e-4f && l <= 1.0f)
                        return normalize(d);
                }
            };
            std::vector<float> in, sampleOut, evalOut;
            const int n = 1024;
            for (int i = 0; i < n; i++) {
                vec3 wi = hemi(i % 16 == 0);
                if (i % 32 == 1)
                    wi = vec3(0.0f, 0.0f, 1.0f);                 /* normal incidence */
                if (i % 32 == 2)
                    wi = normalize(vec3(1.0f, 0.0f, 1e-3f));     /* grazing */
                vec3 wo = hemi(i % 16 == 8);
                float u0 = bu01(), u1 = bu01();
                if (i % 64 == 3) { u0 = 0.0f; u1 = 0.99999994f; }
                push3(in, wi);
                push3(in, wo);
                in.push_back(u0);
                in.push_back(u1);
                powitacq_rgb::Vector3f pwo;
                float pdf = 0.0f;
                powitacq_rgb::Vector3f w = brdf.sample(powitacq_rgb::Vector2f(u0, u1), powitacq_rgb::Vector3f(wi.x(), wi.y(), wi.z()), &pwo, &pdf);
                sampleOut.push_back(w.x()); sampleOut.push_back(w.y()); sampleOut.push_back(w.z());
                sampleOut.push_back(pwo.x()); sampleOut.push_back(pwo.y()); sampleOut.push_back(pwo.z());
                sampleOut.push_back(pdf);
                powitacq_rgb::Vector3f e = brdf.eval(powitacq_rgb::Vector3f(wi.x(), wi.y(), wi.z()), powitacq_rgb::Vector3f(wo.x(), wo.y(), wo.z()));
                evalOut.push_back(e.x()); evalOut.push_back(e.y()); evalOut.push_back(e.z());
                evalOut.push_back(brdf.pdf(powitacq_rgb::Vector3f(wi.x(), wi.y(), wi.z()), powitacq_rgb::Vector3f(wo.x(), wo.y(), wo.z())));
            }
            std::string k = f == 0 ? "rgl_iso" : "rgl_aniso";
            floats((k + "_in").c_str(), in);            /* wi(3) wo(3) u(2) */
            floats((k + "_sample").c_str(), sampleOut); /* weight(3) wo(3) pdf */
            floats((k + "_eval").c_str(), evalOut);     /* f*cos (3) pdf */
        }
    }

    /* ---- the vendored tinyobjloader on the fixture file of tests/golden/obj (argv[2]/obj) ---- */
    if (argc > 2)
        dumpObj(std::string(argv[2]) + "/obj/cases.obj");

    /* ---- the transformations of the reference's own tests/test-transformation.cpp: translate / rotate / scale in every
     * order and two products of three; each as Transformation (10 floats), its toMat4() (16) and the same chain made
     * with mat4 operations (16), which that test asserts to agree within 1e-4 ---- */
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
        std::vector<float> res;
        for (int i = 0; i < 8; i++) {
            push3(res, T[i].translation);
            res.push_back(T[i].rotation.x); res.push_back(T[i].rotation.y); res.push_back(T[i].rotation.z); res.push_back(T[i].rotation.w);
            push3(res, T[i].scaling);
            mat4 TM = T[i].toMat4();
            for (int k = 0; k < 16; k++) res.push_back(TM.values[k]);
            for (int k = 0; k < 16; k++) res.push_back(M[i].values[k]);
        }
        floats("transformation_chains", res);
    }

    /* ---- AnimationKeyframes::at, Transformation::toMat4 / toNormalMatrix / operator* (animation_keyframes.hpp:186-214,
     * transformation.hpp:80-83,105-122,199-205, quaternion slerp gvm.hpp:1765-1797) ---- */
    {
        AnimationKeyframes anim;
        const vec3 axes[5] = { vec3(0.0f, 1.0f, 0.0f), vec3(1.0f, 0.3f, -0.2f), vec3(-0.4f, 0.1f, 1.0f), vec3(0.0f, 0.0f, 1.0f), vec3(0.7f, -0.7f, 0.1f) };
        const float angles[5] = { 0.0f, 40.0f, 170.0f, 260.0f, 260.0f };    /* 170 -> 260 degrees: slerp meets a negative dot product; the last two rotate alike */
        const float times[5] = { -0.5f, 0.25f, 1.0f, 1.75f, 4.0f };
        std::vector<float> kf;
        for (int k = 0; k < 5; k++) {
            Transformation T(vec3(0.3f * k - 0.5f, 0.2f * k * k, -0.7f * k + 0.1f), toQuat(radians(angles[k]), axes[k == 4 ? 3 : k]),
                    vec3(1.0f + 0.25f * k, 1.0f, 1.0f - 0.1f * k));
            anim.addKeyframe(times[k], T);
            kf.push_back(times[k]);
            push3(kf, T.translation);
            kf.push_back(T.rotation.x); kf.push_back(T.rotation.y); kf.push_back(T.rotation.z); kf.push_back(T.rotation.w);
            push3(kf, T.scaling);
        }
        std::vector<float> ts, res;
        Prng prng(5);
        for (int i = 0; i < 96; i++) {
            float t = i < 5 ? times[i] : i < 8 ? (i == 5 ? -3.0f : i == 6 ? 9.0f : 0.25f + 1e-7f) : -0.7f + 5.0f * prng.in01();
            Transformation T = anim.at(t);
            mat4 M = T.toMat4();
            mat3 N = T.toNormalMatrix();
            vec3 p = vec3(prng.in01() - 0.5f, prng.in01() - 0.5f, prng.in01() - 0.5f);
            vec3 viaM = (M * vec4(p, 1.0f)).xyz();
            vec3 viaN = N * p;
            vec3 viaT = T * p;
            ts.push_back(t);
            push3(ts, p);
            push3(res, T.translation);
            res.push_back(T.rotation.x); res.push_back(T.rotation.y); res.push_back(T.rotation.z); res.push_back(T.rotation.w);
            push3(res, T.scaling);
            for (int k = 0; k < 16; k++) res.push_back(M.values[k]);
            for (int k = 0; k < 9; k++) res.push_back(N.values[k]);
            push3(res, viaM);
            push3(res, viaN);
            push3(res, viaT);
        }
        floats("anim_keyframes", kf);      /* per key frame: t, translation (3), rotation xyzw (4), scaling (3) */
        floats("anim_in", ts);             /* per case: t, point (3) */
        floats("anim_out", res);           /* per case: transformation (10), toMat4 (16, column major), normal matrix (9), M * p, N * p, T * p */
    }

    fprintf(out, "\n}\n");
    if (out != stdout)
        fclose(out);
    return 0;
}
