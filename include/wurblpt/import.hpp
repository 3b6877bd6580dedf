/*
 * import.hpp -- importIntoScene(): Wavefront OBJ + MTL -> Scene, with the reference's material
 * mapping and flags (import.hpp:64-504).
 *
 * What the result depends on is kept: the material rules (Lambertian when nothing else is needed,
 * glass on request, ModPhong otherwise; the Tf / d fix-ups; two-sided wrappers; which materials
 * become hot spots), one MeshInstance per (material, shape) with vertices de-duplicated by their
 * (position, normal, texcoord) index triple in order of first use, normals normalised or recomputed,
 * tangents only for normal-mapped materials, bump maps converted to normal maps.
 * Two things differ on purpose:
 *  - the reference imports the materials' geometry in an OpenMP loop whose order of completion
 *    decides the hitable order; here the loop is serial (no material first, then the materials in
 *    MTL order, shapes in file order inside each), so an import is reproducible;
 *  - images come from imageio.hpp (row 0 = bottom) instead of libtgd; files it cannot decode become
 *    the dummy textures the reference uses for load failures.
 *
 * Interface (function names, flags, argument order) and the material rules follow marlam/wurblpt (import.hpp), which is
 * distributed under the MIT licence: Copyright (c) 2023 Martin Lambers <marlam@marlam.de>; the permission notice is
 * reproduced in the LICENSE file of this repository.  The implementation is this repository's own.
 */
#pragma once

#include <cstdio>
#include <array>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "geometryproc.hpp"
#include "imageio.hpp"
#include "material.hpp"
#include "mesh.hpp"
#include "objreader.hpp"
#include "scene.hpp"
#include "texture.hpp"
#include "transformation.hpp"

namespace WurblPT {

/* color.hpp:280-294 */
inline float byte_to_float(uint8_t x) { return x / 255.0f; }
inline uint8_t float_to_byte(float x) { return uint8_t(std::round(x * 255.0f)); }

/* import.hpp:64-90: normal map from a bump (height) map by central differences */
inline ArrayContainer toNormalMap(const ArrayContainer& bumpMap, float bumpScaling = 8.0f)
{
    Array<uint8_t> normalMap(bumpMap.dimension(0), bumpMap.dimension(1), 3);
    const int w = int(normalMap.dimension(0)), h = int(normalMap.dimension(1));
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const uint8_t height_r = bumpMap.get<uint8_t>(size_t(y) * w + std::min(x + 1, w - 1))[0];
            const uint8_t height_l = bumpMap.get<uint8_t>(size_t(y) * w + std::max(x - 1, 0))[0];
            const uint8_t height_t = bumpMap.get<uint8_t>(size_t(std::min(y + 1, h - 1)) * w + x)[0];
            const uint8_t height_b = bumpMap.get<uint8_t>(size_t(std::max(y - 1, 0)) * w + x)[0];
            const vec3 tx = vec3(2.0f, 0.0f, bumpScaling * (byte_to_float(height_r) - byte_to_float(height_l)));
            const vec3 ty = vec3(0.0f, 2.0f, bumpScaling * (byte_to_float(height_t) - byte_to_float(height_b)));
            const vec3 n = normalize(cross(tx, ty));
            uint8_t* rgb = normalMap.get<uint8_t>(size_t(y) * w + x);
            rgb[0] = float_to_byte(0.5f * (n.x() + 1.0f));
            rgb[1] = float_to_byte(0.5f * (n.y() + 1.0f));
            rgb[2] = float_to_byte(0.5f * (n.z() + 1.0f));
        }
    }
    return normalMap;
}

/* Does every texel of an 8-bit image carry one grey level in its first three components?  (Images with fewer than three
 * components are grey by construction.)  Height maps are sometimes stored as RGB files; the importer tells them from normal
 * maps by this (import.hpp:140-160). */
inline bool isGreyImage(const ArrayContainer& img)
{
    if (img.componentCount() < 3)
        return true;
    const size_t n = img.elementCount();
    for (size_t e = 0; e < n; e++) {
        const uint8_t* t = img.get<uint8_t>(e);
        if (!(t[0] == t[1] && t[1] == t[2]))
            return false;
    }
    return true;
}

/* The Texture an MTL statement names: one per (file, factor, offset, linearisation, bump scaling), kept in textureMap
 * (import.hpp:93-169).  bumpScaling > 0: the file is a bump or normal map -- a grey one is a height map and becomes a normal
 * map first.  Files that could not be read (empty containers) and bump maps that are not 8-bit get constant stand-ins. */
inline Texture* importTexture(std::map<std::string, Texture*>& textureMap, const std::map<std::string, ArrayContainer>& textureFileMap,
        const std::string& name, const float factor[3], const float offset[3], int* componentCount = nullptr,
        LinearizeSRGBType linearizeSRGBType = LinearizeSRGB_Auto, float bumpScaling = -1.0f /* < 0: not a bump map */)
{
    const char* lin = linearizeSRGBType == LinearizeSRGB_On ? "on" : linearizeSRGBType == LinearizeSRGB_Off ? "off" : "auto";
    const std::string cacheName = name + "_factor=" + std::to_string(factor[0]) + ',' + std::to_string(factor[1])
        + "_offset=" + std::to_string(offset[0]) + ',' + std::to_string(offset[1]) + "_linsrgb=" + lin
        + "_bumpscal=" + std::to_string(bumpScaling);
    const auto known = textureMap.find(cacheName);
    if (known != textureMap.end())
        return known->second;
    auto keep = [&](Texture* tex) {
        textureMap.emplace(cacheName, tex);
        return tex;
    };
    auto standIn = [&](const vec4& value, const char* why) {
        fprintf(stderr, "    texture %s: %sreplaced with dummy texture\n", cacheName.c_str(), why);
        return keep(new TextureConstant(value));
    };
    const ArrayContainer& file = textureFileMap.at(name);
    if (file.elementCount() == 0)
        return standIn(vec4(0.5f), "");
    if (componentCount)
        *componentCount = int(file.componentCount());
    const bool bump = bumpScaling > 0.0f;
    if (bump && file.componentType() != uint8)
        return standIn(vec4(0.5f, 0.5f, 1.0f, 0.0f), "cannot handle a bump/normal map that is not uint8, ");
    const vec2 f(factor[0], factor[1]), o(offset[0], offset[1]);
    if (bump && isGreyImage(file))
        return keep(createTextureImage(toNormalMap(file, bumpScaling), linearizeSRGBType, f, o));
    return keep(createTextureImage(file, linearizeSRGBType, f, o));
}

constexpr unsigned int ImportBitDisableLightSources = (1 << 0); /* import.hpp:187-196 */
constexpr unsigned int ImportBitDisableHotSpots = (1 << 1);
constexpr unsigned int ImportBitTwoSidedMaterials = (1 << 2);
constexpr unsigned int ImportBitInvertedTf = (1 << 3);
constexpr unsigned int ImportBitWithGlass = (1 << 4);

/* The vertices of one part of a shape (its faces of one material), de-indexed: an OBJ corner is a triple (position, normal,
 * texture coordinate) of independent indices, a Mesh wants one index per distinct corner.  add() hands every triple its
 * number in order of first appearance (the reference's tuple map, import.hpp:416-464).  A part keeps its normals / texture
 * coordinates only as long as every corner so far had them; from the first corner without (or with a normal that is not a
 * direction) the arrays stop growing and the caller recomputes or drops them. */
struct ShapeVertices {
    enum Status { Ok, InvalidNormal /* a warning: normals will be recomputed */, BadPosition, BadAttribute };
    const ObjData& obj;
    std::map<std::array<int, 3>, unsigned int> number;
    std::vector<vec3> positions, normals;
    std::vector<vec2> texcoords;
    std::vector<unsigned int> indices;
    bool haveNormals = true, haveTexCoords = true;

    explicit ShapeVertices(const ObjData& o) : obj(o) {}

    Status add(const ObjIndex& corner)
    {
        const std::array<int, 3> key = { corner.vertex, corner.normal, corner.texcoord };
        const auto seen = number.find(key);
        if (seen != number.end()) {
            indices.push_back(seen->second);
            return Ok;
        }
        if (corner.vertex < 0 || size_t(corner.vertex) >= obj.vertices.size() / 3)
            return BadPosition;
        Status status = Ok;
        positions.push_back(vec3(obj.vertices.data() + 3 * corner.vertex));
        haveNormals = haveNormals && corner.normal >= 0;
        haveTexCoords = haveTexCoords && corner.texcoord >= 0;
        if (haveNormals) {
            if (size_t(corner.normal) >= obj.normals.size() / 3)
                return BadAttribute;
            const vec3 n = vec3(obj.normals.data() + 3 * corner.normal);
            if (all(isfinite(n)) && dot(n, n) >= epsilon) {
                normals.push_back(normalize(n));
            } else {
                haveNormals = false;
                status = InvalidNormal;
            }
        }
        if (haveTexCoords) {
            if (size_t(corner.texcoord) >= obj.texcoords.size() / 2)
                return BadAttribute;
            texcoords.push_back(vec2(obj.texcoords.data() + 2 * corner.texcoord));
        }
        const unsigned int fresh = unsigned(number.size());
        number.emplace(key, fresh);
        indices.push_back(fresh);
        return status;
    }
};

/* MaterialGlass::transparentColorToAbsorption (material_glass.hpp:154-165) */
inline vec3 transparentColorToAbsorption(const vec3& c, float targetDistance = 0.01f)
{
    auto one = [&](float t) { return max(-std::log(t) / targetDistance, 0.0f); };
    return vec3(one(c.r()), one(c.g()), one(c.b()));
}

/* import.hpp:198-504 */
inline bool importIntoScene(Scene& scene, const std::string& filename, const Transformation& transformation = Transformation(),
        unsigned int importBits = 0)
{
    fprintf(stderr, "%s: importing...\n", filename.c_str());
    ObjData obj;
    const bool valid = loadObj(filename, obj);
    if (!obj.warning.empty())
        fprintf(stderr, "  warning: %s", obj.warning.c_str());
    if (!valid) {
        fprintf(stderr, "  error: %s%s: import failure\n", obj.error.c_str(), filename.c_str());
        return false;
    }
    const std::vector<ObjMaterial>& objMaterials = obj.materials;
    std::string basedir = ".";
    const size_t dirSep = filename.find_last_of("/\\");
    if (dirSep != std::string::npos)
        basedir = filename.substr(0, dirSep);

    /* all texture files, each loaded once */
    std::set<std::string> textureFileSet;
    for (const ObjMaterial& M : objMaterials) {
        for (const std::string* n : { &M.normalTex, &M.bumpTex, &M.diffuseTex, &M.specularTex, &M.shininessTex, &M.alphaTex, &M.emissiveTex })
            if (n->size() > 0)
                textureFileSet.insert(*n);
    }
    std::map<std::string, ArrayContainer> textureFileMap;
    for (const std::string& name : textureFileSet) {
        std::string fileName = basedir + "/" + name;
        for (char& c : fileName)
            if (c == '\\')
                c = '/';
        std::string err;
        ArrayContainer img = loadImage(fileName, &err);
        fprintf(stderr, "    %s: %s\n", name.c_str(), err.empty() ? "ok" : err.c_str());
        textureFileMap.insert(std::pair<std::string, ArrayContainer>(name, img));
    }

    /* the materials.  An MTL record becomes the simplest material that can show it (the reference's rules,
     * import.hpp:288-387): Lambertian when nothing but a diffuse colour / texture without alpha is set; glass, on
     * request, for an untextured transparent record; modified Phong for everything else. */
    std::vector<Material*> materials;
    std::vector<std::string> materialNames;
    std::vector<bool> materialIsLight, materialWantsTangents;
    std::map<std::string, Texture*> textureMap;
    const bool lightsOff = (importBits & ImportBitDisableLightSources) != 0;
    auto textureOf = [&](const std::string& file, const ObjTexOpt& opt, int* components = nullptr,
            LinearizeSRGBType linearize = LinearizeSRGB_Auto, float bumpMultiplier = -1.0f /* not a bump map */) -> Texture* {
        if (file.empty())
            return nullptr;
        return importTexture(textureMap, textureFileMap, file, opt.scale, opt.originOffset, components, linearize, bumpMultiplier);
    };
    for (const ObjMaterial& M : objMaterials) {
        /* opacity and transmission filter, reconciled: `d` and `Tf` are two spellings of one thing in the wild */
        vec3 transmission = (importBits & ImportBitInvertedTf) ? vec3(1.0f) - vec3(M.transmittance) : vec3(M.transmittance);
        float opacity = M.dissolve;
        const vec3 diffuse(M.diffuse), specular(M.specular), emission(M.emission);
        if (opacity >= 1.0f && max(transmission) < 1.0f) {
            opacity = average(transmission);
            transmission = vec3(1.0f) - transmission;
        }
        if (opacity < 1.0f && max(transmission) <= 0.0f)
            transmission = (1.0f - opacity) * diffuse;
        Texture* normalMap = M.normalTex.empty()
            ? textureOf(M.bumpTex, M.bumpOpt, nullptr, LinearizeSRGB_Off, M.bumpOpt.bumpMultiplier)
            : textureOf(M.normalTex, M.normalOpt, nullptr, LinearizeSRGB_Off);
        int diffuseComponents = 0;
        Texture* diffuseMap = textureOf(M.diffuseTex, M.diffuseOpt, &diffuseComponents);
        const bool diffuseMapHasAlpha = diffuseMap && (diffuseComponents == 2 || diffuseComponents == 4);
        const bool emits = !lightsOff && (max(emission) > 0.0f || !M.emissiveTex.empty());
        const bool hasSpecular = max(specular) > 0.0f || !M.specularTex.empty();
        const bool opaque = opacity >= 1.0f && M.alphaTex.empty();
        Material* made = nullptr;
        bool isLight = false;
        if (opaque && !diffuseMapHasAlpha && !hasSpecular && !emits) {
            made = new MaterialLambertian(diffuse, diffuseMap);
        } else if ((importBits & ImportBitWithGlass) && opacity < 1.0f && M.alphaTex.empty() && !diffuseMap && M.specularTex.empty()
                && max(emission) <= 0.0f && M.emissiveTex.empty()) {
            made = new MaterialGlass(transparentColorToAbsorption(diffuse), M.ior);
        } else {
            MaterialModPhong* phong = new MaterialModPhong;
            int specularComponents = 0;
            phong->haveNIR = false;
            phong->diffuse = vec4(diffuse, 0.0f);
            phong->diffuseTex = diffuseMap;
            phong->diffuseTexHasAlpha = diffuseMapHasAlpha;
            phong->specular = vec4(specular, 0.0f);
            phong->specularTex = textureOf(M.specularTex, M.specularOpt, &specularComponents);
            phong->specularTexHasAlpha = phong->specularTex && (specularComponents == 2 || specularComponents == 4);
            phong->shininess = M.shininess;
            phong->shininessTex = textureOf(M.shininessTex, M.shininessOpt, nullptr, LinearizeSRGB_Off);
            phong->opacity = opacity;
            phong->opacityTex = textureOf(M.alphaTex, M.alphaOpt, nullptr, LinearizeSRGB_Off);
            phong->indexOfRefraction = M.ior;
            phong->transmissive = vec4(transmission, 0.0f);
            if (!lightsOff) {
                phong->emissive = vec4(emission, 0.0f);
                phong->emissiveTex = textureOf(M.emissiveTex, M.emissiveOpt);
            }
            isLight = dot(phong->emissive, phong->emissive) > 0.0f || phong->emissiveTex;
            made = phong;
        }
        made->normalTex = normalMap;
        materials.push_back(made);
        materialIsLight.push_back(isLight);
        materialWantsTangents.push_back(normalMap != nullptr);
        materialNames.push_back(M.name);
    }
    for (auto it = textureMap.cbegin(); it != textureMap.cend(); it++)
        scene.take(it->second);
    for (size_t i = 0; i < materials.size(); i++)
        scene.take(materials[i], materialNames[i]);
    if (importBits & ImportBitTwoSidedMaterials) {
        for (size_t i = 0; i < materials.size(); i++) {
            MaterialTwoSided* mat = new MaterialTwoSided(materials[i], materials[i]);
            materials[i] = mat;
            scene.take(mat, materialNames[i]);
        }
    }

    /* the shapes: one MeshInstance per (material, shape); serial, see the header of this file (import.hpp:404-498) */
    Material* nullMaterial = scene.take(new MaterialLambertian(vec4(0.5f)));
    for (int matId = -1; matId < int(materials.size()); matId++) {
        for (size_t s = 0; s < obj.shapes.size(); s++) {
            const ObjShape& shape = obj.shapes[s];
            ShapeVertices part(obj);
            for (size_t i = 0; i < shape.indices.size(); i++) {
                if (shape.materialIds[i / 3] != matId)
                    continue;
                const ShapeVertices::Status st = part.add(shape.indices[i]);
                if (st == ShapeVertices::BadPosition)
                    fprintf(stderr, "%s: vertex index out of range in shape '%s'\n", filename.c_str(), shape.name.c_str());
                if (st == ShapeVertices::InvalidNormal)
                    fprintf(stderr, "      warning: invalid normals in shape %zu '%s'\n", s, shape.name.c_str());
                if (st == ShapeVertices::BadPosition || st == ShapeVertices::BadAttribute)
                    return false;
            }
            if (part.indices.empty())
                continue;
            if (!part.haveNormals)
                part.normals = computeNormals(part.positions, part.indices);
            if (!part.haveTexCoords)
                part.texcoords.clear();
            const bool real = matId >= 0;
            Mesh* mesh = scene.take(new Mesh(part.positions, part.normals, part.texcoords, part.indices, transformation,
                        real && materialWantsTangents[matId]));
            const bool hot = real && materialIsLight[matId] && !(importBits & ImportBitDisableHotSpots);
            scene.take(new MeshInstance(mesh, real ? materials[matId] : nullMaterial), hot ? HotSpot : ColdSpot);
        }
    }
    fprintf(stderr, "%s: import done\n", filename.c_str());
    return true;
}

}
