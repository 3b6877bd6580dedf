/*
 * material.hpp -- host-side material descriptions.
 *
 * Same classes and constructor signatures as the reference (material.hpp:158-334,
 * material_lambertian.hpp, light_diffuse.hpp, material_mirror.hpp, material_ggx.hpp,
 * material_glass.hpp, material_modphong.hpp).  The reference evaluates materials through
 * virtual scatter()/scatterToDirection()/emitted(); here each class describes itself as a
 * tagged wpt_material record that the HIP kernel evaluates.  A Material subclass that the
 * kernel does not know returns false from describe() and mcpt() refuses the scene.
 */
#pragma once

#include <utility>

#include "../wurblpt_hip.h"
#include "constants.hpp"
#include "gvm.hpp"
#include "scene_component.hpp"
#include "rgl.hpp"
#include "texture.hpp"

namespace WurblPT {

inline wpt_material wptEmptyMaterial(uint32_t type)
{
    wpt_material m;
    memset(&m, 0, sizeof(m));
    m.type = type;
    m.normal_tex = -1;
    for (int i = 0; i < 5; i++)
        m.tex[i] = -1;
    return m;
}

class Material : public SceneComponent
{
public:
    const Texture* normalTex;

    Material(const Texture* nt = nullptr) : normalTex(nt) {}
    virtual ~Material() {}

    /* The base class neither scatters nor emits (reference material.hpp:158-186). */
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const
    {
        out = wptEmptyMaterial(WPT_MAT_NONE);
        return setNormalTex(out, ctx);
    }

protected:
    bool setNormalTex(wpt_material& out, FlattenContext& ctx) const
    {
        out.normal_tex = ctx.indexOf(normalTex);
        return out.normal_tex != -2;
    }
    static bool setTex(wpt_material& out, int slot, const Texture* tex, FlattenContext& ctx)
    {
        out.tex[slot] = ctx.indexOf(tex);
        return out.tex[slot] != -2;
    }
};

class MaterialTwoSided final : public Material
{
private:
    const Material* _frontSideMaterial;
    const Material* _backSideMaterial;

public:
    MaterialTwoSided(const Material* frontSideMaterial, const Material* backSideMaterial) :
        Material(nullptr), _frontSideMaterial(frontSideMaterial), _backSideMaterial(backSideMaterial)
    {
    }
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_TWOSIDED);
        out.tex[0] = ctx.indexOf(_frontSideMaterial);
        out.tex[1] = ctx.indexOf(_backSideMaterial);
        return out.tex[0] >= 0 && out.tex[1] >= 0;
    }
};

class MaterialLambertian final : public Material
{
private:
    const vec4 _albedo;
    const Texture* _tex;
    const bool haveNIR;

public:
    MaterialLambertian(const vec4& albedo, const Texture* tex = nullptr) : _albedo(albedo), _tex(tex), haveNIR(true) {}
    MaterialLambertian(const vec3& albedo, const Texture* tex = nullptr) : _albedo(albedo, 0.0f), _tex(tex), haveNIR(false) {}
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_LAMBERTIAN);
        out.flags = haveNIR ? WPT_MATF_HAVE_NIR : 0;
        wptSet(out.v[0], _albedo);
        return setTex(out, 0, _tex, ctx) && setNormalTex(out, ctx);
    }
};

class LightDiffuse final : public Material
{
private:
    const vec4 _emit;
    const Texture* _tex;

public:
    LightDiffuse(const vec4& emit, const Texture* tex = nullptr) : _emit(emit), _tex(tex) {}
    LightDiffuse(const vec3& emit, const Texture* tex = nullptr) : _emit(emit, average(emit)), _tex(tex) {}
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_LIGHT_DIFFUSE);
        wptSet(out.v[0], _emit);
        return setTex(out, 0, _tex, ctx) && setNormalTex(out, ctx);
    }
};

class MaterialMirror final : public Material
{
public:
    const vec4 color;
    const Texture* colorTex;
    const bool haveNIR;

    MaterialMirror(const vec4& color = vec4(1.0f), const Texture* tex = nullptr) :
        Material(nullptr), color(color), colorTex(tex), haveNIR(true)
    {
    }
    MaterialMirror(const vec3& color, const Texture* tex = nullptr) :
        Material(nullptr), color(color, average(color)), colorTex(tex), haveNIR(false)
    {
    }
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_MIRROR);
        out.flags = haveNIR ? WPT_MATF_HAVE_NIR : 0;
        wptSet(out.v[0], color);
        return setTex(out, 0, colorTex, ctx) && setNormalTex(out, ctx);
    }
};

/* material_rgl.hpp:46-102: a measured BRDF of the RGL material database (RGB data set; the
 * near-infrared channel is the average of RGB).  The file is read and the model's tables are built
 * when the material is created; an unreadable file is reported when the scene is flattened. */
class MaterialRGL final : public Material
{
private:
    wpt_rgl_brdf _brdf;
    std::vector<float> _pool;
    std::string _error;

public:
    MaterialRGL(const std::string& filenameRgb) : Material(nullptr)
    {
        memset(&_brdf, 0, sizeof(_brdf));
        buildRglBrdf(filenameRgb, _pool, _brdf, _error);
    }
    bool valid() const { return _error.empty(); }
    const std::string& error() const { return _error; }
    const wpt_rgl_brdf& brdf() const { return _brdf; }
    const std::vector<float>& tables() const { return _pool; }
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        if (!valid()) {
            ctx.error = _error;
            return false;
        }
        out = wptEmptyMaterial(WPT_MAT_RGL);
        /* the tables move into the scene's pool: rebase the offsets */
        const uint32_t base = uint32_t(ctx.rglData.size());
        ctx.rglData.insert(ctx.rglData.end(), _pool.begin(), _pool.end());
        wpt_rgl_brdf b = _brdf;
        wpt_rgl_warp* warps[5] = { &b.ndf, &b.sigma, &b.vndf, &b.luminance, &b.rgb };
        for (wpt_rgl_warp* w : warps) {
            for (int k = 0; k < 3; k++)
                w->param_values[k] += base;
            w->data += base;
            if (w->marginal_cdf != WPT_RGL_NONE)
                w->marginal_cdf += base;
            if (w->conditional_cdf != WPT_RGL_NONE)
                w->conditional_cdf += base;
        }
        out.tex[0] = int(ctx.rglBrdfs.size());
        ctx.rglBrdfs.push_back(b);
        return setNormalTex(out, ctx);
    }
};

class MaterialGGX final : public Material
{
public:
    vec4 albedo;
    vec2 roughness;
    const Texture* albedoTex;
    const Texture* roughnessTex;

    MaterialGGX(const vec4& alb, const Texture* albTex, const vec2& r, const Texture* rTex) :
        albedo(alb), roughness(r), albedoTex(albTex), roughnessTex(rTex)
    {
    }
    MaterialGGX() : MaterialGGX(vec4(0.0f), nullptr, vec2(0.0f), nullptr) {}
    MaterialGGX(const vec4& alb, const vec2& r, const Texture* albTex = nullptr, const Texture* rTex = nullptr) :
        MaterialGGX(alb, albTex, r, rTex)
    {
    }
    MaterialGGX(const vec3& alb, const Texture* albTex, const vec2& r, const Texture* rTex) :
        MaterialGGX(vec4(alb, average(alb)), albTex, r, rTex)
    {
    }
    MaterialGGX(const vec3& alb, const vec2& r, const Texture* albTex = nullptr, const Texture* rTex = nullptr) :
        MaterialGGX(vec4(alb, average(alb)), albTex, r, rTex)
    {
    }
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_GGX);
        wptSet(out.v[0], albedo);
        out.f[0] = roughness.x();
        out.f[1] = roughness.y();
        return setTex(out, 0, albedoTex, ctx) && setTex(out, 1, roughnessTex, ctx) && setNormalTex(out, ctx);
    }
};

class MaterialGlass final : public Material
{
private:
    static vec4 toVec4(const vec3& v) { return vec4(v, average(v)); }
    static bool allComponentsEqual(const vec4& v) { return (v.r() == v.g()) && (v.r() == v.b()) && (v.r() == v.a()); }

public:
    const vec4 absorption;
    const vec4 refractiveIndexOfMaterial;
    const vec4 refractiveIndexOfSurroundingMedium;
    const bool chromaticDispersion;

    /* material_glass.hpp:154-188: absorption coefficient that leaves the given colour after targetDistance (1 cm), and back */
    static float transparentColorToAbsorption(float transparentColor, float targetDistance = 0.01f)
    {
        return max(-log(transparentColor) / targetDistance, 0.0f);
    }
    static vec3 transparentColorToAbsorption(const vec3& c, float targetDistance = 0.01f)
    {
        return vec3(transparentColorToAbsorption(c.r(), targetDistance), transparentColorToAbsorption(c.g(), targetDistance),
                transparentColorToAbsorption(c.b(), targetDistance));
    }
    static vec4 transparentColorToAbsorption(const vec4& c, float targetDistance = 0.01f)
    {
        return vec4(transparentColorToAbsorption(c.r(), targetDistance), transparentColorToAbsorption(c.g(), targetDistance),
                transparentColorToAbsorption(c.b(), targetDistance), transparentColorToAbsorption(c.a(), targetDistance));
    }
    static float absorptionToTransparentColor(float absorption, float targetDistance = 0.01f)
    {
        return exp(-absorption / targetDistance);
    }
    static vec3 absorptionToTransparentColor(const vec3& a, float targetDistance = 0.01f)
    {
        return vec3(absorptionToTransparentColor(a.r(), targetDistance), absorptionToTransparentColor(a.g(), targetDistance),
                absorptionToTransparentColor(a.b(), targetDistance));
    }

    MaterialGlass(const vec4& absorption, const vec4& refractiveIndexOfMaterial,
            const vec4& refractiveIndexOfSurroundingMedium = vec4(refractiveIndexOfVacuum)) :
        Material(nullptr), absorption(absorption), refractiveIndexOfMaterial(refractiveIndexOfMaterial),
        refractiveIndexOfSurroundingMedium(refractiveIndexOfSurroundingMedium),
        chromaticDispersion(!allComponentsEqual(refractiveIndexOfMaterial) || !allComponentsEqual(refractiveIndexOfSurroundingMedium))
    {
    }
    MaterialGlass(const vec4& absorption, float refractiveIndexOfMaterial, float refractiveIndexOfSurroundingMedium = refractiveIndexOfVacuum) :
        MaterialGlass(absorption, vec4(refractiveIndexOfMaterial), vec4(refractiveIndexOfSurroundingMedium))
    {
    }
    MaterialGlass(const vec3& absorption, const vec3& refractiveIndexOfMaterial,
            const vec3& refractiveIndexOfSurroundingMedium = vec3(refractiveIndexOfVacuum)) :
        MaterialGlass(toVec4(absorption), toVec4(refractiveIndexOfMaterial), toVec4(refractiveIndexOfSurroundingMedium))
    {
    }
    MaterialGlass(const vec3& absorption, float refractiveIndexOfMaterial, float refractiveIndexOfSurroundingMedium = refractiveIndexOfVacuum) :
        MaterialGlass(toVec4(absorption), vec4(refractiveIndexOfMaterial), vec4(refractiveIndexOfSurroundingMedium))
    {
    }
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_GLASS);
        out.flags = chromaticDispersion ? WPT_MATF_CHROMATIC_DISPERSION : 0;
        wptSet(out.v[0], absorption);
        wptSet(out.v[1], refractiveIndexOfMaterial);
        wptSet(out.v[2], refractiveIndexOfSurroundingMedium);
        return setNormalTex(out, ctx);
    }
};

class MaterialModPhong final : public Material
{
public:
    vec4 diffuse;
    const Texture* diffuseTex;
    bool diffuseTexHasAlpha;
    vec4 specular;
    const Texture* specularTex;
    bool specularTexHasAlpha;
    float shininess;
    const Texture* shininessTex;
    float opacity;
    const Texture* opacityTex;
    float indexOfRefraction;
    vec4 transmissive;
    vec4 emissive;
    const Texture* emissiveTex;
    bool haveNIR;

    MaterialModPhong() :
        diffuse(0.0f), diffuseTex(nullptr), diffuseTexHasAlpha(false), specular(0.0f), specularTex(nullptr),
        specularTexHasAlpha(false), shininess(0.0f), shininessTex(nullptr), opacity(0.0f), opacityTex(nullptr),
        indexOfRefraction(1.0f), transmissive(0.0f), emissive(0.0f), emissiveTex(nullptr), haveNIR(true)
    {
    }
    MaterialModPhong(const vec4& dif, const Texture* difTex, const vec4& spc = vec4(0.0f), const Texture* spcTex = nullptr,
            float shi = 100.0f, const Texture* shiTex = nullptr, float opa = 1.0f, const Texture* opaTex = nullptr) :
        MaterialModPhong()
    {
        diffuse = dif;
        diffuseTex = difTex;
        specular = spc;
        specularTex = spcTex;
        shininess = shi;
        shininessTex = shiTex;
        opacity = opa;
        opacityTex = opaTex;
    }
    MaterialModPhong(const vec3& dif, const Texture* difTex, const vec3& spc = vec3(0.0f), const Texture* spcTex = nullptr,
            float shi = 100.0f, const Texture* shiTex = nullptr, float opa = 1.0f, const Texture* opaTex = nullptr) :
        MaterialModPhong(vec4(dif, average(dif)), difTex, vec4(spc, average(spc)), spcTex, shi, shiTex, opa, opaTex)
    {
        haveNIR = false;
    }
    MaterialModPhong(const vec4& dif, const vec4& spc = vec4(0.0f), float shi = 100.0f, float opa = 1.0f,
            const Texture* difTex = nullptr, const Texture* spcTex = nullptr, const Texture* shiTex = nullptr,
            const Texture* opaTex = nullptr) :
        MaterialModPhong(dif, difTex, spc, spcTex, shi, shiTex, opa, opaTex)
    {
    }
    MaterialModPhong(const vec3& dif, const vec3& spc = vec3(0.0f), float shi = 100.0f, float opa = 1.0f,
            const Texture* difTex = nullptr, const Texture* spcTex = nullptr, const Texture* shiTex = nullptr,
            const Texture* opaTex = nullptr) :
        MaterialModPhong(vec4(dif, average(dif)), difTex, vec4(spc, average(spc)), spcTex, shi, shiTex, opa, opaTex)
    {
        haveNIR = false;
    }
    virtual bool describe(wpt_material& out, FlattenContext& ctx) const override
    {
        out = wptEmptyMaterial(WPT_MAT_MODPHONG);
        out.flags = (haveNIR ? WPT_MATF_HAVE_NIR : 0) | (diffuseTexHasAlpha ? WPT_MATF_DIFFUSE_TEX_HAS_ALPHA : 0)
            | (specularTexHasAlpha ? WPT_MATF_SPECULAR_TEX_HAS_ALPHA : 0);
        wptSet(out.v[0], diffuse);
        wptSet(out.v[1], specular);
        wptSet(out.v[2], transmissive);
        wptSet(out.v[3], emissive);
        out.f[0] = shininess;
        out.f[1] = opacity;
        out.f[2] = indexOfRefraction;
        return setTex(out, 0, diffuseTex, ctx) && setTex(out, 1, specularTex, ctx) && setTex(out, 2, shininessTex, ctx)
            && setTex(out, 3, opacityTex, ctx) && setTex(out, 4, emissiveTex, ctx) && setNormalTex(out, ctx);
    }
};

inline int FlattenContext::indexOf(const Material* mat)
{
    if (!mat) {
        error = "a hitable without material is used";
        return -2;
    }
    auto it = materialIndex.find(mat);
    if (it != materialIndex.end())
        return it->second;
    wpt_material m;
    if (!mat->describe(m, *this)) {
        if (error.empty())
            error = "a Material subclass that the device path does not know is used";
        return -2;
    }
    int idx = int(materials.size());
    materials.push_back(m);
    materialIndex[mat] = idx;
    return idx;
}

}
