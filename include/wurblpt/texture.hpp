/*
 * texture.hpp -- host-side texture descriptions (reference texture.hpp:45-246,
 * texture_image.hpp:39-366).  The reference evaluates textures through a virtual
 * `value(texcoords, t)`; here a texture is a *description* that the flattener turns into a
 * wpt_texture record (include/wurblpt_hip.h) which the HIP kernel evaluates.  A user
 * subclass that the kernel does not know makes `describe()` return false and mcpt() refuses
 * the scene loudly.
 */
#pragma once

#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../wurblpt_hip.h"
#include "array.hpp"
#include "gvm.hpp"
#include "imageio.hpp"
#include "scene_component.hpp"

namespace WurblPT {

enum LinearizeSRGBType { LinearizeSRGB_On, LinearizeSRGB_Off, LinearizeSRGB_Auto };

class Texture;
class Material;

/* Bookkeeping while a Scene is flattened: index assignment and the texel pool. */
class FlattenContext
{
public:
    std::map<const Texture*, int> textureIndex;
    std::vector<wpt_texture> textures;
    std::vector<unsigned char> texels;
    std::map<const Material*, int> materialIndex;
    std::vector<wpt_material> materials;
    std::vector<wpt_rgl_brdf> rglBrdfs; /* measured BRDFs (MaterialRGL) and their tables */
    std::vector<float> rglData;
    std::string error;

    int indexOf(const Texture* tex);   /* -1 for nullptr; -2 on unsupported */
    int indexOf(const Material* mat);  /* -2 on unsupported */
};

class Texture : public SceneComponent
{
public:
    virtual ~Texture() {}
    /* Fill `out`; return false if this texture type cannot run on the device. */
    virtual bool describe(wpt_texture& /* out */, FlattenContext& /* ctx */) const { return false; }
    virtual vec2 texelSize() const { return vec2(0.0f, 0.0f); }
    virtual unsigned int componentCount() const { return 4; }
};

inline void wptSet(float* dst, const vec4& v) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
inline void wptSet2(float* dst, const vec2& v) { dst[0] = v[0]; dst[1] = v[1]; }

inline wpt_texture wptEmptyTexture()
{
    wpt_texture t;
    memset(&t, 0, sizeof(t));
    t.child = -1;
    t.coord_factor[0] = t.coord_factor[1] = 1.0f;
    t.a[0] = t.a[1] = t.a[2] = t.a[3] = 1.0f;
    return t;
}

class TextureConstant final : public Texture
{
private:
    const vec4 _color;

public:
    TextureConstant(const vec4& color) : _color(color) {}
    virtual bool describe(wpt_texture& out, FlattenContext&) const override
    {
        out = wptEmptyTexture();
        out.type = WPT_TEX_CONSTANT;
        wptSet(out.a, _color);
        return true;
    }
    virtual vec2 texelSize() const override { return vec2(1.0f, 1.0f); }
};

class TextureChecker final : public Texture
{
private:
    vec4 _color0, _color1;
    int _horiz, _vert;

public:
    TextureChecker(const vec4& color0, const vec4& color1, int horiz = 9, int vert = 9) :
        _color0(color0), _color1(color1), _horiz(horiz), _vert(vert)
    {
    }
    TextureChecker(const vec3& color0, const vec3& color1, int horiz = 9, int vert = 9) :
        TextureChecker(vec4(color0, average(color0)), vec4(color1, average(color1)), horiz, vert)
    {
    }
    virtual bool describe(wpt_texture& out, FlattenContext&) const override
    {
        out = wptEmptyTexture();
        out.type = WPT_TEX_CHECKER;
        out.width = _horiz;
        out.height = _vert;
        wptSet(out.a, _color0);
        wptSet(out.b, _color1);
        return true;
    }
};

class TextureTransformer final : public Texture
{
private:
    const Texture* _texture;
    const vec2 _coordFactor, _coordOffset;
    const vec4 _valFactor, _valOffset;

public:
    TextureTransformer(const Texture* texture, const vec2& coordFactor, const vec2& coordOffset = vec2(0.0f),
            const vec4& valFactor = vec4(1.0f), const vec4& valOffset = vec4(0.0f)) :
        _texture(texture), _coordFactor(coordFactor), _coordOffset(coordOffset), _valFactor(valFactor), _valOffset(valOffset)
    {
    }
    virtual bool describe(wpt_texture& out, FlattenContext& ctx) const override
    {
        int child = ctx.indexOf(_texture);
        if (child < 0)
            return false;
        out = wptEmptyTexture();
        out.type = WPT_TEX_TRANSFORMER;
        out.child = child;
        wptSet2(out.coord_factor, _coordFactor);
        wptSet2(out.coord_offset, _coordOffset);
        wptSet(out.a, _valFactor);
        wptSet(out.b, _valOffset);
        return true;
    }
    virtual vec2 texelSize() const override { return _texture->texelSize() / _coordFactor; }
    virtual unsigned int componentCount() const override { return _texture->componentCount(); }
};

/* Image texture: bilinear, fract() wrap on the coordinates, clamp on the taps, sRGB
 * linearization per tap (reference texture_image.hpp:85-212). */
class TextureImage final : public Texture
{
private:
    const ArrayContainer _img;
    const bool _linearizeSRGB;
    const vec2 _coordFactor, _coordOffset;
    const vec4 _valFactor, _valOffset;

public:
    TextureImage(const ArrayContainer& img, bool linearizeSRGB, const vec2& coordFactor = vec2(1.0f),
            const vec2& coordOffset = vec2(0.0f), const vec4& valFactor = vec4(1.0f), const vec4& valOffset = vec4(0.0f)) :
        _img(img), _linearizeSRGB(linearizeSRGB), _coordFactor(coordFactor), _coordOffset(coordOffset),
        _valFactor(valFactor), _valOffset(valOffset)
    {
    }
    virtual bool describe(wpt_texture& out, FlattenContext& ctx) const override
    {
        out = wptEmptyTexture();
        out.type = WPT_TEX_IMAGE;
        out.width = _img.dimension(0);
        out.height = _img.dimension(1);
        out.comps = _img.componentCount();
        out.texel_type = _img.componentType() == uint8 ? WPT_TEXEL_U8 : _img.componentType() == uint16 ? WPT_TEXEL_U16 : WPT_TEXEL_F32;
        out.linearize_srgb = _linearizeSRGB ? 1 : 0;
        size_t offset = (ctx.texels.size() + 15) / 16 * 16;
        ctx.texels.resize(offset + _img.dataSize());
        memcpy(ctx.texels.data() + offset, _img.data(), _img.dataSize());
        out.texel_offset = offset;
        wptSet2(out.coord_factor, _coordFactor);
        wptSet2(out.coord_offset, _coordOffset);
        wptSet(out.a, _valFactor);
        wptSet(out.b, _valOffset);
        return true;
    }
    virtual vec2 texelSize() const override { return vec2(1.0f / _img.dimension(0), 1.0f / _img.dimension(1)); }
    virtual unsigned int componentCount() const override { return _img.componentCount(); }
};

/* Factory with the reference's signature (texture_image.hpp:235-240) */
inline Texture* createTextureImage(const ArrayContainer& img, LinearizeSRGBType linearizeSRGBType = LinearizeSRGB_Auto,
        const vec2& coordFactor = vec2(1.0f), const vec2& coordOffset = vec2(0.0f),
        const vec4& valFactor = vec4(1.0f), const vec4& valOffset = vec4(0.0f))
{
    if (img.dimension(0) == 0 || img.dimension(1) == 0 || img.componentCount() < 1 || img.componentCount() > 4) {
        fprintf(stderr, "createTextureImage: not a valid texture image\n");
        return nullptr;
    }
    bool lin = linearizeSRGBType == LinearizeSRGB_On
        || (linearizeSRGBType == LinearizeSRGB_Auto && (img.componentType() == uint8 || img.componentType() == uint16));
    return new TextureImage(img, lin, coordFactor, coordOffset, valFactor, valOffset);
}

/* texture_image.hpp:343-366: the same from an image file (wurblpt-sponza.cpp:56, wurblpt-envmap.cpp:73); decoded by
 * imageio.hpp (PNG, JPEG, TGA, PNM / PFM, Radiance HDR, OpenEXR).  NULL and a line on stderr when the file cannot be used. */
inline Texture* createTextureImage(const std::string& imgFileName, LinearizeSRGBType linearizeSRGBType = LinearizeSRGB_Auto,
        const vec2& coordFactor = vec2(1.0f), const vec2& coordOffset = vec2(0.0f),
        const vec4& valFactor = vec4(1.0f), const vec4& valOffset = vec4(0.0f))
{
    std::string error;
    const ArrayContainer img = loadImage(imgFileName, &error);
    if (img.elementCount() == 0) {
        fprintf(stderr, "%s\n", error.c_str());
        return nullptr;
    }
    if (img.componentType() != uint8 && img.componentType() != uint16 && img.componentType() != float32) {
        fprintf(stderr, "%s: not a valid texture image\n", imgFileName.c_str());
        return nullptr;
    }
    return createTextureImage(img, linearizeSRGBType, coordFactor, coordOffset, valFactor, valOffset);
}

inline int FlattenContext::indexOf(const Texture* tex)
{
    if (!tex)
        return -1;
    auto it = textureIndex.find(tex);
    if (it != textureIndex.end())
        return it->second;
    wpt_texture t;
    if (!tex->describe(t, *this)) {
        error = "a Texture subclass that the device path does not know is used";
        return -2;
    }
    int idx = int(textures.size());
    textures.push_back(t);
    textureIndex[tex] = idx;
    return idx;
}

}
