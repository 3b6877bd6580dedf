/*
 * envmap.hpp -- environment map descriptions (reference envmap.hpp:44-247).
 * The radiance lookup L(), the pdf p() and the sampler d() run in the HIP kernel; the
 * importance tables (envmap.hpp:121-158) are built once by wpt_scene_upload() when
 * initializeImportanceSampling(N) was requested, with the per-bin importance evaluated by the
 * same device code that evaluates L() later.
 */
#pragma once

#include "../wurblpt_hip.h"
#include "texture.hpp"

namespace WurblPT {

class EnvironmentMap
{
protected:
    int N;

public:
    EnvironmentMap() : N(0) {}
    virtual ~EnvironmentMap() {}
    void initializeImportanceSampling(int n) { N = n; }
    bool supportsImportanceSampling() const { return N > 0; }
    virtual bool describe(wpt_envmap& /* out */, FlattenContext& /* ctx */) const { return false; }
};

class EnvironmentMapEquiRect final : public EnvironmentMap
{
public:
    enum Compatibility { CompatibilityMitsuba, CompatibilitySurroundVideo };

private:
    const Compatibility _compatibility;
    const Texture* _tex;

public:
    EnvironmentMapEquiRect(const Texture* tex, Compatibility compat = CompatibilityMitsuba) : _compatibility(compat), _tex(tex) {}
    virtual bool describe(wpt_envmap& out, FlattenContext& ctx) const override
    {
        memset(&out, 0, sizeof(out));
        out.type = WPT_ENV_EQUIRECT;
        out.compat = _compatibility == CompatibilityMitsuba ? WPT_ENV_COMPAT_MITSUBA : WPT_ENV_COMPAT_SURROUND_VIDEO;
        out.tex = ctx.indexOf(_tex);
        out.N = N;
        for (int k = 0; k < 6; k++)
            out.cube_tex[k] = -1;
        return out.tex >= 0;
    }
};

/* envmap.hpp:250-285: six textures, +x -x +y -y +z -z */
class EnvironmentMapCube final : public EnvironmentMap
{
private:
    const Texture* _cubesides[6];

public:
    EnvironmentMapCube(const Texture* posx, const Texture* negx, const Texture* posy, const Texture* negy,
            const Texture* posz, const Texture* negz) : _cubesides { posx, negx, posy, negy, posz, negz } {}
    virtual bool describe(wpt_envmap& out, FlattenContext& ctx) const override
    {
        memset(&out, 0, sizeof(out));
        out.type = WPT_ENV_CUBE;
        out.tex = -1;
        out.N = N;
        for (int k = 0; k < 6; k++) {
            out.cube_tex[k] = ctx.indexOf(_cubesides[k]);
            if (out.cube_tex[k] < 0)
                return false;
        }
        return true;
    }
};

}
