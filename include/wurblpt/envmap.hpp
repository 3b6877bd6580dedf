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
        return out.tex >= 0;
    }
};

}
