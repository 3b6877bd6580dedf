/*
 * wpt_rgl.h -- the measured-BRDF model behind MaterialRGL (material_rgl.hpp:46-102): Dupuy and
 * Jakob's adaptive parameterisation as the reference evaluates it (powitacq_rgb.inl).
 *
 *   warpSample / warpInvert / warpEval   Marginal2D<Dimension>::sample / invert / eval
 *                                        (powitacq_rgb.inl:322-560) over the flat tables that the
 *                                        host builds (include/wurblpt/rgl.hpp)
 *   rglSample / rglEval / rglPdf         BRDF::sample / eval / pdf (powitacq_rgb.inl:1010-1185)
 *
 * Operation order, the explicit fused multiply-adds of the bilinear look-ups and the comparison
 * forms of min/max/clamp follow the reference line by line.  The code is written once and
 * compiled for the device (math policy: wpt_math.h) and by the test oracle (math policy: its
 * portable or libm back end), which is pinned against golden vectors produced by the reference's
 * own implementation (oracle/ref_probe.cpp).
 *
 * The model and the order of its operations are those of powitacq_rgb (Jonathan Dupuy and Wenzel Jakob, "An Adaptive
 * Parameterization for Efficient Material Acquisition and Rendering"), which is distributed under the 3-clause BSD
 * licence: Copyright 2018 Jonathan Dupuy and Wenzel Jakob.  Redistribution and use in source and binary forms, with or
 * without modification, are permitted provided that the conditions of that licence are met; its full text, with the
 * disclaimer, is reproduced in the LICENSE file of this repository.
 */
#ifndef WPT_RGL_H
#define WPT_RGL_H

#include <stdint.h>

#include "../../include/wurblpt_hip.h"

#if defined(__HIPCC__)
#define WPT_RGL_HD __host__ __device__ __forceinline__
/* the model's entry points for the kernels: real calls (the counting and moving-scene kernels, where few lanes of a mixed wave run
 * this long code), or inlined where a translation unit asks for it (WPT_RGL_INLINE: the product kernels for measured BRDFs --
 * single kernel 101.1 -> 104.7 Msamples/s on the Bistro-class frame at 16 spp; the wavefront form's shade kernel with the
 * interleaved table 120.7 as calls, 129.1 inlined, against 125.2 before the table: profiles/r04_measured_brdf_table.txt) */
#ifdef WPT_RGL_INLINE
#define WPT_RGL_ENTRY __host__ __device__ __forceinline__
#else
#define WPT_RGL_ENTRY __host__ __device__ __attribute__((noinline))
#endif
#define WPT_RGL_UNROLL _Pragma("unroll")
#else
#define WPT_RGL_HD inline
#define WPT_RGL_ENTRY inline
#define WPT_RGL_UNROLL
#endif

namespace wptrgl {

constexpr float k_rgl_pi = 3.1415926535897932384626433832795f;
constexpr float k_oneMinusEpsilon = 0.999999940395355225f;

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};

/* std::min / std::max / clamp of the reference (powitacq_rgb.inl:66-68): comparison forms */
WPT_RGL_HD float rmax(float a, float b) { return (a < b) ? b : a; }
WPT_RGL_HD float rmin(float a, float b) { return (b < a) ? b : a; }
WPT_RGL_HD float rclamp(float v, float lo, float hi) { return rmin(rmax(v, lo), hi); }
WPT_RGL_HD uint32_t umin(uint32_t a, uint32_t b) { return (b < a) ? b : a; }
WPT_RGL_HD float sqr(float v) { return v * v; }

/* find_interval (powitacq_rgb.inl:139-158) */
template<typename Predicate>
WPT_RGL_HD uint32_t findInterval(uint32_t size_, const Predicate& pred)
{
    int64_t size = (int64_t)size_ - 2, first = 1;
    while (size > 0) {
        const int64_t half = size >> 1, middle = first + half;
        const bool predResult = pred((uint32_t)middle);
        first = predResult ? middle + 1 : first;
        size = predResult ? size - (half + 1) : half;
    }
    int64_t r = first - 1;
    const int64_t hi = (int64_t)size_ - 2;
    r = r < 0 ? 0 : r; /* clamp: min(max(v, lo), hi) */
    r = hi < r ? hi : r;
    return (uint32_t)r;
}

/* lookup<Dim> (powitacq_rgb.inl:563-581): multilinear interpolation over the parameter slices.  ES: the table's values lie ES floats
 * apart (1: the reference's arrays; 4: the interleaved colour + luminance table, RglIncident::rgbl) */
template<int Dim, int ES = 1>
struct Lookup {
    static WPT_RGL_HD float at(const float* data, uint32_t i0, uint32_t size, const float* pw, const wpt_rgl_warp& w)
    {
        const uint32_t i1 = i0 + w.param_stride[Dim - 1] * size;
        const float w0 = pw[2 * Dim - 2], w1 = pw[2 * Dim - 1];
        const float v0 = Lookup<Dim - 1, ES>::at(data, i0, size, pw, w);
        const float v1 = Lookup<Dim - 1, ES>::at(data, i1, size, pw, w);
        return __builtin_fmaf(v0, w0, v1 * w1);
    }
};
template<int ES>
struct Lookup<0, ES> {
    static WPT_RGL_HD float at(const float* data, uint32_t index, uint32_t, const float*, const wpt_rgl_warp&) { return data[(size_t)index * ES]; }
};

/* find_interval over a parameter grid of at most 16 values, read all at once: the bisection then runs over the sixteen
 * comparison results instead of asking memory for one value per step (each step of the original is a dependent load, and
 * the grids of the database's files have 8 values).  The same comparisons decide the same steps, so the index is the one
 * find_interval gives for ANY contents of the grid, sorted or not; the two values around it come from the registers. */
WPT_RGL_HD uint32_t findIntervalSmall(const float* values, uint32_t size_, float p, float& p0, float& p1)
{
    float v[16];
    uint32_t predMask = 0;
    WPT_RGL_UNROLL
    for (uint32_t i = 0; i < 16; i++) {
        v[i] = i < size_ ? values[i] : 0.0f;
        predMask |= (v[i] <= p ? 1u : 0u) << i;
    }
    int64_t size = (int64_t)size_ - 2, first = 1;
    while (size > 0) {
        const int64_t half = size >> 1, middle = first + half;
        const bool predResult = ((predMask >> (uint32_t)middle) & 1u) != 0;
        first = predResult ? middle + 1 : first;
        size = predResult ? size - (half + 1) : half;
    }
    int64_t r = first - 1;
    const int64_t hi = (int64_t)size_ - 2;
    r = r < 0 ? 0 : r;
    r = hi < r ? hi : r;
    p0 = v[0];
    p1 = v[1];
    WPT_RGL_UNROLL
    for (uint32_t i = 1; i < 15; i++) {
        if ((uint32_t)r == i) {
            p0 = v[i];
            p1 = v[i + 1];
        }
    }
    return (uint32_t)r;
}

/* one parameter's interpolation weights and its part of the slice offset (the body of the loop at the head of
 * Marginal2D::sample / invert / eval, powitacq_rgb.inl:330-349) */
WPT_RGL_HD uint32_t paramWeightsOf(const wpt_rgl_warp& w, const float* pool, int dim, float p, float& w0, float& w1)
{
    if (w.param_size[dim] == 1) {
        w0 = 1.0f;
        w1 = 0.0f;
        return 0;
    }
    const float* values = pool + w.param_values[dim];
    uint32_t index;
    float p0, p1;
    if (w.param_size[dim] <= 16) {
        index = findIntervalSmall(values, w.param_size[dim], p, p0, p1);
    } else {
        index = findInterval(w.param_size[dim], [&](uint32_t idx) { return values[idx] <= p; });
        p0 = values[index];
        p1 = values[index + 1];
    }
    w1 = rclamp((p - p0) / (p1 - p0), 0.0f, 1.0f);
    w0 = 1.0f - w1;
    return w.param_stride[dim] * index;
}

/* parameter-related indices and weights (the common head of sample / invert / eval): they depend on the warp and the
 * parameters alone, so one evaluation serves every look-up into that warp with those parameters */
template<int Dim>
struct ParamCtx {
    float pw[2 * (Dim > 0 ? Dim : 1)];
    uint32_t sliceOffset;
};
template<int Dim>
WPT_RGL_HD ParamCtx<Dim> paramCtx(const wpt_rgl_warp& w, const float* pool, const float* param)
{
    ParamCtx<Dim> c;
    c.sliceOffset = 0;
    for (int dim = 0; dim < Dim; ++dim)
        c.sliceOffset += paramWeightsOf(w, pool, dim, param[dim], c.pw[2 * dim], c.pw[2 * dim + 1]);
    return c;
}
/* a third parameter behind two that are already evaluated (the colour channel of the spectral interpolant) */
WPT_RGL_HD ParamCtx<3> paramCtxAppend(const ParamCtx<2>& first, const wpt_rgl_warp& w, const float* pool, float third)
{
    ParamCtx<3> c;
    c.pw[0] = first.pw[0];
    c.pw[1] = first.pw[1];
    c.pw[2] = first.pw[2];
    c.pw[3] = first.pw[3];
    c.sliceOffset = first.sliceOffset + paramWeightsOf(w, pool, 2, third, c.pw[4], c.pw[5]);
    return c;
}

/* Marginal2D::sample (powitacq_rgb.inl:322-432) */
template<int Dim, int ES = 1>
WPT_RGL_HD V2 warpSample(const wpt_rgl_warp& w, const float* pool, V2 sample, const ParamCtx<Dim>& ctx, float& pdf, const float* dataAt = nullptr)
{
    sample.x = rclamp(sample.x, 1.0f - k_oneMinusEpsilon, k_oneMinusEpsilon);
    sample.y = rclamp(sample.y, 1.0f - k_oneMinusEpsilon, k_oneMinusEpsilon);
    const float* pw = ctx.pw;
    const uint32_t sliceOffset = ctx.sliceOffset;
    const float* marginal = pool + w.marginal_cdf;
    const float* conditional = pool + w.conditional_cdf;
    const float* data = ES == 1 ? pool + w.data : dataAt; /* (the interleaved table holds the same values ES floats apart) */

    /* the row first */
    uint32_t offset = 0;
    if (Dim != 0)
        offset = sliceOffset * w.size_y;
    auto fetchMarginal = [&](uint32_t idx) { return Lookup<Dim>::at(marginal, offset + idx, w.size_y, pw, w); };
    const uint32_t row = findInterval(w.size_y, [&](uint32_t idx) { return fetchMarginal(idx) < sample.y; });
    sample.y -= fetchMarginal(row);

    const uint32_t sliceSize = w.size_x * w.size_y;
    offset = row * w.size_x;
    if (Dim != 0)
        offset += sliceOffset * sliceSize;
    const float r0 = Lookup<Dim>::at(conditional, offset + w.size_x - 1, sliceSize, pw, w);
    const float r1 = Lookup<Dim>::at(conditional, offset + (w.size_x * 2 - 1), sliceSize, pw, w);
    bool isConst = __builtin_fabsf(r0 - r1) < 1e-4f * (r0 + r1);
    sample.y = isConst ? (2.0f * sample.y) : (r0 - __builtin_sqrtf(r0 * r0 - 2.0f * sample.y * (r0 - r1)));
    sample.y /= isConst ? (r0 + r1) : (r0 - r1);

    /* the column next */
    sample.x *= (1.0f - sample.y) * r0 + sample.y * r1;
    auto fetchConditional = [&](uint32_t idx) {
        const float v0 = Lookup<Dim>::at(conditional, offset + idx, sliceSize, pw, w);
        const float v1 = Lookup<Dim>::at(conditional + w.size_x, offset + idx, sliceSize, pw, w);
        return (1.0f - sample.y) * v0 + sample.y * v1;
    };
    const uint32_t col = findInterval(w.size_x, [&](uint32_t idx) { return fetchConditional(idx) < sample.x; });
    sample.x -= fetchConditional(col);
    offset += col;

    const float v00 = Lookup<Dim, ES>::at(data, offset, sliceSize, pw, w);
    const float v10 = Lookup<Dim, ES>::at(data + ES, offset, sliceSize, pw, w);
    const float v01 = Lookup<Dim, ES>::at(data + ES * w.size_x, offset, sliceSize, pw, w);
    const float v11 = Lookup<Dim, ES>::at(data + ES * (w.size_x + 1), offset, sliceSize, pw, w);
    const float c0 = __builtin_fmaf((1.0f - sample.y), v00, sample.y * v01);
    const float c1 = __builtin_fmaf((1.0f - sample.y), v10, sample.y * v11);
    isConst = __builtin_fabsf(c0 - c1) < 1e-4f * (c0 + c1);
    sample.x = isConst ? (2.0f * sample.x) : (c0 - __builtin_sqrtf(c0 * c0 - 2.0f * sample.x * (c0 - c1)));
    sample.x /= isConst ? (c0 + c1) : (c0 - c1);

    V2 r;
    r.x = ((float)col + sample.x) * w.patch_size[0];
    r.y = ((float)row + sample.y) * w.patch_size[1];
    pdf = ((1.0f - sample.x) * c0 + sample.x * c1) * (w.inv_patch_size[0] * w.inv_patch_size[1]);
    return r;
}

/* Marginal2D::invert (powitacq_rgb.inl:435-514) */
template<int Dim>
WPT_RGL_HD V2 warpSample(const wpt_rgl_warp& w, const float* pool, V2 sample, const float* param, float& pdf)
{
    return warpSample<Dim>(w, pool, sample, paramCtx<Dim>(w, pool, param), pdf);
}

template<int Dim>
WPT_RGL_HD V2 warpInvert(const wpt_rgl_warp& w, const float* pool, V2 sample, const ParamCtx<Dim>& ctx, float& pdfOut)
{
    const float* pw = ctx.pw;
    const uint32_t sliceOffset = ctx.sliceOffset;
    const float* marginal = pool + w.marginal_cdf;
    const float* conditional = pool + w.conditional_cdf;
    const float* data = pool + w.data;

    sample.x *= w.inv_patch_size[0];
    sample.y *= w.inv_patch_size[1];
    const uint32_t posX = umin((uint32_t)sample.x, w.size_x - 2u);
    const uint32_t posY = umin((uint32_t)sample.y, w.size_y - 2u);
    sample.x -= (float)(int32_t)posX;
    sample.y -= (float)(int32_t)posY;

    uint32_t offset = posX + posY * w.size_x;
    const uint32_t sliceSize = w.size_x * w.size_y;
    if (Dim != 0)
        offset += sliceOffset * sliceSize;

    const float v00 = Lookup<Dim>::at(data, offset, sliceSize, pw, w);
    const float v10 = Lookup<Dim>::at(data + 1, offset, sliceSize, pw, w);
    const float v01 = Lookup<Dim>::at(data + w.size_x, offset, sliceSize, pw, w);
    const float v11 = Lookup<Dim>::at(data + w.size_x + 1, offset, sliceSize, pw, w);
    const float w1x = sample.x, w1y = sample.y, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float c0 = __builtin_fmaf(w0y, v00, w1y * v01);
    const float c1 = __builtin_fmaf(w0y, v10, w1y * v11);
    const float pdf = __builtin_fmaf(w0x, c0, w1x * c1);

    sample.x *= c0 + 0.5f * sample.x * (c1 - c0);
    const float v0 = Lookup<Dim>::at(conditional, offset, sliceSize, pw, w);
    const float v1 = Lookup<Dim>::at(conditional + w.size_x, offset, sliceSize, pw, w);
    sample.x += (1.0f - sample.y) * v0 + sample.y * v1;

    offset = posY * w.size_x;
    if (Dim != 0)
        offset += sliceOffset * sliceSize;
    const float r0 = Lookup<Dim>::at(conditional, offset + w.size_x - 1, sliceSize, pw, w);
    const float r1 = Lookup<Dim>::at(conditional, offset + (w.size_x * 2 - 1), sliceSize, pw, w);
    sample.x /= (1.0f - sample.y) * r0 + sample.y * r1;

    sample.y *= r0 + 0.5f * sample.y * (r1 - r0);
    offset = posY;
    if (Dim != 0)
        offset += sliceOffset * w.size_y;
    sample.y += Lookup<Dim>::at(marginal, offset, w.size_y, pw, w);

    pdfOut = pdf * (w.inv_patch_size[0] * w.inv_patch_size[1]);
    return sample;
}

/* Marginal2D::eval (powitacq_rgb.inl:520-560) */
template<int Dim>
WPT_RGL_HD V2 warpInvert(const wpt_rgl_warp& w, const float* pool, V2 sample, const float* param, float& pdfOut)
{
    return warpInvert<Dim>(w, pool, sample, paramCtx<Dim>(w, pool, param), pdfOut);
}

template<int Dim, int ES = 1>
WPT_RGL_HD float warpEval(const wpt_rgl_warp& w, const float* pool, V2 pos, const ParamCtx<Dim>& ctx, const float* dataAt = nullptr)
{
    const float* pw = ctx.pw;
    const uint32_t sliceOffset = ctx.sliceOffset;
    const float* data = ES == 1 ? pool + w.data : dataAt;
    pos.x *= w.inv_patch_size[0];
    pos.y *= w.inv_patch_size[1];
    const uint32_t ox = umin((uint32_t)pos.x, w.size_x - 2u);
    const uint32_t oy = umin((uint32_t)pos.y, w.size_y - 2u);
    const float w1x = pos.x - (float)(int32_t)ox, w1y = pos.y - (float)(int32_t)oy;
    const float w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    uint32_t index = ox + oy * w.size_x;
    const uint32_t size = w.size_x * w.size_y;
    if (Dim != 0)
        index += sliceOffset * size;
    const float v00 = Lookup<Dim, ES>::at(data, index, size, pw, w);
    const float v10 = Lookup<Dim, ES>::at(data + ES, index, size, pw, w);
    const float v01 = Lookup<Dim, ES>::at(data + ES * w.size_x, index, size, pw, w);
    const float v11 = Lookup<Dim, ES>::at(data + ES * (w.size_x + 1), index, size, pw, w);
    return __builtin_fmaf(w0y, __builtin_fmaf(w0x, v00, w1x * v10), w1y * __builtin_fmaf(w0x, v01, w1x * v11))
        * (w.inv_patch_size[0] * w.inv_patch_size[1]);
}
template<int Dim>
WPT_RGL_HD float warpEval(const wpt_rgl_warp& w, const float* pool, V2 pos, const float* param)
{
    return warpEval<Dim>(w, pool, pos, paramCtx<Dim>(w, pool, param));
}

/* BRDF convenience functions (powitacq_rgb.inl:870-884) */
WPT_RGL_HD float u2theta(float u) { return sqr(u) * (k_rgl_pi / 2.0f); }
WPT_RGL_HD float u2phi(float u) { return (2.0f * u - 1.0f) * k_rgl_pi; }
WPT_RGL_HD float phi2u(float phi) { return (phi + k_rgl_pi) / (2.0f * k_rgl_pi); }
template<class M> WPT_RGL_HD float theta2u(float theta) { return __builtin_sqrtf(theta * (2.0f / k_rgl_pi)); }
/* elevation (powitacq_rgb.inl:1012-1014).  The reference calls the unqualified `asin`, which for
 * a float argument is the C library's double function there, and multiplies by 2.f in double
 * before the result is rounded to float once: M::twiceAsin(x) = float(2.0 * asin(double(x))). */
template<class M> WPT_RGL_HD float elevation(V3 d)
{
    return M::twiceAsin(0.5f * __builtin_sqrtf(sqr(d.x) + sqr(d.y) + sqr(d.z - 1.0f)));
}
WPT_RGL_HD float dot3(V3 a, V3 b)
{
    float r = 0.0f;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
WPT_RGL_HD V3 normalize3(V3 v)
{
    const float l = __builtin_sqrtf(dot3(v, v));
    V3 r;
    r.x = v.x / l;
    r.y = v.y / l;
    r.z = v.z / l;
    return r;
}

/* the three colour channels of the spectral interpolant, clipped (POWITACQ_CLIP_RGB); `first` = the incident direction's
 * two parameters evaluated on the warp b.rgb */
/* The same from the interleaved table (RglIncident::rgbl: red, green, blue, luminance of one grid point side by side, 16 bytes):
 * warpEval<3> on b.rgb reads, for every corner of the patch, the channel's own slice and its neighbour's -- 2 x 4 parameter taps
 * x 4 corners x 3 channels = 96 loads from 24 lines -- although the three channels interpolate the SAME three two-parameter values
 * per corner (the channel parameter is 0, 1 or 2: one of its two weights is exactly 0, the value still takes part as the reference
 * computes it).  Here each corner's three two-parameter values are evaluated once (the operations of Lookup<2> inside Lookup<3>, on
 * the same numbers) and every channel combines its two with its own weights: 48 loads from 8 lines, the same bits. */
WPT_RGL_HD bool rglInterleavable(const wpt_rgl_brdf& b)
{
    const wpt_rgl_warp& c = b.rgb;
    const wpt_rgl_warp& l = b.luminance;
    return c.dims == 3 && l.dims == 2 && c.size_x == l.size_x && c.size_y == l.size_y && c.param_size[2] == 3 && c.param_stride[2] == 1
        && c.param_size[0] == l.param_size[0] && c.param_size[1] == l.param_size[1]
        && c.param_stride[0] == 3 * l.param_stride[0] && c.param_stride[1] == 3 * l.param_stride[1];
}
struct alignas(16) RglQuad { /* one record of the interleaved table */
    float r, g, b, lum;
};
/* the colour at `pos` and, with LUM, the luminance warp's value there (BRDF::eval and BRDF::pdf ask for both at one position):
 * sixteen 16-byte loads -- four parameter taps for each of the patch's four corners -- serve all four values */
template<bool LUM>
WPT_RGL_HD V3 rglColourInterleaved(const wpt_rgl_brdf& b, const float* pool, const float* table, V2 pos, const ParamCtx<2>& first,
        const ParamCtx<2>& lumCtx, float& lumOut)
{
    const wpt_rgl_warp& w = b.rgb;
    pos.x *= w.inv_patch_size[0];
    pos.y *= w.inv_patch_size[1];
    const uint32_t ox = umin((uint32_t)pos.x, w.size_x - 2u);
    const uint32_t oy = umin((uint32_t)pos.y, w.size_y - 2u);
    const float w1x = pos.x - (float)(int32_t)ox, w1y = pos.y - (float)(int32_t)oy;
    const float w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const uint32_t size = w.size_x * w.size_y;
    const RglQuad* quads = reinterpret_cast<const RglQuad*>(table);
    /* the slice of (phi_i, theta_i) in the interleaved table: the colour warp's slice offset without its channel, over 3; the
     * luminance warp's own slice offset (the same number where the two grids hold the same values) */
    const uint32_t at = ox + oy * w.size_x;
    const uint32_t index = at + (first.sliceOffset / 3u) * size, indexLum = at + lumCtx.sliceOffset * size;
    const uint32_t s0 = b.luminance.param_stride[0] * size, s1 = b.luminance.param_stride[1] * size;
    /* every channel's weights and slice along the third parameter (0, 1, 2: one of the two weights is exactly 0) */
    float cw0[3], cw1[3];
    uint32_t ch[3];
    WPT_RGL_UNROLL
    for (int i = 0; i < 3; ++i)
        ch[i] = paramWeightsOf(w, pool, 2, (float)i, cw0[i], cw1[i]);
    const float* pw = first.pw;
    const float* lw = lumCtx.pw;
    float rows[2][4]; /* per row of the patch: fma(w0x, left corner, w1x * right corner) of red, green, blue, luminance */
    WPT_RGL_UNROLL
    for (int row = 0; row < 2; row++) {
        float v[2][4];
        WPT_RGL_UNROLL
        for (int col = 0; col < 2; col++) {
            const uint32_t corner = (uint32_t)col + (uint32_t)row * w.size_x;
            const RglQuad q00 = quads[index + corner], q10 = quads[index + corner + s0], q01 = quads[index + corner + s1], q11 = quads[index + corner + s0 + s1];
            /* Lookup<2> of the three channel slices (the operations of Lookup<2> inside Lookup<3>, on the same numbers) */
            const float two[3] = {
                __builtin_fmaf(__builtin_fmaf(q00.r, pw[0], q10.r * pw[1]), pw[2], __builtin_fmaf(q01.r, pw[0], q11.r * pw[1]) * pw[3]),
                __builtin_fmaf(__builtin_fmaf(q00.g, pw[0], q10.g * pw[1]), pw[2], __builtin_fmaf(q01.g, pw[0], q11.g * pw[1]) * pw[3]),
                __builtin_fmaf(__builtin_fmaf(q00.b, pw[0], q10.b * pw[1]), pw[2], __builtin_fmaf(q01.b, pw[0], q11.b * pw[1]) * pw[3]) };
            WPT_RGL_UNROLL
            for (int i = 0; i < 3; ++i) {
                const float a = ch[i] == 0 ? two[0] : two[1], bb = ch[i] == 0 ? two[1] : two[2];
                v[col][i] = __builtin_fmaf(a, cw0[i], bb * cw1[i]);
            }
            if (LUM) {
                const RglQuad l00 = index == indexLum ? q00 : quads[indexLum + corner], l10 = index == indexLum ? q10 : quads[indexLum + corner + s0],
                              l01 = index == indexLum ? q01 : quads[indexLum + corner + s1], l11 = index == indexLum ? q11 : quads[indexLum + corner + s0 + s1];
                v[col][3] = __builtin_fmaf(__builtin_fmaf(l00.lum, lw[0], l10.lum * lw[1]), lw[2], __builtin_fmaf(l01.lum, lw[0], l11.lum * lw[1]) * lw[3]);
            } else {
                v[col][3] = 0.0f;
            }
        }
        WPT_RGL_UNROLL
        for (int i = 0; i < 4; ++i)
            rows[row][i] = __builtin_fmaf(w0x, v[0][i], w1x * v[1][i]);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory"); /* the second row's loads stay behind the first row's arithmetic: registers */
#endif
    }
    float fr[3];
    WPT_RGL_UNROLL
    for (int i = 0; i < 3; ++i) {
        fr[i] = __builtin_fmaf(w0y, rows[0][i], w1y * rows[1][i]) * (w.inv_patch_size[0] * w.inv_patch_size[1]);
        fr[i] = rmax(0.0f, fr[i]);
    }
    if (LUM)
        lumOut = __builtin_fmaf(w0y, rows[0][3], w1y * rows[1][3]) * (b.luminance.inv_patch_size[0] * b.luminance.inv_patch_size[1]);
    V3 r;
    r.x = fr[0];
    r.y = fr[1];
    r.z = fr[2];
    return r;
}

WPT_RGL_HD V3 rglColour(const wpt_rgl_brdf& b, const float* pool, V2 sample, const ParamCtx<2>& first)
{
    float fr[3];
    for (int i = 0; i < 3; ++i) {
        fr[i] = warpEval<3>(b.rgb, pool, sample, paramCtxAppend(first, b.rgb, pool, (float)i));
        fr[i] = rmax(0.0f, fr[i]);
    }
    V3 r;
    r.x = fr[0];
    r.y = fr[1];
    r.z = fr[2];
    return r;
}

/* Everything BRDF::sample, eval and pdf derive from the INCIDENT direction alone (powitacq_rgb.inl:1016-1183): its
 * angles, its place in the unit square, the parameter weights of the three warps that are parameterised by it, and the
 * projected-area term.  A path evaluates the model up to three times at one hit with one incident direction (sample,
 * then eval and pdf towards the light): this is evaluated once for them. */
struct RglIncident {
    float theta_i, phi_i;
    V2 u_wi;
    ParamCtx<2> vndf, luminance, rgb;
    float d; /* 4 * sigma(u_wi) */
    /* offset of this BRDF's interleaved colour + luminance table in the pool, or WPT_RGL_NONE: the device's copy of the pool has
     * one per BRDF whose two warps share their grids (wpt_capi.hip builds it at upload; the test oracle has none) */
    uint32_t rgbl;
};
template<class M>
WPT_RGL_HD RglIncident rglIncident(const wpt_rgl_brdf& b, const float* pool, V3 wi, uint32_t rgbl = WPT_RGL_NONE)
{
    RglIncident inc;
    inc.rgbl = rgbl;
    inc.theta_i = elevation<M>(wi);
    inc.phi_i = M::atan2(wi.y, wi.x);
    inc.u_wi.x = theta2u<M>(inc.theta_i);
    inc.u_wi.y = phi2u(inc.phi_i);
    const float params[2] = { inc.phi_i, inc.theta_i };
    inc.vndf = paramCtx<2>(b.vndf, pool, params);
    inc.luminance = paramCtx<2>(b.luminance, pool, params);
    inc.rgb = paramCtx<2>(b.rgb, pool, params);
    inc.d = 4 * warpEval<0>(b.sigma, pool, inc.u_wi, params);
    return inc;
}

/* BRDF::eval (powitacq_rgb.inl:1056-1100: f_r * cos) and BRDF::pdf (:1016-1050) for one pair of directions: they share
 * the half vector, its angles and the inverted sample position, which are evaluated once here */
template<class M>
WPT_RGL_HD void rglEvalPdfWith(const wpt_rgl_brdf& b, const float* pool, const RglIncident& inc, V3 wi, V3 wo, V3& frOut, float& pdfOut)
{
    V3 zero;
    zero.x = zero.y = zero.z = 0.0f;
    frOut = zero;
    pdfOut = 0.0f;
    if (wi.z <= 0 || wo.z <= 0)
        return;
    V3 s;
    s.x = wi.x + wo.x;
    s.y = wi.y + wo.y;
    s.z = wi.z + wo.z;
    const V3 wm = normalize3(s);
    const float theta_m = elevation<M>(wm), phi_m = M::atan2(wm.y, wm.x);
    V2 u_wm;
    u_wm.x = theta2u<M>(theta_m);
    u_wm.y = phi2u(b.isotropic ? (phi_m - inc.phi_i) : phi_m);
    u_wm.y = u_wm.y - __builtin_floorf(u_wm.y);
    float vndfPdf;
    const V2 sample = warpInvert<2>(b.vndf, pool, u_wm, inc.vndf, vndfPdf);
    /* eval */
    const bool interleaved = inc.rgbl != WPT_RGL_NONE;
    float pdf = 0.0f; /* BRDF::pdf's luminance term: from the interleaved table it comes with the colour */
    V3 fr = interleaved ? rglColourInterleaved<true>(b, pool, pool + inc.rgbl, sample, inc.rgb, inc.luminance, pdf) : rglColour(b, pool, sample, inc.rgb);
    const float params[2] = { inc.phi_i, inc.theta_i };
    const float n = warpEval<0>(b.ndf, pool, u_wm, params);
    const float d = inc.d;
    fr.x = fr.x * n / d;
    fr.y = fr.y * n / d;
    fr.z = fr.z * n / d;
    frOut = fr;
    /* pdf */
    if (!interleaved)
        pdf = warpEval<2>(b.luminance, pool, sample, inc.luminance);
    const float sinThetaM = __builtin_sqrtf(sqr(wm.x) + sqr(wm.y));
    const float jacobian = rmax(2.0f * sqr(k_rgl_pi) * u_wm.x * sinThetaM, 1e-6f) * 4.0f * dot3(wi, wm);
    pdfOut = vndfPdf * pdf / jacobian;
}

/* BRDF::sample (powitacq_rgb.inl:1106-1183): returns f_r * cos / pdf, the outgoing direction
 * (zero when the sample fails) and the pdf */
template<class M>
WPT_RGL_HD V3 rglSampleWith(const wpt_rgl_brdf& b, const float* pool, const RglIncident& inc, V2 u, V3 wi, V3& woOut, float& pdfOut)
{
    V3 zero;
    zero.x = zero.y = zero.z = 0.0f;
    woOut = zero;
    pdfOut = 0.0f;
    if (wi.z <= 0)
        return zero;
    V2 sample;
    sample.x = u.y;
    sample.y = u.x;
    float lumPdf;
    const bool interleaved = inc.rgbl != WPT_RGL_NONE;
    sample = interleaved ? warpSample<2, 4>(b.luminance, pool, sample, inc.luminance, lumPdf, pool + inc.rgbl + 3) : warpSample<2>(b.luminance, pool, sample, inc.luminance, lumPdf);
    float ndfPdf;
    const V2 u_wm = warpSample<2>(b.vndf, pool, sample, inc.vndf, ndfPdf);
    float phi_m = u2phi(u_wm.y);
    const float theta_m = u2theta(u_wm.x);
    if (b.isotropic)
        phi_m += inc.phi_i;
    const float sinPhiM = M::sin(phi_m), cosPhiM = M::cos(phi_m), sinThetaM = M::sin(theta_m), cosThetaM = M::cos(theta_m);
    V3 wm;
    wm.x = cosPhiM * sinThetaM;
    wm.y = sinPhiM * sinThetaM;
    wm.z = cosThetaM;
    const float dwm = dot3(wm, wi);
    V3 wo;
    wo.x = wm.x * 2.0f * dwm - wi.x;
    wo.y = wm.y * 2.0f * dwm - wi.y;
    wo.z = wm.z * 2.0f * dwm - wi.z;
    if (wo.z <= 0)
        return zero;
    float unused = 0.0f;
    V3 fr = interleaved ? rglColourInterleaved<false>(b, pool, pool + inc.rgbl, sample, inc.rgb, inc.luminance, unused) : rglColour(b, pool, sample, inc.rgb);
    const float params[2] = { inc.phi_i, inc.theta_i };
    const float n = warpEval<0>(b.ndf, pool, u_wm, params);
    const float d = inc.d;
    fr.x = fr.x * n / d;
    fr.y = fr.y * n / d;
    fr.z = fr.z * n / d;
    const float jacobian = rmax(2.0f * sqr(k_rgl_pi) * u_wm.x * sinThetaM, 1e-6f) * 4.0f * dot3(wi, wm);
    const float pdf = ndfPdf * lumPdf / jacobian;
    woOut = wo;
    pdfOut = pdf;
    fr.x /= pdf;
    fr.y /= pdf;
    fr.z /= pdf;
    return fr;
}

/* the reference's three entry points, as its callers use them (the test oracle; ref_probe's golden vectors pin them) */
template<class M>
WPT_RGL_ENTRY float rglPdf(const wpt_rgl_brdf& b, const float* pool, V3 wi, V3 wo)
{
    if (wi.z <= 0 || wo.z <= 0)
        return 0.0f;
    V3 fr;
    float pdf;
    rglEvalPdfWith<M>(b, pool, rglIncident<M>(b, pool, wi), wi, wo, fr, pdf);
    return pdf;
}
template<class M>
WPT_RGL_ENTRY V3 rglEval(const wpt_rgl_brdf& b, const float* pool, V3 wi, V3 wo)
{
    V3 fr;
    fr.x = fr.y = fr.z = 0.0f;
    if (wi.z <= 0 || wo.z <= 0)
        return fr;
    float pdf;
    rglEvalPdfWith<M>(b, pool, rglIncident<M>(b, pool, wi), wi, wo, fr, pdf);
    return fr;
}
template<class M>
WPT_RGL_ENTRY V3 rglSample(const wpt_rgl_brdf& b, const float* pool, V2 u, V3 wi, V3& woOut, float& pdfOut)
{
    V3 zero;
    zero.x = zero.y = zero.z = 0.0f;
    woOut = zero;
    pdfOut = 0.0f;
    if (wi.z <= 0)
        return zero;
    return rglSampleWith<M>(b, pool, rglIncident<M>(b, pool, wi), u, wi, woOut, pdfOut);
}

/* the same for the kernels: out of line (few lanes run them, and they are long), the incident direction's context by
 * reference so that scatter and the evaluation towards the light share one */
template<class M>
WPT_RGL_ENTRY void rglIncidentCall(const wpt_rgl_brdf& b, const float* pool, V3 wi, RglIncident& inc, uint32_t rgbl = WPT_RGL_NONE)
{
    inc = rglIncident<M>(b, pool, wi, rgbl);
}
template<class M>
WPT_RGL_ENTRY V3 rglSampleCall(const wpt_rgl_brdf& b, const float* pool, const RglIncident& inc, V2 u, V3 wi, V3& woOut, float& pdfOut)
{
    return rglSampleWith<M>(b, pool, inc, u, wi, woOut, pdfOut);
}
template<class M>
WPT_RGL_ENTRY void rglEvalPdfCall(const wpt_rgl_brdf& b, const float* pool, const RglIncident& inc, V3 wi, V3 wo, V3& frOut, float& pdfOut)
{
    rglEvalPdfWith<M>(b, pool, inc, wi, wo, frOut, pdfOut);
}

} /* namespace wptrgl */

#endif
