/*
 * wpt_postproc.h -- per-pixel operations of the output side: colour space conversions
 * (color.hpp:226-310) and the tone mapping / quantisation steps of postproc.hpp:44-108
 * (toSRGB, maxLuminance, uniformRationalQuantization, scaleLuminance).
 * Written once; compiled for the device (kernels in wpt_capi.hip) and by the test oracle, whose
 * elementary functions are pinned to the reference's color.hpp (oracle/ref_probe.cpp).
 */
#ifndef WPT_POSTPROC_H
#define WPT_POSTPROC_H

#include <stdint.h>

#if defined(__HIPCC__)
#define WPT_PP_HD __host__ __device__ __forceinline__
#else
#define WPT_PP_HD inline
#endif

namespace wptpp {

struct V3 {
    float x, y, z;
};

/* color.hpp:247-253 */
WPT_PP_HD V3 rgbToXyz(V3 rgb)
{
    V3 r;
    r.x = 100.0f * (0.412453f * rgb.x + 0.357580f * rgb.y + 0.180423f * rgb.z);
    r.y = 100.0f * (0.212671f * rgb.x + 0.715160f * rgb.y + 0.072169f * rgb.z);
    r.z = 100.0f * (0.019334f * rgb.x + 0.119193f * rgb.y + 0.950227f * rgb.z);
    return r;
}

/* color.hpp:255-261 */
WPT_PP_HD V3 xyzToRgb(V3 xyz)
{
    V3 r;
    r.x = 0.01f * (+3.240479f * xyz.x - 1.537150f * xyz.y - 0.498535f * xyz.z);
    r.y = 0.01f * (-0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z);
    r.z = 0.01f * (+0.055648f * xyz.x - 0.204023f * xyz.y + 1.057311f * xyz.z);
    return r;
}

/* color.hpp:226-237: new luminance, old chromaticity */
WPT_PP_HD V3 adjustY(V3 xyz, float newY)
{
    V3 r;
    r.x = r.y = r.z = 0.0f;
    const float sum = xyz.x + xyz.y + xyz.z;
    if (xyz.y <= 0.0f || sum <= 0.0f)
        return r;
    const float x = xyz.x / sum;
    const float y = xyz.y / sum;
    const float f = newY / y;
    r.x = f * x;
    r.y = newY;
    r.z = f * (1.0f - x - y);
    return r;
}

/* color.hpp:265-268; M::pow is the back end's powf */
template<class M> WPT_PP_HD float rgbToSrgbHelper(float x)
{
    return (x <= 0.0031308f ? (x * 12.92f) : (1.055f * M::pow(x, 1.0f / 2.4f) - 0.055f));
}

/* color.hpp:297-300: round half away from zero, then the conversion to uint8_t */
WPT_PP_HD uint8_t floatToByte(float x)
{
    return (uint8_t)__builtin_roundf(x * 255.0f);
}

/* postproc.hpp:44-61: one component */
template<class M> WPT_PP_HD uint8_t toSrgbByte(float v)
{
    const float c = v < 1.0f ? v : 1.0f; /* min(v, 1.0f), gvm.hpp:88 */
    return floatToByte(rgbToSrgbHelper<M>(c));
}

/* postproc.hpp:65-75: the luminance maxLuminance() reduces over */
WPT_PP_HD float luminance(V3 rgb) { return rgbToXyz(rgb).y; }

/* postproc.hpp:77-92 */
WPT_PP_HD V3 uniformRationalQuantization(V3 rgb, float maxVal, float brightness)
{
    V3 xyz = rgbToXyz(rgb);
    const float oldY = xyz.y / 100.0f;
    const float newY = brightness * oldY / ((brightness - 1.0f) * oldY + maxVal);
    xyz = adjustY(xyz, newY * 100.0f);
    return xyzToRgb(xyz);
}

/* postproc.hpp:94-108 */
WPT_PP_HD V3 scaleLuminance(V3 rgb, float factor, float clamp)
{
    V3 xyz = rgbToXyz(rgb);
    float newY = factor * xyz.y;
    if (clamp > 0.0f && newY > 100.0f * clamp)
        newY = 100.0f * clamp;
    xyz = adjustY(xyz, newY);
    return xyzToRgb(xyz);
}

} /* namespace wptpp */

#endif
