/*
 * postproc.hpp -- the output side of a rendering: sRGB conversion and dynamic range reduction
 * (reference interface: libwurblpt/postproc.hpp:44-108, same function names and arguments).
 *
 * Each function hands the frame to the HIP library (wpt_postproc_host in wurblpt_hip.h), which
 * runs one thread per pixel on the device; results equal the reference's loops value for value
 * (tests/test_gpu_parity.py).  Like everything else here there is no CPU fallback: without a
 * device the functions throw.
 */
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../wurblpt_hip.h"
#include "array.hpp"

namespace WurblPT {

namespace postprocdetail {

/* first three components of every element, tightly packed */
inline std::vector<float> packRgb(const Array<float>& img)
{
    if (img.componentCount() < 3)
        throw std::runtime_error("post-processing needs an RGB image");
    std::vector<float> rgb(img.elementCount() * 3);
    for (size_t i = 0; i < img.elementCount(); i++)
        for (size_t c = 0; c < 3; c++)
            rgb[3 * i + c] = img[i][c];
    return rgb;
}

inline void run(int op, const std::vector<float>& rgb, void* out, float a, float b)
{
    if (rgb.empty())
        return;
    if (wpt_postproc_host(op, rgb.data(), out, rgb.size() / 3, a, b) != WPT_OK)
        throw std::runtime_error(std::string("post-processing failed: ") + wpt_last_error());
}

inline Array<float> floatResult(int op, const Array<float>& img, float a, float b)
{
    const std::vector<float> rgb = packRgb(img);
    std::vector<float> out(rgb.size());
    run(op, rgb, out.data(), a, b);
    Array<float> r(img.dimension(0), img.dimension(1), img.componentCount());   /* further components stay 0 */
    r.globalTagList() = img.globalTagList();
    for (size_t i = 0; i < r.elementCount(); i++)
        for (size_t c = 0; c < 3; c++)
            r[i][c] = out[3 * i + c];
    return r;
}

}

/* Convert RGB(float) to SRGB(uint8); postproc.hpp:44-61 */
inline Array<uint8_t> toSRGB(const Array<float>& img)
{
    const std::vector<float> rgb = postprocdetail::packRgb(img);
    Array<uint8_t> r(img.dimension(0), img.dimension(1), 3);
    r.globalTagList() = img.globalTagList();
    postprocdetail::run(0, rgb, r.data(), 0.0f, 0.0f);
    return r;
}

/* postproc.hpp:65-75 */
inline float maxLuminance(const Array<float>& img)
{
    float lum = 0.0f;
    postprocdetail::run(3, postprocdetail::packRgb(img), &lum, 0.0f, 0.0f);
    return lum;
}

/* Schlick's uniform rational quantization, brightness in [1,inf); results are in [0,1] and can go
 * straight to toSRGB(); postproc.hpp:77-93 */
inline Array<float> uniformRationalQuantization(const Array<float>& img, float maxVal, float brightness)
{
    return postprocdetail::floatResult(1, img, maxVal, brightness);
}

/* postproc.hpp:95-110 */
inline Array<float> scaleLuminance(const Array<float>& img, float factor, float clamp = 1.0f)
{
    return postprocdetail::floatResult(2, img, factor, clamp);
}

}
