/*
 * tgd/io.hpp -- TGD::save / TGD::load as the reference's applications call them (wurblpt-cornellbox.cpp:274-275:
 * `TGD::save(hdrImg, "image.exr")`), forwarding to this framework's own writers and decoders (wurblpt/imageio.hpp:
 * PNG, PPM / PGM, PFM, OpenEXR, PFS; reading also TGA, Radiance HDR, JPEG).  See tgd/array.hpp for what this is and
 * is not.  Files named *.tgd get this framework's own raw layout (the text header below, then the data), which is
 * not claimed to be libtgd's native format: read them back with TGD::load of this header.
 */
#pragma once

#include <stdexcept>
#include <string>

#include "array.hpp"
#include "../wurblpt/imageio.hpp"

namespace TGD {

/* libtgd reports failures through an Error value; here a failure throws (the applications do not look at the result) */
inline void save(const ArrayContainer& array, const std::string& fileName)
{
    std::string error;
    if (!WurblPT::saveImage(array, fileName, &error))
        throw std::runtime_error("TGD::save: " + error);
}

inline ArrayContainer load(const std::string& fileName)
{
    std::string error;
    ArrayContainer a = WurblPT::loadImage(fileName, &error);
    if (a.elementCount() == 0)
        throw std::runtime_error("TGD::load: " + error);
    return a;
}

}
