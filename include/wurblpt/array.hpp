/*
 * array.hpp -- the image container of textures, sensor frames and ground truth arrays.
 *
 * The reference uses libtgd's TGD::ArrayContainer / TGD::Array<T> for this (texture_image.hpp:45,
 * sensor_rgb.hpp:37) and its applications spell the type TGD::Array<float> (wurblpt-cornellbox.cpp:271).
 * libtgd is an external library; where it is not installed, include/tgd/array.hpp of this repository provides the
 * container under the same names, and WurblPT:: sees it through the aliases below either way.
 */
#pragma once

#if defined(__has_include)
#if __has_include(<tgd/array.hpp>)
#include <tgd/array.hpp>
#else
#include "../tgd/array.hpp"
#endif
#else
#include "../tgd/array.hpp"
#endif

namespace WurblPT {

using TGD::Array;
using TGD::ArrayContainer;
using TGD::ComponentType;
using TGD::ComponentTypeOf;
using TGD::TagList;
using TGD::componentTypeSize;
using TGD::float32;
using TGD::int32;
using TGD::uint16;
using TGD::uint8;

}
