/* constants.hpp -- physical constants used by scene descriptions (reference constants.hpp:29-42) */
#pragma once

namespace WurblPT {

constexpr float refractiveIndexOfVacuum = 1.0f;
constexpr float refractiveIndexOfAir = 1.00028f;

}
