/* constants.hpp -- what scene descriptions need of the reference's constants.hpp:29-42: the refractive indices that
 * MaterialGlass takes as defaults.  (The kernels keep their own numeric constants in wpt_device.h.) */
#pragma once

namespace WurblPT {

/* n of the medium a camera ray starts in, and of air at standard conditions for scenes that want it */
inline constexpr float refractiveIndexOfVacuum = 1.0f, refractiveIndexOfAir = 1.00028f;

}
