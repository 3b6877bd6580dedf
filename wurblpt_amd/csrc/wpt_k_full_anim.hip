/* wpt_k_full_anim.hip -- instantiates wpt_pathtrace<FEAT_ALL | FEAT_ANIM, false, false>: all features plus an exposure
 * interval (per-ray time, moving camera) and animated mesh instances; 2 waves per SIMD for the wider path state */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullAnim(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_ALL | FEAT_ANIM, false, false, 2>, args, grid, COLD_BYTES, stream);
}

}
