/* wpt_k_full_rgl.hip -- instantiates wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false>: all features plus measured BRDFs */
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: the measured-BRDF model's incident-direction part is shared between scatter and the evaluation towards the light */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullRgl(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false, 4>, args, grid, COLD_BYTES, stream);
}

}
