/* wpt_k_full_rgl_count.hip -- instantiates wpt_pathtrace<FEAT_ALL | FEAT_RGL, true, false> (work counters) */
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: the measured-BRDF model's incident-direction part is shared between scatter and the evaluation towards the light */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullRglCount(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_ALL | FEAT_RGL, true, false, 2>), grid, dim3(WG), COLD_BYTES, stream, args);
}

}
