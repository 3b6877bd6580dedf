/* wpt_k_full_rgl.hip -- instantiates wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false>: all features plus measured BRDFs */
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: the measured-BRDF model's incident-direction part is shared between scatter and the evaluation towards the light */
#define WPT_RGL_INLINE /* wpt_rgl.h: the measured-BRDF model inlined (this kernel: + 3.5 %) */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

/* three waves per SIMD (168 registers): the measured-BRDF evaluation spills heavily at 128 (Bistro-class 16-spp frame 1423 against
 * 1477 ms) */

namespace wptk {

void launchFullRgl(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false, 3>, args, grid, COLD_BYTES, stream);
}

}
