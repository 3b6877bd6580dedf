/* wpt_k_full_rgl_wide.hip -- instantiates wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false, 3, true>: measured BRDFs, the wide walk */
#define WPT_MATERIAL_CACHE /* as wpt_k_full_rgl.hip */
#define WPT_RGL_INLINE /* wpt_rgl.h: the measured-BRDF model inlined (this kernel: + 3.5 %) */
#define WPT_MATH_TABLES_IN_LDS
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullRglWide(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false, 3, true>, args, grid, COLD_BYTES, stream);
}

}
