/* wpt_k_basic_lds.hip -- instantiates wpt_pathtrace<FEAT_BASIC, false, true> (one variant per file: parallel builds) */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasicLds(const KernelArgs& args, dim3 grid, size_t sceneLdsBytes, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_BASIC, false, true, 4>, args, grid, COLD_BYTES + sceneLdsBytes, stream);
}

}
