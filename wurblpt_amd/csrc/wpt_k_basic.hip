/* wpt_k_basic.hip -- instantiates wpt_pathtrace<FEAT_BASIC, false, false> (one variant per file: parallel builds) */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasic(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_BASIC, false, false, 4>, args, grid, COLD_BYTES, stream);
}

}
