/* wpt_k_full_count.hip -- instantiates wpt_pathtrace<FEAT_ALL, true, false> (one variant per file: parallel builds) */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullCount(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_ALL, true, false, 2>), grid, dim3(WG), COLD_BYTES, stream, args);
}

}
