/* wpt_k_basic_lds_redeal.hip -- instantiates wpt_pathtrace<FEAT_BASIC, false, true, 3, true>: the kernel with the scene in LDS
 * whose workgroups deal their paths to their lanes anew at every look at the lane counts (one variant per file) */
#define WPT_MATH_TABLES_IN_LDS
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasicLdsRedeal(const KernelArgs& args, dim3 grid, size_t sceneLdsBytes, hipStream_t stream)
{
    launchMaybePooled(wpt_pathtrace<FEAT_BASIC, false, true, 3, true>, args, grid, COLD_BYTES + sceneLdsBytes + REDEAL_BYTES, stream);
}

}
