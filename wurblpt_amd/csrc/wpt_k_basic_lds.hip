/* wpt_k_basic_lds.hip -- instantiates wpt_pathtrace<FEAT_BASIC, false, true> (one variant per file: parallel builds) */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#include <cstdlib>

#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasicLds(const KernelArgs& args, dim3 grid, size_t sceneLdsBytes, hipStream_t stream)
{
    /* measurements: WPT_EXTRA_LDS bytes of LDS more per workgroup (what a design that keeps more per path in LDS would pay in
     * workgroups per compute unit; the pixel pool sizes the launch by the occupancy that results) */
    static const size_t extra = getenv("WPT_EXTRA_LDS") ? size_t(atol(getenv("WPT_EXTRA_LDS"))) : 0;
    launchMaybePooled(wpt_pathtrace<FEAT_BASIC, false, true, 4>, args, grid, COLD_BYTES + sceneLdsBytes + extra, stream);
}

}
