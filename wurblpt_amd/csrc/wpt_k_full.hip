/* wpt_k_full.hip -- instantiates wpt_pathtrace<FEAT_ALL, false, false> (one variant per file: parallel builds) */
#define WPT_MATH_TABLES_IN_LDS /* this unit's kernels keep the tables of expf / powf in LDS (wpt_math.h) */
#ifdef WPT_TOP_IN_LDS
#define WPT_TOP_IN_LDS_HERE WPT_TOP_IN_LDS /* variant build: this unit's kernel walks the top of the tree from LDS */
#endif
#ifdef WPT_WIDE_WALK
#define WPT_WIDE_WALK_HERE /* variant build: this unit's kernel walks the tree collapsed by one level */
#endif
#include "wpt_pathtrace.inc.h"

#ifndef WPT_FULL_FEATURES
#define WPT_FULL_FEATURES FEAT_ALL /* experiments: a narrower set for scenes that need no more */
#endif
#ifndef WPT_FULL_OCC
#define WPT_FULL_OCC 4 /* experiments: 3 = 168 registers, no spills, three workgroups per compute unit */
#endif

namespace wptk {

void launchFull(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
#ifdef WPT_TOP_IN_LDS
    launchMaybePooled(wpt_pathtrace<WPT_FULL_FEATURES, false, false, WPT_FULL_OCC>, args, grid, COLD_BYTES + WPT_TOP_IN_LDS * 32, stream);
#else
    launchMaybePooled(wpt_pathtrace<WPT_FULL_FEATURES, false, false, WPT_FULL_OCC>, args, grid, COLD_BYTES, stream);
#endif
}

}
