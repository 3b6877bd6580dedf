/* wpt_k_wf_full.hip -- the wavefront kernels (wpt_wavefront.inc.h) for all features: wf_trace with spheres, wf_shade<FEAT_ALL> */
#define WPT_MATH_TABLES_IN_LDS
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfFull, FEAT_ALL, true, false)
}
