/* wpt_k_wf_full_rgl.hip -- the wavefront kernels (wpt_wavefront.inc.h) with measured BRDFs: wf_shade<FEAT_ALL | FEAT_RGL> */
#define WPT_MATH_TABLES_IN_LDS
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfFullRgl, FEAT_ALL | FEAT_RGL, true, false)
}
