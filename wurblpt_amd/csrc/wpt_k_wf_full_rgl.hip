/* wpt_k_wf_full_rgl.hip -- the wavefront kernels (wpt_wavefront.inc.h) with measured BRDFs: wf_shade<FEAT_ALL | FEAT_RGL> */
#define WPT_MATH_TABLES_IN_LDS
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: scatter keeps what it read from the textures for the evaluation towards the light */
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfFullRgl, FEAT_ALL | FEAT_RGL, true, false)
}
