/* wpt_k_wf_full.hip -- the wavefront kernels (wpt_wavefront.inc.h) for all features: wf_shade<FEAT_ALL> */
#define WPT_MATH_TABLES_IN_LDS
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: scatter keeps what it read from the textures for the evaluation towards the light */
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfFull, FEAT_ALL, true)
}
