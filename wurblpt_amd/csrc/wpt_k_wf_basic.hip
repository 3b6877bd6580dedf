/* wpt_k_wf_basic.hip -- the wavefront kernels (wpt_wavefront.inc.h) for the basic feature set: wf_shade<FEAT_BASIC> */
#define WPT_MATH_TABLES_IN_LDS
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: scatter keeps what it read from the textures for the evaluation towards the light */
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfBasic, FEAT_BASIC, false)
}
