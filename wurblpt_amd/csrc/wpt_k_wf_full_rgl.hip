/* wpt_k_wf_full_rgl.hip -- the wavefront kernels (wpt_wavefront.inc.h) with measured BRDFs: wf_shade<FEAT_ALL | FEAT_RGL> */
/* three waves per SIMD: the measured-BRDF evaluation is long and spills at 128 registers (measured: Bistro-class 1136 against 1248 ms
 * per 16-spp frame; the other shade kernels are indifferent) */
#define WF_SHADE_WAVES 3
#define WPT_RGL_INLINE /* wpt_rgl.h: the model inlined (with the interleaved colour + luminance table: 125.2 -> 129.1 Msamples/s at 16 spp; as calls 120.7) */
#define WPT_MATH_TABLES_IN_LDS
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: scatter keeps what it read from the textures for the evaluation towards the light */
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfFullRgl, FEAT_ALL | FEAT_RGL, true)
}
