/* wpt_k_wf_full_rgl.hip -- the wavefront kernels (wpt_wavefront.inc.h) with measured BRDFs: wf_shade<FEAT_ALL | FEAT_RGL> */
/* two waves per SIMD: the measured-BRDF evaluation is long -- at 128 registers it spilled 384 B per lane (Bistro-class 16-spp frame
 * 1248 ms), at 168 still some (1136 ms; with the interleaved table and the model inlined 1028); built for two waves it takes 214 registers and 64 B of stack, and the compiler keeps
 * more of the evaluation's independent loads in flight: 1010 ms (profiles/r04_measured_brdf_table.txt); the kernel waits for the depth
 * of its look-ups, which more waves do not shorten.  The other shade kernels are indifferent. */
#define WF_SHADE_WAVES 2
#define WPT_RGL_INLINE /* wpt_rgl.h: the model inlined (with the interleaved colour + luminance table: 125.2 -> 129.1 Msamples/s at 16 spp; as calls 120.7) */
#define WPT_MATH_TABLES_IN_LDS
#define WPT_MATERIAL_CACHE /* wpt_blocks.h: scatter keeps what it read from the textures for the evaluation towards the light */
#include "wpt_wavefront.inc.h"

namespace wptk {
WPT_WF_LAUNCHERS(wfFullRgl, FEAT_ALL | FEAT_RGL, true)
}
