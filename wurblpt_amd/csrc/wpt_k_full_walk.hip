/* wpt_k_full_walk.hip -- wpt_pathtrace<FEAT_ALL> once more, compiled WITH the machine-level loop-invariant code motion
 * that the other kernels switch off (Makefile): the hoisted values cost the long round registers, but the traversal loop
 * is scheduled better for it, and for trees far larger than the caches the frame time is traversal (10 M triangles:
 * 56.3 against 51.8 Msamples/s; the Sponza-class frame the other way round, 107 against 117).  It also leaves out the
 * node prefetch, which in this build costs what it gains elsewhere.  Launches pick by tree size. */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullWalk(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_ALL, false, false, 4, 1>), grid, dim3(WG), COLD_BYTES, stream, args);
}

}
