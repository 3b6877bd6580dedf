/* wpt_k_wf_trace.hip -- the wavefront form's trace kernel (wpt_wavefront.inc.h), with and without sphere leaves: the one unit
 * that instantiates it */
#include "wpt_wavefront.inc.h"

namespace wptk {

void launchWfTrace(bool spheres, const WfArgs& a, dim3 grid, hipStream_t stream)
{
    const size_t lds = size_t(a.topNodes) * 32;
    if (spheres)
        hipLaunchKernelGGL((wf_trace<true>), grid, dim3(WG), lds, stream, a);
    else
        hipLaunchKernelGGL((wf_trace<false>), grid, dim3(WG), lds, stream, a);
}

int wfTraceBlocksPerCu(bool spheres, size_t dynamicLdsBytes)
{
    int perCu = 0;
    const hipError_t e = spheres ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, wf_trace<true>, (int)WG, dynamicLdsBytes)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, wf_trace<false>, (int)WG, dynamicLdsBytes);
    return e == hipSuccess ? perCu : 0;
}

}
