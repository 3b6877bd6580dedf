/* wpt_k_wfg_basic_lds.hip -- instantiates wpt_pathtrace_wf<FEAT_BASIC, true, true>: pixel states in global memory */
#include "wpt_pathtrace_wf.inc.h"

namespace wptk {

void launchWfgBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace_wf<FEAT_BASIC, true, true>), grid, dim3(WF_WG), ldsBytes, stream, args);
}

}
