/* wpt_k_wfg_full.hip -- instantiates wpt_pathtrace_wf<FEAT_ALL, false, true>: pixel states in global memory */
#include "wpt_pathtrace_wf.inc.h"

namespace wptk {

void launchWfgFull(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace_wf<FEAT_ALL, false, true>), grid, dim3(WF_WG), ldsBytes, stream, args);
}

}
