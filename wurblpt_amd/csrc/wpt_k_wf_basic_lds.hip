/* wpt_k_wf_basic_lds.hip -- instantiates wpt_pathtrace_wf<FEAT_BASIC, true> (one variant per file: parallel builds) */
#include "wpt_pathtrace_wf.inc.h"

namespace wptk {

void launchWfBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream)
{
    /* more than 64 KiB of dynamic LDS has to be asked for once per kernel */
    static bool configured = false;
    auto kernel = wpt_pathtrace_wf<FEAT_BASIC, true>;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(WF_WG), ldsBytes, stream, args);
}

}
