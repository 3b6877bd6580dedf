/* wpt_k_pc_basic_lds.hip -- instantiates wpt_pathtrace_pc<FEAT_BASIC, true> (one variant per file: parallel builds) */
#include "wpt_pathtrace_pc.inc.h"

namespace wptk {

void launchPcBasicLds(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace_pc<FEAT_BASIC, true>), grid, dim3(PC_WG), ldsBytes, stream, args);
}

}
