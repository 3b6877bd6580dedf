/* wpt_k_pc_basic.hip -- instantiates wpt_pathtrace_pc<FEAT_BASIC, false> (one variant per file: parallel builds) */
#include "wpt_pathtrace_pc.inc.h"

namespace wptk {

void launchPcBasic(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace_pc<FEAT_BASIC, false>), grid, dim3(PC_WG), 0, stream, args);
}

}
