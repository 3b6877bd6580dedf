/* wpt_k_pc_full.hip -- instantiates wpt_pathtrace_pc<FEAT_ALL, false> (one variant per file: parallel builds) */
#include "wpt_pathtrace_pc.inc.h"

namespace wptk {

void launchPcFull(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace_pc<FEAT_ALL, false>), grid, dim3(PC_WG), 0, stream, args);
}

}
