/* wpt_k_basic_lds.hip -- instantiates wpt_pathtrace<FEAT_BASIC, false, true> (one variant per file: parallel builds) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasicLds(const KernelArgs& args, dim3 grid, size_t sceneLdsBytes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_BASIC, false, true, 4>), grid, dim3(WG), COLD_BYTES + sceneLdsBytes, stream, args);
}

}
