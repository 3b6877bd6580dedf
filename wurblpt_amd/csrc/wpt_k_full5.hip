/* wpt_k_full5.hip -- wpt_pathtrace<FEAT_ALL>, five waves per SIMD (five 256-thread workgroups with eight cold slots
 * per lane fill a CU's 160 KiB of LDS exactly) */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFull5(const KernelArgs& args, uint32_t lanes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_ALL, false, false, 5>), dim3((lanes + WG - 1) / WG), dim3(WG), COLD_BYTES, stream, args);
}

}
