/* wpt_k_basic_lds5.hip -- wpt_pathtrace<FEAT_BASIC>, scene in LDS, five waves per SIMD: 320-thread workgroups, seven
 * cold slots per lane (launches whose path length gates are open), at most 96 registers */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasicLds5Gateless(const KernelArgs& args, uint32_t lanes, size_t sceneLdsBytes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_BASIC, false, true, 5, WG5, false>), dim3((lanes + WG5 - 1) / WG5), dim3(WG5),
            (SLOT_COUNT - 1) * WG5 * 16 + sceneLdsBytes, stream, args);
}

}
