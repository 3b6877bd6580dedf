/* wpt_k_basic_lds_pairs.hip -- instantiates wpt_pathtrace<FEAT_BASIC, false, true, 4, ..., PAIRS = true>: the LDS kernel with
 * paired node steps.  It is picked for launches that cannot fill more than two waves per SIMD (one rank's share of a
 * small frame over many GPUs): a wave alone issues vector instructions a quarter of the time, and two node steps per
 * scheduling decision are fewer scalar instructions per step (measured 1.04x there, 0.975x on a full frame). */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchBasicLdsPairs(const KernelArgs& args, dim3 grid, size_t ldsBytes, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_BASIC, false, true, 4, ORDERED_KERNELS, true>), grid, dim3(WG), ldsBytes, stream, args);
}

}
