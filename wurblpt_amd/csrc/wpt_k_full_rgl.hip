/* wpt_k_full_rgl.hip -- instantiates wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false>: all features plus measured BRDFs */
#include "wpt_pathtrace.inc.h"

namespace wptk {

void launchFullRgl(const KernelArgs& args, dim3 grid, hipStream_t stream)
{
    hipLaunchKernelGGL((wpt_pathtrace<FEAT_ALL | FEAT_RGL, false, false, 4>), grid, dim3(WG), COLD_BYTES, stream, args);
}

}
