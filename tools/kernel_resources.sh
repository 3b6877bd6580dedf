#!/bin/bash
# usage: bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
# registers, scratch and waves per SIMD of every kernel, as the compiler reports them for the flags of wurblpt_amd/csrc/Makefile
cd "$(dirname "$0")/../wurblpt_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage"
echo "# translation unit : kernel : VGPRs, scratch bytes per lane, waves per SIMD, static LDS bytes per workgroup (dynamic LDS comes on top: 33 312 B for the path kernels and the shade kernels)"
for f in wpt_k_*.hip wpt_capi.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fPIC -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kr_$$.o 2>&1 \
    | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - \
    | grep -E "Function Name: (_ZN4wptk|_ZN12_GLOBAL)" \
    | sed -E "s/Function Name: ([^ \t]*)[ \t]*VGPRs: ([0-9]*)[ \t]*ScratchSize \[bytes\/lane\]: ([0-9]*)[ \t]*Occupancy \[waves\/SIMD\]: ([0-9]*)[ \t]*LDS Size \[bytes\/block\]: ([0-9]*)/$f : \1 : vgpr \2 scratch \3 waves\/SIMD \4 lds \5/"
done
rm -f /tmp/kr_$$.o
