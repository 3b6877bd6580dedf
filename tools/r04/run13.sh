#!/bin/bash
# round 4, thirteenth GPU call: launch geometry of the wavefront form on the measured-BRDF frame (16 spp) after the light ray
# travels beside the continuation: groups, chunk, lanes before a refill, step budget, trace workgroups per CU
O=gpurun_out/r04n
mkdir -p $O
timeout -k 10 900 python tools/wf_check.py time measured_like_3840x2160_529spp_rgl 4 2,0,0 1,0,0 3,0,0 4,0,0 2,64,0 2,256,0 2,0,0x800 2,0,0x2000 2,0,0x1000000 2,0,0x4000000 2,0,0xffff0000 0x302,0,0 0x402,0,0 3,0,0x2000 > $O/wf_geometry.txt 2>&1
cat $O/wf_geometry.txt | grep -v "^Building\|^Linear\|amdgpu.ids"
