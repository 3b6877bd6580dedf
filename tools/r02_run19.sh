#!/bin/bash
set -o pipefail
O=gpurun_out/r02z
mkdir -p $O
timeout -k 10 200 python tools/size_scan.py > $O/size_scan.txt 2>&1 || exit 1
grep spp $O/size_scan.txt
timeout -k 10 400 python tools/sched_sweep.py cornell 16 > $O/sweep_cornell.txt 2>&1 || exit 1
tail -1 $O/sweep_cornell.txt; grep default $O/sweep_cornell.txt
timeout -k 10 500 python tools/sched_sweep.py sponza 8 > $O/sweep_sponza.txt 2>&1 || exit 1
tail -1 $O/sweep_sponza.txt; grep default $O/sweep_sponza.txt
