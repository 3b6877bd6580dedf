#!/bin/bash
# usage (GPU box): bash tools/ab.sh <name> "<extra hipcc flags>" <command ...>
# A/B of a build variant: builds a second pair of libraries with the extra flags into wurblpt_amd/lib_<name> (objects in
# wurblpt_amd/csrc/build_<name>), then runs the command twice -- with the product libraries and with the variant
# (WPT_LIB_DIR=lib_<name>; bench.py prints the library it ran under "library").  The flags are part of the output, so a
# quoted number can be reproduced from the tree:  bash tools/ab.sh steps4 "-DWPT_LDS_STEPS=4" python bench.py --no-cpu-baseline
# Variant code in the tree (none of it is in the product library): -DWPT_LDS_STEPS=n, -DWPT_LDS_PREFETCH=1, -DWPT_TOP_IN_LDS=200 (the
# single kernel walks the tree's upper levels from LDS), -DWPT_WIDE_WALK (the all-features single kernel walks the tree collapsed by one level: a prototype, slower and not exact yet),
# -DWPT_EVAL_BEHIND_RAY (measured BRDFs are evaluated towards the light behind
# the light ray, wavefront form), -DWPT_FULL_FEATURES=mask / -DWPT_FULL_OCC=3 / -DWPT_RGL_OCC=n, -DWF_TRACE_WAVES=n, -DWF_TRACE_UNIFIED=1,
# -DWPT_SEPARATE_STARTS.  Building on the GPU box costs box time: build here (make -C wurblpt_amd/csrc BUILD=build_x LIB=../lib_x
# EXTRA=...) and run with WPT_LIB_DIR=lib_x instead, the libraries travel with the snapshot.
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$ROOT/wurblpt_amd/csrc" BUILD=build_$NAME LIB=../lib_$NAME EXTRA="$EXTRA"
echo "== product build: $*"
"$@"
echo "== variant $NAME (EXTRA=$EXTRA): $*"
WPT_LIB_DIR=lib_$NAME "$@"
