#!/bin/bash
# usage (GPU box): bash tools/ab.sh <name> "<extra hipcc flags>" <command ...>
# A/B of a build variant: builds a second pair of libraries with the extra flags into wurblpt_amd/lib_<name> (objects in
# wurblpt_amd/csrc/build_<name>), then runs the command twice -- with the product libraries and with the variant
# (WPT_LIB_DIR=lib_<name>; bench.py prints the library it ran under "library").  The flags are part of the output, so a
# quoted number can be reproduced from the tree:  bash tools/ab.sh steps4 "-DWPT_LDS_STEPS=4" python bench.py --no-cpu-baseline
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$ROOT/wurblpt_amd/csrc" BUILD=build_$NAME LIB=../lib_$NAME EXTRA="$EXTRA"
echo "== product build: $*"
"$@"
echo "== variant $NAME (EXTRA=$EXTRA): $*"
WPT_LIB_DIR=lib_$NAME "$@"
