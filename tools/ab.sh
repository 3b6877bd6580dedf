#!/bin/bash
# usage (GPU box): bash tools/ab.sh <name> "<extra hipcc flags>" <command ...>
# A/B of a build variant: builds a second pair of libraries with the extra flags into wurblpt_amd/lib_<name> (objects in
# wurblpt_amd/csrc/build_<name>), then runs the command twice -- with the product libraries and with the variant
# (WPT_LIB_DIR=lib_<name>; bench.py prints the library it ran under "library").  The flags are part of the output, so a
# quoted number can be reproduced from the tree.  The product sources carry no variant code since round 4: an experiment is a
# change in a work tree of its own (git worktree add ...; make -C <tree>/wurblpt_amd/csrc LIB=$PWD/wurblpt_amd/lib_x), run with
# WPT_LIB_DIR=lib_x; the variants round 3 measured (-DWPT_LDS_STEPS, -DWPT_LDS_PREFETCH, -DWPT_TOP_IN_LDS, -DWPT_EVAL_BEHIND_RAY,
# -DWF_TRACE_UNIFIED, -DWPT_SEPARATE_STARTS, WPT_REDEAL, WPT_XCD_BANDS, leaf records) are in the tree at commit 5178f84.
# Building on the GPU box costs box time: build here and run with WPT_LIB_DIR=lib_x, the libraries travel with the snapshot.
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$ROOT/wurblpt_amd/csrc" BUILD=build_$NAME LIB=../lib_$NAME EXTRA="$EXTRA"
echo "== product build: $*"
"$@"
echo "== variant $NAME (EXTRA=$EXTRA): $*"
WPT_LIB_DIR=lib_$NAME "$@"
