#!/bin/bash
# usage: bash tools/variant_build.sh <name> <sed expression> [file]
# An experiment without variant code in the product sources: copies wurblpt_amd/csrc to gpurun_out/variant_<name>/csrc, applies
# the sed expression to [file] (default wpt_pathtrace.inc.h) there and builds that tree's libraries into wurblpt_amd/lib_<name>
# (they travel to the GPU box; run with WPT_LIB_DIR=lib_<name>).  The expression is printed: a quoted number names its change.
set -e
NAME=$1; EXPR=$2; FILE=${3:-wpt_pathtrace.inc.h}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
V=$ROOT/gpurun_out/variant_$NAME
rm -rf "$V"; mkdir -p "$V/wurblpt_amd" "$V/include"
cp -r "$ROOT/wurblpt_amd/csrc" "$V/wurblpt_amd/csrc"; rm -rf "$V/wurblpt_amd/csrc/build"
cp -r "$ROOT/wurblpt_amd/host" "$V/wurblpt_amd/host"
cp -r "$ROOT/include/." "$V/include/"
sed -i "$EXPR" "$V/wurblpt_amd/csrc/$FILE"
echo "variant $NAME: sed '$EXPR' $FILE"; diff <(cat "$ROOT/wurblpt_amd/csrc/$FILE") "$V/wurblpt_amd/csrc/$FILE" || true
make -s -j8 -C "$V/wurblpt_amd/csrc" LIB="$ROOT/wurblpt_amd/lib_$NAME"
