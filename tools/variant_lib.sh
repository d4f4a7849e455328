#!/bin/bash
# Builds an A/B variant of the kernel library: granite_amd/lib_<name> = the product build with ONE translation unit recompiled with extra flags.
#   tools/variant_lib.sh <name> <unit> "<flags>"        e.g.  tools/variant_lib.sh w6 lighting "-DLV_WAVES_PER_EU=6"
# Run here (hipcc cross-compiles); the directory travels to the GPU box with the snapshot and is selected with GRANITE_LIB_DIR=lib_<name>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; UNIT=$2; FLAGS=$3
make -s -j8 -C "$ROOT/granite_amd/csrc" ARCH=gfx950
rm -rf "$ROOT/granite_amd/lib_$NAME"
cp -r "$ROOT/granite_amd/lib" "$ROOT/granite_amd/lib_$NAME"
rm -f "$ROOT/granite_amd/lib_$NAME/obj/$UNIT.o"
make -s -C "$ROOT/granite_amd/csrc" ARCH=gfx950 OUT=../lib_$NAME EXTRA_$UNIT="$FLAGS"
echo "built granite_amd/lib_$NAME ($UNIT: $FLAGS)"
