#!/bin/bash
# A/B of lighting-kernel builds on one GPU box: alone (tools/lighting_only.py), in the frame (bench.py driver line) and, with the stamp
# build, the per-tile timeline and phase split (tools/lighting_stamps.py).  Variant libraries are built HERE (in the container) first:
#   make -C granite_amd/csrc OUT=../lib_<name> EXTRA_lighting="-D..."
#   make -C granite_amd/csrc OUT=../lib_stamp EXTRA_lighting=-DLV_STAMP
# Usage (through gpurun): bash tools/lighting_ab.sh <tag> lib lib_<name> ...
TAG=${1:-light_ab}; shift; O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2; do
  for lib in "${@:-lib}"; do
    GRANITE_LIB_DIR=$lib timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $lib /"
  done
done
for lib in "${@:-lib}"; do
  GRANITE_LIB_DIR=$lib timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${lib}.json 2>/dev/null
  python tools/bench_brief.py $O/bench_${lib}.json | sed "s/^/$lib /"
done
if [ -d granite_amd/lib_stamp ]; then
  GRANITE_LIB_DIR=lib_stamp timeout 200 python tools/lighting_stamps.py $O/tiles.txt > /dev/null 2>$O/tiles.err; head -30 $O/tiles.txt
fi
