#!/bin/bash
# A/B of lighting-kernel builds and launch forms on one GPU box: alone (tools/lighting_only.py), in the frame (bench.py driver line) and,
# with the stamp build, the per-tile timeline (tools/lighting_stamps.py).  Variant libraries are built HERE (in the container) first:
#   make -C granite_amd/csrc OUT=../lib_<name> EXTRA_lighting="-D..."       (LV_* switches at the top of lighting.hip)
#   make -C granite_amd/csrc OUT=../lib_stamp EXTRA_lighting=-DLV_STAMP
# Usage (through gpurun): bash tools/lighting_ab.sh <tag> lib lib_<name> ...     forms: GR_LIGHTING_STATIC=banded, GR_LIGHTING_PERSISTENT=1
TAG=${1:-light_ab}; shift; O=gpurun_out/$TAG; mkdir -p $O
for lib in "${@:-lib}"; do
  for form in screen banded persistent; do
    unset GR_LIGHTING_STATIC GR_LIGHTING_PERSISTENT
    [ $form = banded ] && export GR_LIGHTING_STATIC=banded
    [ $form = persistent ] && export GR_LIGHTING_PERSISTENT=1
    [ $form != screen ] && [ $lib != lib ] && continue
    for i in 1 2; do GRANITE_LIB_DIR=$lib timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $lib $form /"; done
    GRANITE_LIB_DIR=$lib timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${lib}_$form.json 2>/dev/null
    python tools/bench_brief.py $O/bench_${lib}_$form.json | sed "s/^/$lib $form /"
    if [ $lib = lib ] && [ -d granite_amd/lib_stamp ]; then
      GRANITE_LIB_DIR=lib_stamp timeout 200 python tools/lighting_stamps.py $O/tiles_$form.txt > /dev/null 2>$O/tiles_$form.err; head -24 $O/tiles_$form.txt
    fi
  done
done
