#!/bin/bash
# Round 6, first GPU batch: lighting alone / in the frame for register-fit and workgroup-shape variants, raw per-tile stamp records for
# tools/lighting_residency.py.  Variant libraries are built in the container (tools/variant_lib.sh).
O=gpurun_out/r06a; mkdir -p $O
alone() { # lib wgs
  ( export GRANITE_LIB_DIR=$1; [ "$2" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$2; timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $1 wgs=$2 /" )
}
for round in 1 2; do
  alone lib -; alone lib 4; alone lib 6; alone lib_w6 6; alone lib_w6 5; alone lib_w7 7; alone lib_w7 6; alone lib_lw1 -; alone lib_lw2 -; alone lib_lw1 6; alone lib_lw1 8
done 2>&1 | tee $O/alone.txt
frame() { # name lib wgs
  ( export GRANITE_LIB_DIR=$2; [ "$3" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$3
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$1.json 2>/dev/null
    python tools/bench_brief.py $O/bench_$1.json | sed "s/^/frame $1 /" )
}
for round in 1 2; do
  frame base.$round lib -; frame w6.$round lib_w6 6; frame w7.$round lib_w7 7; frame lw1.$round lib_lw1 -; frame lw2.$round lib_lw2 -
done 2>&1 | tee $O/frame.txt
LV_STAMP_DUMP=$O/stamps_base GRANITE_LIB_DIR=lib_stamp timeout 200 python tools/lighting_stamps.py $O/tiles_base.txt > /dev/null 2>$O/tiles_base.err
LV_STAMP_DUMP=$O/stamps_lw1 GRANITE_LIB_DIR=lib_stamp_lw1 timeout 200 python tools/lighting_stamps.py $O/tiles_lw1.txt > /dev/null 2>$O/tiles_lw1.err
tail -3 $O/tiles_base.txt $O/tiles_lw1.txt
