#!/bin/bash
# Round 6, second GPU batch: the lighting kernel with its four memory round trips overlapped (lib) against round 5's (lib_r5).
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_lighting_adversarial.py tests/test_gpu_packed_hdr.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_lighting.txt
alone() { ( export GRANITE_LIB_DIR=$1; [ "$2" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$2; timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $1 wgs=$2 /" ) }
for round in 1 2 3; do alone lib_r5 -; alone lib -; alone lib 4; done 2>&1 | tee $O/alone.txt
frame() { ( export GRANITE_LIB_DIR=$2; [ "$3" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$3
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$1.json 2>/dev/null
    python tools/bench_brief.py $O/bench_$1.json | sed "s/^/frame $1 /" ) }
for round in 1 2; do frame r5.$round lib_r5 -; frame new.$round lib -; frame new4.$round lib 4; done 2>&1 | tee $O/frame.txt
LV_STAMP_DUMP=$O/stamps_new GRANITE_LIB_DIR=lib_stamp timeout 200 python tools/lighting_stamps.py $O/tiles_new.txt > /dev/null 2>$O/tiles_new.err
head -40 $O/tiles_new.txt | tail -12
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_fullsize.txt
