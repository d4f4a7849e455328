#!/bin/bash
# occupancy sensitivity of the lighting kernel (alone, 4K / 4096 lights)
O=gpurun_out/r02b2; mkdir -p $O
for cfg in "2 2" "2 3" "2 4" "2 5" "1 4" "1 6" "1 7" "1 8"; do
  set -- $cfg
  GR_LIGHTING_PX=$1 GR_LIGHTING_WGS_PER_CU=$2 timeout 120 python tools/lighting_only.py 2>&1 | tail -1
done | tee $O/occupancy.txt
