#!/bin/bash
O=gpurun_out/r06p; mkdir -p $O
for w in - 3 4 5 8; do ( [ "$w" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$w; timeout 300 python tools/lighting_two_streams.py 2>&1 | tail -3 ); done | tee $O/two_streams.txt
