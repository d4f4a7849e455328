#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
alone() { ( export GRANITE_LIB_DIR=$1; [ "$2" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$2; timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $1 wgs=$2 /" ) }
for round in 1 2 3; do for l in "$@"; do alone $l -; done; done 2>&1 | tee $O/alone.txt
