#!/bin/bash
# Round 6, third GPU batch: is the lighting launch bound by the vector pipe or by latency?  Occupancy sweep (a latency-bound kernel scales with
# resident waves, a pipe-bound one does not) and PMC passes of round 5's kernel and the overlapped-chain kernel.
O=gpurun_out/r06c; mkdir -p $O
alone() { ( export GRANITE_LIB_DIR=$1; [ "$2" != "-" ] && export GR_LIGHTING_WGS_PER_CU=$2; timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $1 wgs=$2 /" ) }
for round in 1 2; do for w in 1 2 3 4 5; do alone lib_r5 $w; alone lib $w; done; done 2>&1 | tee $O/occupancy.txt
GRANITE_LIB_DIR=lib_r5 bash tools/pmc_lighting.sh r06c/pmc_r5 > /dev/null 2>&1; cat gpurun_out/r06c/pmc_r5/summary.txt
GRANITE_LIB_DIR=lib bash tools/pmc_lighting.sh r06c/pmc_new > /dev/null 2>&1; cat gpurun_out/r06c/pmc_new/summary.txt
rm -rf gpurun_out/r06c/pmc_*/cls* gpurun_out/r06c/pmc_*/cyc*
