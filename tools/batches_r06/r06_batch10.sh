#!/bin/bash
# Round 6: the narrow path of the restructured lighting loop against the previous kernel (lib_head): instruction counts (PMC) and times, default scene.
O=$PWD/gpurun_out/r06k; mkdir -p $O; ROOT=$PWD
alone() { ( export GRANITE_LIB_DIR=$1; timeout 200 python tools/lighting_only.py 3840 2160 $2 2>/dev/null | sed "s/^/alone $1 /" ) }
for round in 1 2 3; do for l in lib lib_head; do alone $l default; done; done 2>&1 | tee $O/alone.txt
cd /tmp && export TMPDIR=/tmp
for l in lib lib_head; do
  GRANITE_LIB_DIR=$l rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU2 --kernel-trace -d $O/pmc_$l -o pmc --output-format csv -- python $ROOT/tools/lighting_only.py 3840 2160 default > $O/pmc_$l.log 2>&1
  python $ROOT/tools/pmc_summary.py $O/pmc_$l | tee $O/pmc_$l.txt
done
cd $ROOT
for sc in depth_split hot_spot; do alone lib $sc; alone lib_head $sc; done 2>&1 | tee -a $O/alone.txt
timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_fullsize.py -x -q -m gpu -k "lighting_matches or wide or worst_case or config3 or bruteforce" 2>&1 | tail -4 | tee $O/pytest.txt
