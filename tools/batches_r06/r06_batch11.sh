#!/bin/bash
O=$PWD/gpurun_out/r06l; mkdir -p $O; ROOT=$PWD
alone() { ( export GRANITE_LIB_DIR=$1; timeout 200 python tools/lighting_only.py 3840 2160 $2 2>/dev/null | sed "s/^/alone $1 /" ) }
for round in 1 2 3; do for l in lib lib_expect lib_head; do alone $l default; done; done 2>&1 | tee $O/alone.txt
cd /tmp && export TMPDIR=/tmp
for l in lib lib_expect; do
  GRANITE_LIB_DIR=$l rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU2 --kernel-trace -d $O/pmc_$l -o pmc --output-format csv -- python $ROOT/tools/lighting_only.py 3840 2160 default > $O/pmc_$l.log 2>&1
  python $ROOT/tools/pmc_summary.py $O/pmc_$l | grep "kernel\|k_lighting" | tee $O/pmc_$l.txt
done
