#!/bin/bash
# Round 6: two-group sphere culling in the wide-window path against the kernel before it (lib_prev), parity and time.
O=$PWD/gpurun_out/r06x; mkdir -p $O; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | tee $O/pytest.txt
alone() { ( export GRANITE_LIB_DIR=$1; timeout 200 python tools/lighting_only.py 3840 2160 $2 2>/dev/null | sed "s/^/alone $1 /" ) }
for round in 1 2 3; do for sc in default depth_split hot_spot; do for l in lib lib_prev; do alone $l $sc; done; done; done 2>&1 | tee $O/alone.txt
cd /tmp && export TMPDIR=/tmp
for l in lib lib_prev; do
  GRANITE_LIB_DIR=$l rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace -d $O/pmc_$l -o pmc --output-format csv -- python $ROOT/tools/lighting_only.py 3840 2160 default > $O/pmc_$l.log 2>&1
  python $ROOT/tools/pmc_summary.py $O/pmc_$l | grep "kernel\|k_lighting" | tee $O/pmc_$l.txt
done
find $O -name "*.csv" -delete
