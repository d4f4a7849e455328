#!/bin/bash
# PMC passes over the lighting kernel alone (tools/lighting_only.py, 4K / 4096 lights): instruction-class histogram and
# issue / wait cycles.  Counters only (+ kernel trace), one pass per group.  Usage (GPU box): tools/pmc_lighting.sh <tag>
set -u
TAG=${1:-pmc_light}; SIZE="${2:-} ${3:-}"   # optional: width height
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" --output-format csv -- python "$ROOT/tools/lighting_only.py" $SIZE > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
run cls1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU
run cls2 SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_FLOPS_FP32 SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VSKIPPED
run cyc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run cyc2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
