#!/bin/bash
# rocprofv3 PMC passes over a short bench.py run (separate passes: SQ issue counters, FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md's HBM/rocprofv3 section prescribes.  Usage (on the GPU box): tools/pmc_passes.sh <tag> [bench args]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" --output-format csv -- python "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM
run cls SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_BRANCH
run fetch FETCH_SIZE
run write WRITE_SIZE
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# the raw per-dispatch CSVs are tens of MiB per pass: gpurun copies at most 64 MiB back, the summaries are what is kept
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
