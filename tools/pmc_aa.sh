#!/bin/bash
# PMC passes over the stand-alone AA kernel timings (tools/aa_time.py): what bounds FXAA / SMAA / TAA, and (separate passes) what they fetch
# and write against their algorithmic bytes -> gpurun_out/pmc_aa/aa_traffic.json (copy to profiles/aa_traffic.json: bench.py quotes it in the
# config-4 line).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_aa; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" --output-format csv -- python "$ROOT/tools/aa_time.py" 3840 2160 > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
python "$ROOT/tools/pmc_aa_traffic.py" "$OUT/summary.json" "$OUT/aa_traffic.json"
grep -E "taa|fxaa|smaa" "$OUT/summary.txt" | head -40
# the raw per-dispatch CSVs are tens of MiB per pass: gpurun copies at most 64 MiB back, the summaries are what is kept
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
