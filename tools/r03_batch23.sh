#!/bin/bash
# Round 3, twenty-third GPU batch: rocprofv3 --kernel-trace --stats and PMC (instructions, LDS, FETCH / WRITE) of the AA kernels on
# the test card alone (the per-kernel means of the earlier passes mix both inputs).
O=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AA_TIME_INPUT=card
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kstats -o aa --output-format csv -- python $GRAFT_REPO_ROOT/tools/aa_time.py 3840 2160 > $O/aa_time_under_rocprof.txt 2>&1
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/aa_kernel_stats.csv; rm -rf $O/kstats; grep -E "fxaa|smaa|taa" $O/aa_kernel_stats.csv | cut -c1-170
run() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d "$O/pmc/$name" -o "$name" --output-format csv -- python $GRAFT_REPO_ROOT/tools/aa_time.py 3840 2160 > "$O/pmc_$name.log" 2>&1; echo "$name rc=$?"; }
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run mem FETCH_SIZE WRITE_SIZE
python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$O/pmc" > "$O/pmc_aa_test_card.txt" 2>&1
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; rm -rf $O/pmc
grep -E "^kernel|fxaa|smaa|taa" "$O/pmc_aa_test_card.txt" | cut -c1-260
