#!/bin/bash
# Round 3, twenty-fourth GPU batch: HBM traffic (FETCH_SIZE / WRITE_SIZE) of the AA kernels on the test card, few launches per kernel.
O=$GRAFT_REPO_ROOT/gpurun_out/r03x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AA_TIME_INPUT=card AA_TIME_REPS=2
timeout 500 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d "$O/pmc/mem" -o mem --output-format csv -- python $GRAFT_REPO_ROOT/tools/aa_time.py 3840 2160 > "$O/pmc_mem.log" 2>&1; echo "mem rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$O/pmc" > "$O/pmc_aa_traffic_test_card.txt" 2>&1
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; rm -rf $O/pmc
grep -E "^kernel|fxaa|smaa|taa" "$O/pmc_aa_traffic_test_card.txt" | cut -c1-200
