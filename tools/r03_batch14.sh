#!/bin/bash
# Round 3, fourteenth GPU batch: FXAA interior workgroups + scalar compaction prefix, SMAA diagonal searches as integer walks.
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_aa.py tests/test_gpu_fullsize.py::test_config4_smaa_taa_sequence_matches_oracle_at_4k tests/test_gpu_headless.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python tools/aa_time.py > $O/aa_time.txt 2>&1; grep -E "FXAA|Low  |High |Ultra|TAA|edge pixels" $O/aa_time.txt
GRANITE_SMAA_FLOAT_DIAG_WALKS=1 timeout 300 python tools/aa_time.py > $O/aa_time_float_walks.txt 2>&1; grep -E "Ultra.*blend_weight" $O/aa_time_float_walks.txt
timeout 200 python bench.py --workload config4_4k_smaa_taa > $O/bench_config4.json 2> $O/bench_config4.err; python tools/bench_brief.py $O/bench_config4.json
timeout 400 bash tools/pmc_aa.sh > $O/pmc_aa.log 2>&1; cp gpurun_out/pmc_aa/summary.txt $O/pmc_aa_summary.txt; cp gpurun_out/pmc_aa/summary.json $O/pmc_aa_summary.json; rm -rf gpurun_out/pmc_aa
grep -E "^kernel|k_fxaa_fast|k_smaa_weights_bits|k_smaa_pack" $O/pmc_aa_summary.txt | cut -c1-230
