#!/bin/bash
# Round 3, third GPU batch: the reworked bit-plane weight pass (32-bit tiles, +-64 halo, parallel pack), SLP on / off for aa.hip.
O=gpurun_out/r03c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_aa.py tests/test_gpu_fullsize.py::test_config4_smaa_taa_sequence_matches_oracle_at_4k tests/test_gpu_strips.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python tools/aa_time.py > $O/aa_time.txt 2>&1; grep -E "FXAA|Ultra|High  |TAA|edge pixels" $O/aa_time.txt
echo "--- SLP-vectorised aa.hip"
GRANITE_LIB_DIR=lib_slp timeout 300 python tools/aa_time.py > $O/aa_time_slp.txt 2>&1; grep -E "FXAA|Ultra|TAA" $O/aa_time_slp.txt
timeout 200 python bench.py --workload config4_4k_smaa_taa > $O/bench_config4.json 2> $O/bench_config4.err; python tools/bench_brief.py $O/bench_config4.json
timeout 400 bash tools/pmc_aa.sh > $O/pmc_aa.log 2>&1; cp gpurun_out/pmc_aa/summary.txt $O/pmc_aa_summary.txt; cp gpurun_out/pmc_aa/summary.json $O/pmc_aa_summary.json
