#!/bin/bash
# Round 3, fourth GPU batch: compacted edge-pixel lists (SMAA weights, FXAA), branch-free bit-tile sampler; clock-ramp experiment.
O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_aa.py tests/test_gpu_fullsize.py::test_config4_smaa_taa_sequence_matches_oracle_at_4k tests/test_gpu_strips.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python tools/aa_time.py > $O/aa_time.txt 2>&1; grep -E "FXAA|Low  |Ultra|TAA|edge pixels" $O/aa_time.txt
timeout 200 python bench.py --workload config4_4k_smaa_taa > $O/bench_config4.json 2> $O/bench_config4.err; python tools/bench_brief.py $O/bench_config4.json
echo "--- driver-style run (20 steps), warm-up 5 vs 400 frames"
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/bench_w5.json 2>/dev/null; python tools/bench_brief.py $O/bench_w5.json
timeout 200 python bench.py --steps 20 --warmup 400 --no-cpu-baseline --sustain-seconds 0 > $O/bench_w400.json 2>/dev/null; python tools/bench_brief.py $O/bench_w400.json
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200.json 2>/dev/null; python tools/bench_brief.py $O/bench_200.json
timeout 400 bash tools/pmc_aa.sh > $O/pmc_aa.log 2>&1; cp gpurun_out/pmc_aa/summary.txt $O/pmc_aa_summary.txt; cp gpurun_out/pmc_aa/summary.json $O/pmc_aa_summary.json
