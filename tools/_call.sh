bash tools/pmc_aa.sh 2>&1 | tail -3
timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline --sustain-seconds 1 > gpurun_out/r05_bench_config4_final.json 2>/dev/null; python tools/bench_brief.py gpurun_out/r05_bench_config4_final.json
timeout 600 python -m pytest tests/test_gpu_aa.py tests/test_gpu_strips.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
