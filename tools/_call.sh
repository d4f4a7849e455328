timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r05_pytest_gpu2.log | tail -3
timeout 300 python tools/ulp_hist.py > gpurun_out/r05_ulp_lighting_4k.json 2>/dev/null; cat gpurun_out/r05_ulp_lighting_4k.json
timeout 600 python tools/pyramid_ulp_hist.py > gpurun_out/r05_pyramid_ulp_4k.json 2>gpurun_out/r05_pyramid_ulp_4k.err; cut -c1-3000 gpurun_out/r05_pyramid_ulp_4k.json; tail -3 gpurun_out/r05_pyramid_ulp_4k.err
timeout 400 bash tools/multirank_one_gpu.sh 2 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0.5 > gpurun_out/r05_multirank_2.json 2> gpurun_out/r05_multirank_2.err; tail -c 2500 gpurun_out/r05_multirank_2.json; echo; tail -3 gpurun_out/r05_multirank_2.err
timeout 900 bash tools/pmc_passes.sh pmc_r05a > gpurun_out/r05_pmc_passes.log 2>&1; cut -c1-330 gpurun_out/pmc_r05a/summary.txt | grep -v "rocclr"
