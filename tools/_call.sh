bash tools/pmc_aa.sh 2>&1 | tail -12
cat gpurun_out/pmc_aa/aa_traffic.json | head -50
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_driver_line_final.json 2>/dev/null; python tools/bench_brief.py gpurun_out/r05_driver_line_final.json
timeout 400 python bench.py > gpurun_out/r05_bench_default.json 2>/dev/null; python tools/bench_brief.py gpurun_out/r05_bench_default.json
