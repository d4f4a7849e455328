#!/bin/bash
# Round 3, second GPU batch: AA parity subset after the bit-plane weight pass, stand-alone AA timings, config-4 bench, PMC.
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_aa.py tests/test_gpu_fullsize.py tests/test_gpu_strips.py tests/test_gpu_headless.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 300 python tools/aa_time.py > $O/aa_time.txt 2>&1; cat $O/aa_time.txt | tail -40
timeout 200 python bench.py --workload config4_4k_smaa_taa > $O/bench_config4.json 2> $O/bench_config4.err; python tools/bench_brief.py $O/bench_config4.json 2>/dev/null || tail -c 300 $O/bench_config4.json
timeout 400 bash tools/pmc_aa.sh > $O/pmc_aa.log 2>&1; cp gpurun_out/pmc_aa/summary.txt $O/pmc_aa_summary.txt 2>/dev/null; tail -12 $O/pmc_aa.log | cut -c1-200
