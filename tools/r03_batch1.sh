#!/bin/bash
# Round 3, first GPU batch: the whole -m gpu suite after the sampler-model change + the rebuilt FXAA / SMAA edge / blend / TAA
# kernels, their stand-alone timings at 4K, PMC passes over them, and the config-4 / default bench lines.
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 300 python tools/aa_time.py > $O/aa_time.txt 2>&1; cat $O/aa_time.txt | tail -40
timeout 200 python bench.py --workload config4_4k_smaa_taa > $O/bench_config4.json 2> $O/bench_config4.err; tail -c 600 $O/bench_config4.json
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
timeout 400 bash tools/pmc_aa.sh > $O/pmc_aa.log 2>&1; cp gpurun_out/pmc_aa/summary.txt $O/pmc_aa_summary.txt 2>/dev/null; tail -30 $O/pmc_aa.log
