#!/bin/bash
# Round 3, fifth GPU batch: the whole -m gpu suite (sampler model, rebuilt AA kernels, B10G11R11 targets), the packed-HDR bench
# line, the default bench line, fresh PMC counters for profiles/pmc_traffic.json and the rocprofv3 kernel statistics.
O=gpurun_out/r03e; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
timeout 200 python bench.py --workload config3_4k_4096lights_b10g11r11 > $O/bench_packed.json 2> $O/bench_packed.err; python tools/bench_brief.py $O/bench_packed.json
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json
timeout 200 python bench.py --workload config4_4k_smaa_taa > $O/bench_config4.json 2> $O/bench_config4.err; python tools/bench_brief.py $O/bench_config4.json
timeout 300 python tools/aa_time.py > $O/aa_time.txt 2>&1; grep -E "FXAA|Low  |Ultra|TAA|edge pixels" $O/aa_time.txt
timeout 600 bash tools/pmc_passes.sh pmc_r03 > $O/pmc_passes.log 2>&1; python tools/pmc_to_traffic.py gpurun_out/pmc_r03/summary.json $O/pmc_traffic.json 3 | tail -2
cp gpurun_out/pmc_r03/summary.txt $O/pmc_counters_per_kernel.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03e/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03e/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-150
