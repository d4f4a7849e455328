#!/bin/bash
# Round 3, sixth GPU batch: whole -m gpu suite (fused cluster front, exact SSR, packed HDR through the executor), config 2 / 3 bench
# lines with the cluster build as two launches vs launch by launch, PMC counters for profiles/pmc_traffic.json, rocprofv3 kernel stats.
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -60 > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt | cut -c1-300
for wl in config2_1080p_256lights config3_4k_4096lights; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
  GRANITE_CLUSTER_SEPARATE_LAUNCHES=1 timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_${wl}_separate.json 2>/dev/null; python tools/bench_brief.py $O/bench_${wl}_separate.json
done
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2>/dev/null; python tools/bench_brief.py $O/bench_driver_style.json
timeout 600 bash tools/pmc_passes.sh pmc_r03 > $O/pmc_passes.log 2>&1; python tools/pmc_to_traffic.py gpurun_out/pmc_r03/summary.json $O/pmc_traffic.json 3 | tail -2
cp gpurun_out/pmc_r03/summary.txt $O/pmc_counters_per_kernel.txt; cp gpurun_out/pmc_r03/summary.json $O/pmc_summary.json; rm -rf gpurun_out/pmc_r03
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03f/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03f/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -14 $O/bench_kernel_stats.csv | cut -c1-160
du -sh gpurun_out
