#!/bin/bash
# Round 3, seventh GPU batch: whole -m gpu suite (fog quad, checkpoint / replay, cluster build as upload + front + binning, RCCL
# stand-in gating), cluster A/B, the 2-rank one-GPU functional run with its rccl record, PMC counters, kernel stats.
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -60 > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt | cut -c1-300
for wl in config2_1080p_256lights config3_4k_4096lights; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
  GRANITE_CLUSTER_SEPARATE_LAUNCHES=1 timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_${wl}_separate.json 2>/dev/null; python tools/bench_brief.py $O/bench_${wl}_separate.json
done
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json
timeout 300 bash tools/multirank_one_gpu.sh 2 > $O/multirank_2.json 2> $O/multirank_2.err; tail -c 700 $O/multirank_2.json
timeout 600 bash tools/pmc_passes.sh pmc_r03 > $O/pmc_passes.log 2>&1; python tools/pmc_to_traffic.py gpurun_out/pmc_r03/summary.json $O/pmc_traffic.json 3 | tail -2
cp gpurun_out/pmc_r03/summary.txt $O/pmc_counters_per_kernel.txt; cp gpurun_out/pmc_r03/summary.json $O/pmc_summary.json; rm -rf gpurun_out/pmc_r03
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03g/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03g/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -8 $O/bench_kernel_stats.csv | cut -c1-140
du -sh gpurun_out
