#!/bin/bash
# Round 3, final evidence batch at HEAD: whole -m gpu suite, smoke, every bench workload, the driver's command line, the 2-rank
# one-GPU functional run, PMC counters (separate passes), kernel stats under rocprofv3.  Everything lands in gpurun_out/r03z.
O=gpurun_out/r03z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_driver_line.$i.json; done
for wl in config1_256_post_only config2_1080p_256lights config3_4k_4096lights_b10g11r11 config4_4k_smaa_taa config5_8k; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
done
timeout 300 bash tools/multirank_one_gpu.sh 2 > $O/multirank_2.json 2> $O/multirank_2.err; tail -c 400 $O/multirank_2.json; echo
timeout 600 bash tools/pmc_passes.sh pmc_r03 > $O/pmc_passes.log 2>&1; python tools/pmc_to_traffic.py gpurun_out/pmc_r03/summary.json $O/pmc_traffic.json 3 | tail -2
cp gpurun_out/pmc_r03/summary.txt $O/pmc_counters_per_kernel.txt; rm -rf gpurun_out/pmc_r03
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -6 $O/bench_kernel_stats.csv | cut -c1-140
du -sh gpurun_out
