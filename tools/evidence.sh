#!/bin/bash
# End-of-round evidence at HEAD, one GPU box, everything into gpurun_out/<tag> (copy what is to be judged into profiles/):
#   whole -m gpu suite, smoke(), every bench workload + the driver's command line, the 2-rank one-GPU functional run,
#   PMC counters of a short bench run (separate passes: tools/pmc_passes.sh) -> pmc_traffic.json, kernel stats under rocprofv3.
# Usage (through gpurun): bash tools/evidence.sh r04z
TAG=${1:-evidence}; O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_brief.py $O/bench_default.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_driver_line.$i.json; done
for wl in config1_256_post_only config2_1080p_256lights config3_4k_4096lights_b10g11r11 config4_4k_smaa_taa config5_8k; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --sustain-seconds 1 > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
done
timeout 400 bash tools/multirank_one_gpu.sh 2 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0.5 > $O/multirank_2.json 2> $O/multirank_2.err; tail -c 600 $O/multirank_2.json; echo
timeout 900 bash tools/pmc_passes.sh pmc_$TAG > $O/pmc_passes.log 2>&1; python tools/pmc_to_traffic.py gpurun_out/pmc_$TAG/summary.json $O/pmc_traffic.json ${TAG//[^0-9]/} | tail -2
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_counters_per_kernel.txt; rm -rf gpurun_out/pmc_$TAG
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kstats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null)
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kstats; head -6 $O/bench_kernel_stats.csv | cut -c1-140
du -sh gpurun_out
