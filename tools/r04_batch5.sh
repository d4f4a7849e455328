#!/bin/bash
# Round 4, fifth GPU batch: whole suite; lighting walk packed (lib) vs scalar (lib_nopk); AA tile order: screen (lib_aa0) vs XCD rows 1 (lib) / 2 / 4.
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt | cut -c1-400
for lib in lib lib_nopk lib lib_nopk; do
  for i in 1 2; do GRANITE_LIB_DIR=$lib timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $lib /"; done
  GRANITE_LIB_DIR=$lib timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_$lib.json | sed "s/^/$lib /"
done
for lib in lib_aa0 lib lib_aa2 lib_aa4; do
  GRANITE_LIB_DIR=$lib timeout 300 python tools/aa_time.py > $O/aa_time_$lib.txt 2>&1; echo "== $lib"; grep -E "FXAA|Ultra|TAA|card|noise" $O/aa_time_$lib.txt | cut -c1-150
done
for lib in lib_aa0 lib; do
  GRANITE_LIB_DIR=$lib timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/bench_config4_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_config4_$lib.json | sed "s/^/$lib /"
done
