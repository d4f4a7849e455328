#!/bin/bash
# Round 4, sixth GPU batch: whole suite at HEAD (failures printed), SLP on / off for the lighting walk.
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $O/pytest_gpu_full.txt; grep -n "out of tolerance\|Mismatched\|^FAILED\|^E  .*Error\|passed\|failed" $O/pytest_gpu_full.txt | cut -c1-400 | head -40; tail -60 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt; rm $O/pytest_gpu_full.txt
for lib in lib lib_noslp lib lib_noslp; do
  for i in 1 2; do GRANITE_LIB_DIR=$lib timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $lib /"; done
  GRANITE_LIB_DIR=$lib timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_$lib.json | sed "s/^/$lib /"
done
