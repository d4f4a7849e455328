#!/bin/bash
# Round 4, second GPU batch: persistent lighting with one queue per workgroup + stealing.
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_lighting_adversarial.py tests/test_gpu_packed_hdr.py tests/test_gpu_strips.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -8 | cut -c1-300 > $O/pytest_lighting.txt; cat $O/pytest_lighting.txt
for form in persistent 1; do
  if [ $form = persistent ]; then unset GR_LIGHTING_STATIC; else export GR_LIGHTING_STATIC=$form; fi
  for i in 1 2; do timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone form=$form /"; done
  for i in 1 2; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$form.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_$form.$i.json | sed "s/^/form=$form /"; done
done
unset GR_LIGHTING_STATIC
GRANITE_LIB_DIR=lib_stamp timeout 200 python tools/lighting_stamps.py $O/stamps_persistent.txt > /dev/null 2>$O/stamps_persistent.err; head -22 $O/stamps_persistent.txt
for wgs in 3; do GR_LIGHTING_WGS_PER_CU=$wgs timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone persistent wgs=$wgs /"; done
