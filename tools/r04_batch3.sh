#!/bin/bash
# Round 4, third GPU batch: waves per workgroup (4 / 2 / 1), the sat form of 1 / dist^2, ticket-counter microbenchmark.
O=gpurun_out/r04c; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_bench tools/atomic_bench.hip && timeout 120 /tmp/atomic_bench > $O/atomic_bench.txt 2>&1; cat $O/atomic_bench.txt
for lib in lib lib_w2 lib_w1 lib_a2 lib_w1a2; do
  export GRANITE_LIB_DIR=$lib
  for i in 1 2; do timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $lib /"; done
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_$lib.json | sed "s/^/$lib /"
done
for lib in lib_w1a2 lib; do
  GRANITE_LIB_DIR=$lib timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_lighting_adversarial.py tests/test_gpu_packed_hdr.py tests/test_gpu_strips.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -8 | cut -c1-400 > $O/pytest_$lib.txt; echo "== pytest $lib"; cat $O/pytest_$lib.txt
done
