#!/bin/bash
# Round 4, eighth GPU batch: colour accumulation of the lighting walk on the matrix pipe (v_mfma_f32_4x4x1, -DLV_MFMA_ACC=1): parity, alone, frame.
O=gpurun_out/r04h; mkdir -p $O
for lib in lib_mfma; do
GRANITE_LIB_DIR=$lib timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_lighting_adversarial.py tests/test_gpu_fullsize.py tests/test_gpu_packed_hdr.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 | cut -c1-400
done
for lib in lib lib_mfma lib_mfma_noslp lib lib_mfma lib_mfma_noslp; do
  for i in 1 2; do GRANITE_LIB_DIR=$lib timeout 120 python tools/lighting_only.py 2>/dev/null | sed "s/^/alone $lib /"; done
  for i in 1 2; do GRANITE_LIB_DIR=$lib timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$lib.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_$lib.$i.json | sed "s/^/$lib /"; done
done
