#!/bin/bash
# Round 6: window gap + floor list in the lighting kernel: parity on the three scenes, the launch alone against the previous kernel (lib_head)
# and two SLP-threshold builds, the VALU class-mix microbenchmark.
O=gpurun_out/r06i; mkdir -p $O
./tools/valu_mix_bench.bin > $O/valu_mix.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_lighting.py tests/test_gpu_lighting_adversarial.py tests/test_gpu_packed_hdr.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
alone() { ( export GRANITE_LIB_DIR=$1; timeout 200 python tools/lighting_only.py 3840 2160 $2 2>/dev/null | sed "s/^/alone $1 /" ) }
for round in 1 2; do for sc in default depth_split hot_spot; do for l in lib lib_head; do alone $l $sc; done; done; for l in lib_slp4 lib_slp8; do alone $l default; done; done 2>&1 | tee $O/alone.txt
for sc in default depth_split hot_spot; do python tools/ulp_hist.py 3840 2160 4096 $sc > $O/ulp_$sc.txt 2>&1; head -c 600 $O/ulp_$sc.txt; echo; done
for sc in default depth_split hot_spot; do
  timeout 300 python bench.py --scene $sc --steps 100 --warmup 10 --sustain-seconds 1 --no-cpu-baseline > $O/bench_$sc.json 2>/dev/null
  python tools/bench_brief.py $O/bench_$sc.json | sed "s/^/frame $sc /"
done | tee $O/frame.txt
