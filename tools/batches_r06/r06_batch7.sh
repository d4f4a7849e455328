#!/bin/bash
# Round 6: the two worst-case scenes (granite_amd/synth.py): parity at 4K, the lighting launch alone, the frame.
O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "worst_case or config3" 2>&1 | tail -5 | tee $O/pytest.txt
for round in 1 2; do for sc in default depth_split hot_spot; do timeout 200 python tools/lighting_only.py 3840 2160 $sc 2>/dev/null | sed "s/^/alone /"; done; done | tee $O/alone.txt
for sc in default depth_split hot_spot; do
  timeout 300 python bench.py --scene $sc --steps 100 --warmup 10 --sustain-seconds 1 --no-cpu-baseline > $O/bench_$sc.json 2>/dev/null
  python tools/bench_brief.py $O/bench_$sc.json | sed "s/^/frame $sc /"
done | tee $O/frame.txt
