O=gpurun_out/r02b4; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_fullsize.py tests/test_gpu_app.py -q -m gpu -x 2>&1 | tail -5
  timeout 120 python tools/lighting_only.py 2>&1 | tail -1
  timeout 200 python tools/ulp_hist.py 2>&1 | tail -1
  python tools/frame_parts.py full ) 2>&1 | tee $O/light_ab.txt
