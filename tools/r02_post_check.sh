O=gpurun_out/r02b9; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_app.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py tests/test_gpu_strips.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|assert" | head
  python tools/frame_parts.py full; python tools/frame_parts.py postonly
  python tools/post_only.py 2>&1 | tail -2 ) 2>&1 | tee $O/post_check.txt
