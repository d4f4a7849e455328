GR_UP_ALL_THREADS=512 timeout 600 python -m pytest tests/test_gpu_post.py -x -q -k "fused_upsample" 2>&1 | grep -E "passed|failed" | tail -2
bash tools/frame_ab.sh r05_up512 "config4_4k_smaa_taa config3_4k_4096lights" wide narrow:GR_UP_ALL_THREADS=512 -- --steps 100 --warmup 10 --sustain-seconds 1
