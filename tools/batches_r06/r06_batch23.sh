#!/bin/bash
O=gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_app.py tests/test_gpu_golden.py tests/test_gpu_headless.py tests/test_gpu_graph_random.py tests/test_gpu_lighting.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
bash tools/frame_ab.sh r06w/ab "config2_1080p_256lights" staged upload:GRANITE_CLUSTER_READ_STAGED_MAX_LIGHTS=0 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06w/ab20 "config2_1080p_256lights" staged upload:GRANITE_CLUSTER_READ_STAGED_MAX_LIGHTS=0 -- --steps 20 --warmup 5 --sustain-seconds 1 2>&1 | tee $O/ab20.txt
