#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_app.py tests/test_gpu_graph_random.py tests/test_gpu_fullsize.py tests/test_gpu_headless.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
bash tools/frame_ab.sh r06g/ab "config4_4k_smaa_taa config3_4k_4096lights" split nosplit:GRANITE_SPLIT_TAIL=0 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
