#!/bin/bash
# Round 6: the front's stream alternates with the frame's parity: executor tests, A/B against one front stream on configs 3 / 4 / 2 / 5.
O=gpurun_out/r06o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_app.py tests/test_gpu_graph_random.py tests/test_gpu_fullsize.py tests/test_gpu_headless.py tests/test_gpu_golden.py tests/test_gpu_strips.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
bash tools/frame_ab.sh r06o/ab "config3_4k_4096lights config4_4k_smaa_taa" alt one:GRANITE_ALTERNATE_FRONT=0 -- --steps 200 --warmup 20 --sustain-seconds 1 2>&1 | tee $O/ab.txt
bash tools/frame_ab.sh r06o/ab20 "config3_4k_4096lights config2_1080p_256lights config5_8k" alt one:GRANITE_ALTERNATE_FRONT=0 -- --steps 20 --warmup 5 --sustain-seconds 1 2>&1 | tee $O/ab20.txt
