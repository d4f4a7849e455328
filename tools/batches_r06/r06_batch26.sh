#!/bin/bash
# Round 6: events of frames the host has already waited for cost no query: executor tests, host time per frame on configs 1 / 2 / 3.
O=gpurun_out/r06aa; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_app.py tests/test_gpu_graph_random.py tests/test_gpu_golden.py tests/test_gpu_headless.py tests/test_gpu_strips.py tests/test_gpu_multiprocess.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $O/pytest.txt
for wl in config1_256_post_only config2_1080p_256lights config3_4k_4096lights; do for i in 1 2; do
  timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --sustain-seconds 1 --no-cpu-baseline > $O/bench_$wl.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.$i.json
done; done | tee $O/bench.txt
