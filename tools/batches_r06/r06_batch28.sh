#!/bin/bash
# Round 6: the HDR input pass as a conditional pass (config 1: four runtime calls a frame instead of seven): executor tests, config 1 lines.
O=gpurun_out/r06ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_app.py tests/test_gpu_graph_random.py tests/test_gpu_golden.py tests/test_gpu_headless.py tests/test_gpu_post.py tests/test_gpu_aa.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $O/pytest.txt
for i in 1 2 3; do timeout 300 python bench.py --workload config1_256_post_only --steps 200 --warmup 20 --sustain-seconds 1 --no-cpu-baseline > $O/bench_c1_200.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_c1_200.$i.json; done | tee $O/bench.txt
for i in 1 2 3; do timeout 300 python bench.py --workload config1_256_post_only --steps 20 --warmup 5 --sustain-seconds 1 --no-cpu-baseline > $O/bench_c1_20.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_c1_20.$i.json; done | tee -a $O/bench.txt
