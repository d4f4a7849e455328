#!/bin/bash
# Round 4, ninth GPU batch: the last run of a stream publishes under the device's frame fence (3 event records per frame instead of 6): executor tests, frame times.
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_app.py tests/test_gpu_graph_random.py tests/test_gpu_strips.py tests/test_gpu_headless.py tests/test_gpu_multiprocess.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 | cut -c1-400
for i in 1 2 3; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench.$i.json; done
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $O/bench200.json 2>/dev/null; python tools/bench_brief.py $O/bench200.json | sed "s/^/200 steps /"
for wl in config1_256_post_only config2_1080p_256lights config4_4k_smaa_taa; do for i in 1 2; do timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.$i.json | sed "s/^/$wl /"; done; done
