#!/bin/bash
# Round 3, nineteenth GPU batch: downsample-0 + downsample-1 as one launch: parity (levels, bands, whole frames, ranks) and what it buys.
O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_post.py tests/test_gpu_app.py tests/test_gpu_golden.py tests/test_gpu_strips.py tests/test_gpu_fullsize.py tests/test_gpu_graph_random.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-300
for i in 1 2; do
  timeout 300 python bench.py --workload config2_1080p_256lights --no-cpu-baseline > $O/c2.$i.json 2>/dev/null; python tools/bench_brief.py $O/c2.$i.json
  GR_NO_MID_FUSION=1 timeout 300 python bench.py --workload config2_1080p_256lights --no-cpu-baseline > $O/c2_nomid.$i.json 2>/dev/null; python tools/bench_brief.py $O/c2_nomid.$i.json
  timeout 300 python bench.py --no-cpu-baseline > $O/def.$i.json 2>/dev/null; python tools/bench_brief.py $O/def.$i.json
  GR_NO_MID_FUSION=1 timeout 300 python bench.py --no-cpu-baseline > $O/def_nomid.$i.json 2>/dev/null; python tools/bench_brief.py $O/def_nomid.$i.json
done
timeout 300 python bench.py --workload config1_256_post_only --no-cpu-baseline > $O/c1.json 2>/dev/null; python tools/bench_brief.py $O/c1.json
GR_NO_MID_FUSION=1 timeout 300 python bench.py --workload config1_256_post_only --no-cpu-baseline > $O/c1_nomid.json 2>/dev/null; python tools/bench_brief.py $O/c1_nomid.json
