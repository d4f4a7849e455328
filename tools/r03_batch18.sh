#!/bin/bash
# Round 3, eighteenth GPU batch: fused pyramid tails on odd level sizes (1080p): parity, config 2.
O=gpurun_out/r03r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_post.py tests/test_gpu_app.py tests/test_gpu_golden.py tests/test_gpu_headless.py tests/test_gpu_fullsize.py tests/test_gpu_packed_hdr.py tests/test_gpu_strips.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python bench.py --workload config2_1080p_256lights --no-cpu-baseline > $O/c2.json 2>/dev/null; python tools/bench_brief.py $O/c2.json
timeout 300 python bench.py > $O/def.json 2>/dev/null; python tools/bench_brief.py $O/def.json
