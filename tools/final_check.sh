#!/bin/bash
# Round-end check on the GPU box: whole -m gpu suite, smoke(), the SPD timing, the default bench line.
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 > gpurun_out/final/pytest_gpu.txt
cat gpurun_out/final/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 120 python tools/spd_time.py 2>&1 | tail -3 | tee gpurun_out/final/spd_time.txt
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/final/bench.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/final/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('valu_instructions_per_launch'))
PY
