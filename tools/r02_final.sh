#!/bin/bash
# Round-2 final evidence on the GPU box: the whole -m gpu suite, smoke(), bench + rocprofv3 stats + PMC passes (r02_profile.sh),
# the hipEvent timeline of the pipelined frame, frame parts, and the stand-alone kernel timings of the widening rows.
O=gpurun_out/r02final; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash tools/r02_profile.sh
GR_TIMING_DUMP=$O/timeline_hipevents.txt timeout 120 python tools/gpu_timeline.py > /dev/null 2>&1; tail -60 $O/timeline_hipevents.txt > $O/timeline_tail.txt
for m in full postonly config4; do timeout 120 python tools/frame_parts.py $m 300 2>&1 | head -8; done | tee $O/frame_parts.txt
timeout 120 python tools/lighting_only.py 2>&1 | tail -3 | tee $O/lighting_only.txt
timeout 200 python tools/aa_time.py > $O/aa_time.txt 2>&1; tail -5 $O/aa_time.txt
timeout 100 python tools/ssr_time.py 2>&1 | tail -6 | tee $O/ssr_time.txt
timeout 100 python tools/spd_time.py 2>&1 | tail -3 | tee $O/spd_time.txt
timeout 100 python tools/hiz_time.py 2>&1 | tail -2 | tee $O/hiz_time.txt
