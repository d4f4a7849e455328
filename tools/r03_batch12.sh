#!/bin/bash
# Round 3, twelfth GPU batch: the front of odd / even frames on two streams (whole GPU suite + A/B against the single front stream).
O=gpurun_out/r03l; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt | cut -c1-300
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
s=j.get('sustained') or {}
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'sustained %.4f' % s.get('ms_per_step',0), 'light_us %.1f' % (r.get('avg_launch_us') or 0), 'host %.3f' % j.get('host_busy_ms_per_step',0))
PY
}
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --sustain-seconds 0.5 > $O/alt.$i.json 2>/dev/null; brief $O/alt.$i.json two_front_streams
  GRANITE_SINGLE_FRONT_STREAM=1 timeout 200 python bench.py --no-cpu-baseline --sustain-seconds 0.5 > $O/single.$i.json 2>/dev/null; brief $O/single.$i.json one_front_stream
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/alt20.$i.json 2>/dev/null; brief $O/alt20.$i.json two_front_streams
  GRANITE_SINGLE_FRONT_STREAM=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/single20.$i.json 2>/dev/null; brief $O/single20.$i.json one_front_stream
done
for wl in config2_1080p_256lights config4_4k_smaa_taa; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline --sustain-seconds 0.5 > $O/alt_$wl.json 2>/dev/null; brief $O/alt_$wl.json two_$wl
  GRANITE_SINGLE_FRONT_STREAM=1 timeout 200 python bench.py --workload $wl --no-cpu-baseline --sustain-seconds 0.5 > $O/single_$wl.json 2>/dev/null; brief $O/single_$wl.json one_$wl
done
