#!/bin/bash
# Round 3, sixteenth GPU batch: bench.py with the clock-settle phase -- the driver's command line, the default run, off for comparison.
O=gpurun_out/r03p; mkdir -p $O
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
s=j.get('sustained') or {}
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'Gpx/s %.2f' % (j['value']/1000), 'sustained %.4f' % s.get('ms_per_step',0), 'light_us %.1f' % (r.get('avg_launch_us') or 0), 'frac %.3f' % r['frac'], 'settle', j['clock_settle']['frames'])
PY
}
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.$i.json 2>/dev/null; brief $O/driver.$i.json driver_line
done
GRANITE_BENCH_SETTLE_MS=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_cold.json 2>/dev/null; brief $O/driver_cold.json driver_line_no_settle
timeout 300 python bench.py > $O/default.json 2>/dev/null; brief $O/default.json default
timeout 300 python bench.py --workload config2_1080p_256lights --no-cpu-baseline > $O/config2.json 2>/dev/null; brief $O/config2.json config2
