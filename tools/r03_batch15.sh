#!/bin/bash
# Round 3, fifteenth GPU batch: is the slow start of a short timed region a clock ramp?  N un-bracketed frames right before it.
O=gpurun_out/r03o; mkdir -p $O
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'light_us %.1f' % (r.get('avg_launch_us') or 0))
PY
}
for n in 0 10 40 160 0 40; do
  GRANITE_BENCH_PREROLL_FRAMES=$n timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/pre$n.json 2>/dev/null; brief $O/pre$n.json preroll_$n
done
