#!/bin/bash
# Round 3, tenth GPU batch: (1) lighting launch durations over a 20-frame run (ramp after an idle gap?), (2) HBM probe + warm-up
# right before the timed frames vs the old order, (3) up-tail 256 vs 1024 threads on the same box.
O=gpurun_out/r03j; mkdir -p $O
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
s=j.get('sustained') or {}
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'sustained %.4f' % s.get('ms_per_step',0), 'light_us %.1f' % (r.get('avg_launch_us') or 0), 'host %.3f' % j.get('host_busy_ms_per_step',0))
PY
}
GRANITE_BENCH_ORDER=old GR_TIMING_DUMP=$O/spans_old.txt timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/dump_old.json 2>/dev/null
python tools/span_ramp.py $O/spans_old.txt lighting > $O/ramp_old.txt; tail -22 $O/ramp_old.txt
GR_TIMING_DUMP=$O/spans_new.txt timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/dump_new.json 2>/dev/null
python tools/span_ramp.py $O/spans_new.txt lighting > $O/ramp_new.txt; tail -22 $O/ramp_new.txt
for i in 1 2 3; do
  GRANITE_BENCH_ORDER=old timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0.3 > $O/old.$i.json 2>/dev/null; brief $O/old.$i.json old_order
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0.3 > $O/new.$i.json 2>/dev/null; brief $O/new.$i.json late_probe
  GRANITE_UP_TAIL_WIDE=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0.3 > $O/new_wide.$i.json 2>/dev/null; brief $O/new_wide.$i.json late_probe_wide_tail
done
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline > $O/def.$i.json 2>/dev/null; brief $O/def.$i.json default_narrow
  GRANITE_UP_TAIL_WIDE=1 timeout 200 python bench.py --no-cpu-baseline > $O/def_wide.$i.json 2>/dev/null; brief $O/def_wide.$i.json default_wide
done
timeout 200 python bench.py --workload config2_1080p_256lights --no-cpu-baseline > $O/c2.json 2>/dev/null; brief $O/c2.json config2_narrow
GRANITE_UP_TAIL_WIDE=1 timeout 200 python bench.py --workload config2_1080p_256lights --no-cpu-baseline > $O/c2_wide.json 2>/dev/null; brief $O/c2_wide.json config2_wide
