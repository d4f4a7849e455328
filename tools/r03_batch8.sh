#!/bin/bash
# Round 3, eighth GPU batch: what the driver's 20-step run pays for -- event fences (system vs device scope) and bracket count.
O=gpurun_out/r03h; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_app.py -q -m gpu -k "saved_state or replay" 2>&1 | tail -3
run() { # name, env...
  local name=$1; shift
  for i in 1 2 3; do
    env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0.5 > $O/$name.$i.json 2>/dev/null
    python - "$O/$name.$i.json" "$name" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
print(sys.argv[2], 'ms/step %.4f' % j['ms_per_step'], 'sustained %.4f' % j['sustained']['ms_per_step'], 'light_us %.1f' % (r.get('avg_launch_us') or 0), 'brackets', r.get('launches'))
PY
  done
}
run dev16 A=1
run sys16 GRANITE_TIMING_EVENT_SYSTEM_FENCE=1 GRANITE_SYNC_EVENT_SYSTEM_FENCE=1
run devsync_systime GRANITE_TIMING_EVENT_SYSTEM_FENCE=1
run dev5 GRANITE_BENCH_MIN_BRACKETS=5
run dev1 GRANITE_BENCH_MIN_BRACKETS=1
for wl in config2_1080p_256lights; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python tools/bench_brief.py $O/bench_$wl.json
  GRANITE_SYNC_EVENT_SYSTEM_FENCE=1 timeout 200 python bench.py --workload $wl --no-cpu-baseline > $O/bench_${wl}_sysfence.json 2>/dev/null; python tools/bench_brief.py $O/bench_${wl}_sysfence.json
done
