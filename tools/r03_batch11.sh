#!/bin/bash
# Round 3, eleventh GPU batch: what sits between two long kernels on one stream (tools/stream_gap_bench.hip), stream priorities.
O=gpurun_out/r03k; mkdir -p $O
hipcc -O2 --offload-arch=gfx950 -Wno-unused-result tools/stream_gap_bench.hip -o /tmp/stream_gap_bench 2>/dev/null && /tmp/stream_gap_bench | tee $O/stream_gap.txt
brief() { python - "$1" "$2" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
r=j['roofline']
s=j.get('sustained') or {}
print(sys.argv[2], 'K', j['steps'], 'ms/step %.4f' % j['ms_per_step'], 'sustained %.4f' % s.get('ms_per_step',0), 'light_us %.1f' % (r.get('avg_launch_us') or 0), 'host %.3f' % j.get('host_busy_ms_per_step',0))
PY
}
for p in hml lmh mmm hhl llh hlm; do
  GRANITE_STREAM_PRIORITIES=$p timeout 200 python bench.py --no-cpu-baseline --sustain-seconds 0.5 > $O/prio_$p.json 2>/dev/null; brief $O/prio_$p.json prio_$p
done
